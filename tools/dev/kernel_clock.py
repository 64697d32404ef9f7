"""dev tool: the clock conv3x3_halo2_kernel actually runs at, from in-kernel stamps (MI355X_MICROARCH.md, DVFS give-back item 6).
    tools/ablate.sh conv3x3_halo2 512
    python tools/dev/real_layer_bench.py capture
    CVPCE_LIB=$PWD/cvpce_amd/libcvpce_hip_conv3x3_halo2_dbg512.so python tools/dev/kernel_clock.py [conv3_2,conv4_2,...]
Each layer runs back to back for >= 2 s on its REAL pipeline input (256 crops); every workgroup's wave 0 stamps s_memtime (shader
clock) and s_memrealtime (100 MHz) around its K loop in the LAST launch.  Prints the median in-kernel clock, the layer's TFLOP/s
over the last group of launches, and that rate against the bf16 MFMA peak AT that clock (2.5 PFLOP/s x clock / 2.4 GHz)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import _lib, ops, synthetic

dev = torch.device('cuda')
L = _lib.lib
L.cvpce_debug_halo2_clock.restype = ctypes.c_int
L.cvpce_debug_halo2_clock.argtypes = [ctypes.c_void_p]
enc = synthetic.synthetic_macvgg(seed=1).to(dev)
eng = enc.engine()
names = ['conv2_1', 'conv2_2', 'conv3_1', 'conv3_2', 'conv3_3', 'conv4_1', 'conv4_2', 'conv4_3', 'conv5_1', 'conv5_2', 'conv5_3']
want = sys.argv[1].split(',') if len(sys.argv) > 1 else ['conv3_2', 'conv3_3', 'conv4_2', 'conv5_1']
convs = dict(zip(names, [(k, pc) for k, pc in eng.plan if k in ('conv', 'conv_pool')]))
for nm in want:
    kind, pc = convs[nm]
    x = torch.load(f'/tmp/real_{nm}.pt').to(dev)
    n, h, w, c = x.shape
    flop = 2.0 * n * h * w * pc.cout * 9 * pc.cin
    run = lambda: ops.conv2d(x, pc, act=1, pool=kind == 'conv_pool')
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    while True:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        tf = 20 * flop / (e0.elapsed_time(e1) * 1e-3) / 1e12
        if time.perf_counter() - t0 >= 2.0:
            break
    buf = (ctypes.c_ulonglong * 2048)()
    assert L.cvpce_debug_halo2_clock(buf) == 0
    clk = sorted(buf[2 * i] / buf[2 * i + 1] * 100.0 for i in range(1024) if buf[2 * i + 1] > 0)
    med = clk[len(clk) // 2]
    peak = 2500.0 * med / 2400.0
    print(f'{nm}: {tf:7.1f} TFLOP/s   in-kernel clock median {med:6.0f} MHz (p10 {clk[len(clk) // 10]:.0f}, p90 {clk[9 * len(clk) // 10]:.0f}, '
          f'{len(clk)} workgroups)   peak at that clock {peak:6.0f}   fraction {tf / peak:.3f}   (of nominal {tf / 2500:.3f})', flush=True)
