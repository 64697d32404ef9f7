"""dev: where two waves of conv3x3_halo3_kernel spend their cycles, from in-kernel s_memtime sums (side library built with
tools/ablate.sh conv3x3_halo3 128; layer inputs from tools/dev/real_layer_bench.py capture)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import ops, synthetic, _lib
dev = torch.device('cuda')
enc = synthetic.synthetic_macvgg(seed=1).to(dev)
eng = enc.engine()
names = ['conv2_1', 'conv2_2']
convs = [(k, pc) for k, pc in eng.plan if k in ('conv', 'conv_pool')][:2]
_lib.lib.cvpce_debug_halo3_stamps.restype = ctypes.c_int
for nm, (kind, pc) in zip(names, convs):
    x = torch.load(f'/tmp/real_{nm}.pt').to(dev)
    for _ in range(5):
        ops.conv2d(x, pc, act=1, pool=kind == 'conv_pool')
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 12)()
    assert _lib.lib.cvpce_debug_halo3_stamps(buf) == 0
    for w in range(2):
        wt, hv, hb, ep, tot, steps = [int(v) for v in buf[6 * w:6 * w + 6]]
        print(f'{nm} wave {"05"[w]}: loop {tot} cycles (100 MHz ticks x ?), weight waits {wt} ({100.0 * wt / tot:.1f} %), hand-off vmcnt {hv} '
              f'({100.0 * hv / tot:.1f} %), hand-off barrier {hb} ({100.0 * hb / tot:.1f} %), epilogue {ep} ({100.0 * ep / tot:.1f} %), '
              f'steps with a timed weight wait {steps}, per step {tot / max(1, steps * 6 / 4):.0f}')
