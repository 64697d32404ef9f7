"""dev tool: do two kernels that each need half the chip run side by side when launched on two streams?  (bare MFMA probe, 128 workgroups of 256 threads, 96 KiB LDS each = one workgroup per CU)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd.torch_ops import T
dev = torch.device('cuda')
operands = torch.randn(1 << 16, device=dev).to(torch.bfloat16)
sinks = [torch.empty(256 * 256, dtype=torch.float32, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(which, wgs, iters=200000):
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in which:
        with torch.cuda.stream(streams[i]):
            T.probe_mfma_bf16(1, iters, operands, sinks[i], wgs)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3


for wgs in (64, 128, 256):
    run([0, 1], wgs)
    a = min(run([0], wgs) for _ in range(3))
    both = min(run([0, 1], wgs) for _ in range(3))
    print(f'{wgs} workgroups per kernel: one kernel {a:.2f} ms, two kernels on two streams {both:.2f} ms', flush=True)

# the same with CU-masked streams: 224 workgroups on 7 XCDs beside 32 workgroups on the 8th
import ctypes


def masked_stream(cu_bits):
    hip = ctypes.CDLL('libamdhip64.so')
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (32 * w + b) in cu_bits) for w in range(8)])
    st = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words) == 0
    return torch.cuda.ExternalStream(st.value)


for name, pick_b in (('one XCD (bits = 7 mod 8)', {i for i in range(256) if i % 8 == 7}), ('last 32 bits', set(range(224, 256)))):
    streams[0], streams[1] = masked_stream(set(range(256)) - pick_b), masked_stream(pick_b)

    def run2(which, iters=200000):
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in which:
            with torch.cuda.stream(streams[i]):
                T.probe_mfma_bf16(1, iters, operands, sinks[i], 224 if i == 0 else 32)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) * 1e3

    run2([0, 1])
    a, b = min(run2([0]) for _ in range(3)), min(run2([1]) for _ in range(3))
    both = min(run2([0, 1]) for _ in range(3))
    print(f'masked, B = {name}: 224 WGs on A {a:.2f} ms, 32 WGs on B {b:.2f} ms, together {both:.2f} ms', flush=True)

# sequences of SHORT launches instead of one long launch per stream (the embedder is ~40 launches of 1-3 ms, the detector ~150 of 10-100 us)
for na, ita, nb, itb in ((40, 8000, 40, 8000), (40, 8000, 600, 400), (40, 8000, 3000, 80)):
    def seq(which):
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in which:
            with torch.cuda.stream(streams[i]):
                for _ in range(na if i == 0 else nb):
                    T.probe_mfma_bf16(1, ita if i == 0 else itb, operands, sinks[i], 224 if i == 0 else 32)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) * 1e3
    seq([0, 1])
    a, b = min(seq([0]) for _ in range(2)), min(seq([1]) for _ in range(2))
    both = min(seq([0, 1]) for _ in range(2))
    print(f'masked sequences: A {na} x {ita} iters = {a:.2f} ms, B {nb} x {itb} iters = {b:.2f} ms, together {both:.2f} ms', flush=True)
