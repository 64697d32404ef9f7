"""dev: what each byte stream of conv3x3_halo2_kernel costs in CLOCK and WATTS (review item 6, round 6).  Runs the embedder on 800 random
crops with the work lists off for ~3 s under the amdgpu hwmon sampler of bench.py and prints: ms per pass, the kernel's TFLOP/s (HIP events),
median shader clock and board power.  Run once per library: the shipped one and the timing-only ablations built by
`tools/ablate.sh conv3x3_halo2 8 32 4 44` (8 no weight loads in the K loop, 32 no fragment reads, 4 no patch DMA, 44 all three: MFMAs on
stale registers only -- the results are wrong by construction, only time / clock / power are read)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from cvpce_amd import ops, synthetic
from cvpce_amd.models import classification as C

dev = torch.device('cuda:0')
enc = synthetic.synthetic_macvgg(seed=1).to(dev)
eng = enc.engine()
g = torch.Generator().manual_seed(3)
crops = (torch.rand(800, 256, 256, 4, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
crops[..., 3] = 0
for _ in range(3):
    eng.embed_packed(crops)
torch.cuda.synchronize()
with bench.ClockSampler(dev) as cs:
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < 3.0:
        eng.embed_packed(crops)
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
clk = cs.summary()
ops.PROFILE = ops.ConvProfile()
for _ in range(3):
    eng.embed_packed(crops)
summ = ops.PROFILE.summary()
ops.PROFILE = None
kname = sys.argv[1] if len(sys.argv) > 1 else 'conv3x3_halo2_kernel'
h2 = summ[kname]
print(f"{os.path.basename(os.environ.get('CVPCE_LIB', 'libcvpce_hip.so')):44s} {dt / n * 1e3:7.2f} ms/pass  {kname} {h2['flops'] / h2['ms'] / 1e9:7.1f} TFLOP/s "
      f"({h2['ms'] / h2['launches'] * 1e3:6.0f} us/launch)  sclk {clk['sclk_mhz_median']} MHz  power {clk['power_w_median']} W", flush=True)
