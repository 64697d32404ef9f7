"""Per-layer conv microbenchmark (dev tool): VGG16 / ResNet shapes through ops.conv2d, TFLOP/s per layer."""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpce_amd import ops

LAYERS = {
    # name: (N, Cin, H, W, Cout, k, stride, pad, pool)
    'vgg1_1': (256, 3, 256, 256, 64, 3, 1, 1, False),
    'vgg1_2': (256, 64, 256, 256, 64, 3, 1, 1, True),
    'vgg2_1': (256, 64, 128, 128, 128, 3, 1, 1, False),
    'vgg2_2': (256, 128, 128, 128, 128, 3, 1, 1, True),
    'vgg3_1': (256, 128, 64, 64, 256, 3, 1, 1, False),
    'vgg3_2': (256, 256, 64, 64, 256, 3, 1, 1, False),
    'vgg4_1': (256, 256, 32, 32, 512, 3, 1, 1, False),
    'vgg4_2': (256, 512, 32, 32, 512, 3, 1, 1, False),
    'vgg4_3': (256, 512, 32, 32, 512, 3, 1, 1, True),
    'vgg5_1': (256, 512, 16, 16, 512, 3, 1, 1, False),
    'res_c2_1x1': (8, 64, 200, 200, 256, 1, 1, 0, False),
    'res_c3_3x3': (8, 128, 100, 100, 128, 3, 1, 1, False),
    'head_p3': (8, 256, 100, 100, 256, 3, 1, 1, False),
    # small-M detector layers (4 images): latency-bound launches of the register-staged kernel
    'l3_c1': (4, 1024, 50, 50, 256, 1, 1, 0, False),
    'l4_c1': (4, 2048, 25, 25, 512, 1, 1, 0, False),
    'l4_c2': (4, 512, 25, 25, 512, 3, 1, 1, False),
    'l4_c3': (4, 512, 25, 25, 2048, 1, 1, 0, False),
    'l4_c3_res': (4, 512, 25, 25, 2048, 1, 1, 0, False),
}

ap = argparse.ArgumentParser()
ap.add_argument('--layers', default=','.join(LAYERS))
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--generic', type=int, default=0)
ap.add_argument('--dbg', type=int, default=0)
ap.add_argument('--no-halo', action='store_true')
ap.add_argument('--relu-input', action='store_true', help='post-ReLU-like input (half zeros, non-negative) instead of N(0,1): the data regime of the real pipeline (matters: the MFMA kernels are power-limited)')
ap.add_argument('--stem', action='store_true', help='time the fused VGG stem kernel on 256 crops')
args = ap.parse_args()
dev = torch.device('cuda')
ops.FORCE_GENERIC_CONV = args.generic
ops.USE_HALO_3X3 = not args.no_halo

if args.stem:
    g = torch.Generator().manual_seed(0)
    ps = ops.PackedStem(torch.randn(64, 3, 3, 3, generator=g) / 27 ** 0.5, torch.zeros(64), torch.randn(64, 64, 3, 3, generator=g) / 24.0, torch.zeros(64), device=dev)
    x = torch.zeros(256, 256, 256, 8, dtype=torch.bfloat16); x[..., :3] = torch.randn(256, 256, 256, 3, generator=g).to(torch.bfloat16); x = x.to(dev)
    for _ in range(2): y = ops.vgg_stem(x, ps)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(args.iters): y = ops.vgg_stem(x, ps)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / args.iters
    print(f'vgg_stem 256 crops {ms:8.3f} ms  {ps.flops_per_pixel * 256 * 65536 / ms / 1e9:8.1f} TFLOP/s')
    sys.exit(0)
for name in args.layers.split(','):
    n, cin, h, w, cout, k, stride, pad, pool = LAYERS[name]
    g = torch.Generator().manual_seed(0)
    pc = ops.PackedConv(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5, torch.zeros(cout), stride, pad, device=dev)
    x = (torch.randn(n, h, w, pc.cin_pad, generator=g)).to(torch.bfloat16).to(dev)
    if args.relu_input:
        x = torch.relu(x)
    res = None
    if name.endswith('_res'):
        ho, wo = pc.out_hw(h, w)
        res = torch.randn(n, ho, wo, cout, generator=g).to(torch.bfloat16).to(dev)
    for _ in range(2):
        y = ops.conv2d(x, pc, act=1, pool=pool, residual=res)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        y = ops.conv2d(x, pc, act=1, pool=pool, residual=res)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    ho, wo = pc.out_hw(h, w)
    flops = 2.0 * n * ho * wo * cout * k * k * cin
    print(f'{name:12s} {ops.ConvProfile.variant(pc, n * ho * wo):34s} {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s  in {x.numel() * 2 / 1e6:7.1f} MB out {y.numel() * 2 / 1e6:7.1f} MB')
