"""dev tool: detector pass time with parts of the schedule removed (upper bounds of what work on them can gain).
usage: abl_detector_parts.py <images> <dpi>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import synthetic, ops
n, dpi = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device('cuda')
imgs = [synthetic.shelf_image(i, 2048, 2048).to(dev) for i in range(n)]


def timed(tag, patch):
    det = synthetic.synthetic_gln(seed=0, detections_per_img=dpi).to(dev)
    eng = det.engine()
    patch(eng)
    for _ in range(4):
        eng.detect(imgs, 1, dpi)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        eng.detect(imgs, 1, dpi)
    e1.record(); torch.cuda.synchronize()
    print(f'{tag:34s} {e0.elapsed_time(e1) / 20:.3f} ms per pass', flush=True)


def no_gauss(eng):
    buf = {}
    def g(c2, p3):
        k = c2.shape
        if k not in buf:
            buf[k] = torch.zeros((c2.shape[0], c2.shape[1] * 2, c2.shape[2] * 2, 1), dtype=torch.float32, device=dev)
        return buf[k]
    eng.gaussian_branch = g


def no_subnet(eng):
    orig = eng.gaussian_branch
    buf = {}
    def g(c2, p3):
        x = ops.conv2d(c2, eng.g_lateral, residual=p3, res_mode=2)
        x = ops.conv2d(x, eng.g_block1, act=1)
        x = ops.conv2d(x, eng.g_block2, act=1)
        k = c2.shape
        if k not in buf:
            buf[k] = torch.zeros((c2.shape[0], c2.shape[1] * 2, c2.shape[2] * 2, 1), dtype=torch.float32, device=dev)
        return buf[k]
    eng.gaussian_branch = g


def no_post(eng):
    orig = eng.postprocess
    cache = {}
    def p(cls, reg, *a):
        if 'o' not in cache:
            cache['o'] = orig(cls, reg, *a)
        return cache['o']
    eng.postprocess = p


for rep in range(2):
    timed('full', lambda e: None)
    timed('without the Gaussian branch', no_gauss)
    timed('without the Gaussian subnet', no_subnet)
    timed('without post-processing', no_post)
