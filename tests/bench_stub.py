"""CPU stand-in for bench.py's `build_pipeline` (selected with CVPCE_BENCH_STUB=bench_stub:build; tests/test_dist_cpu.py): the same keys,
the same sharding and the same ONE gallery all_gather through cvpce_amd.dist, with toy arithmetic in place of the HIP kernels, so that
bench.py's main() can be driven end to end under N gloo ranks on a box without a GPU.  Test infrastructure: nothing here is product code."""
import time

import torch

D = 16


def _embed(x):
    """(b,3,h,w) -> (b,D): deterministic per image, independent of the batch it is in."""
    p = torch.nn.functional.adaptive_avg_pool2d(x, (2, 2)).flatten(1)          # 12 values
    return torch.cat((p, x.amax(dim=(1, 2, 3))[:, None], x.amin(dim=(1, 2, 3))[:, None], x.mean(dim=(1, 2, 3))[:, None],
                      x.std(dim=(1, 2, 3))[:, None]), dim=1)


def _image(g, size):
    return torch.rand(3, size, size, generator=torch.Generator().manual_seed(g))


class StubPipeline:
    def __init__(self, gallery, dpi):
        self.gallery, self.dpi = gallery, dpi

    def run(self, images, stage_events=None, proposals=None):
        n, dpi = len(images), self.dpi
        emb = _embed(torch.stack(images))
        idx = torch.cdist(emb, self.gallery).argmin(dim=1)
        boxes = torch.zeros(n, dpi, 4)
        scores = torch.zeros(n, dpi)
        for i, img in enumerate(images):                      # per-image results that depend on that image only
            boxes[i, :, 2:] = img[0, :dpi, :2] * 100 + 1
            scores[i] = img[1, 0, :dpi].sort(descending=True).values
        counts = [dpi // 2 + i % 3 for i in range(n)]
        return {'boxes': boxes, 'scores': scores, 'labels': torch.zeros(n, dpi, dtype=torch.int64), 'det_count': torch.full((n,), dpi),
                'count': torch.tensor(counts), 'indices': idx[:, None, None].expand(n, dpi, 1).contiguous(), 'counts_host': counts,
                'embeddings': emb}


def build(args, rank, world, dev, ipg, dpi):
    from cvpce_amd import dist as cdist
    size = min(args.image_size, 64)
    s, e = cdist.shard_range(args.gallery, rank, world)
    t0 = time.perf_counter()
    local = _embed(torch.stack([_image(100000 + i, 16) for i in range(s, e)])) if e > s else torch.empty(0, D)
    gallery = cdist.all_gather_rows(local, args.gallery, rank, world)
    ids = cdist.shard_images(world * ipg, rank, world)
    host = [_image(g, size) for g in ids]
    return {'pipe': StubPipeline(gallery, min(dpi, size)), 'images': host, 'host_images': host, 'ids': ids, 'gallery': gallery,
            't_gallery': time.perf_counter() - t0,
            'rank0_leg': lambda: time.sleep(3.0)}     # stands for rank 0's long side legs: the other ranks must be able to finish meanwhile
