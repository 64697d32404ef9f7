#!/usr/bin/env python3
"""Headline benchmark: shelf images/sec end-to-end (detect + RoI-crop + embed + match) on MI355X.

Contract: python bench.py --gpus N --steps K --warmup W   (N>1: launched by torch.distributed.run, one
rank per GPU over RCCL).  One "step" = one pass of the whole hot path over one batch of
`--images-per-gpu` synthetic SKU-110K-shaped shelf images per GPU (weak scaling), inputs resident in
HBM when the timed region starts.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--images-per-gpu', type=int, default=8)
    ap.add_argument('--image-size', type=int, default=2048)
    ap.add_argument('--gallery', type=int, default=3200)
    ap.add_argument('--detections-per-img', type=int, default=200)
    ap.add_argument('--match-dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-peaks', action='store_true', help='skip the measured-peak microbenchmarks (library GEMM, device copy)')
    return ap.parse_args()


def cpu_baseline(det_sd, enc_sd, dpi, gallery_emb, image_size):
    """The oracle (a port: plain fp32 torch CPU ops restating the reference) on a bounded sample of the same workload,
    on this box's host cores (about 10-30 s of CPU work): 3 shelf images through the detector, 64 crops through
    crop+embed, one image's worth of queries through the reference's literal matcher in batches of 32 (production.py's
    batch loop); extrapolated to one image with P = dpi proposals."""
    from oracle import gln as og, macvgg as ovgg, crop as ocrop, match as omatch
    from cvpce_amd import synthetic
    cores = min(16, os.cpu_count() or 1)   # the GPU box's CPU share for one GPU; more threads only oversubscribe
    torch.set_num_threads(cores)
    n_img, n_crop = 3, 64
    imgs = [synthetic.shelf_image(s, image_size, image_size) for s in range(n_img)]
    t = time.perf_counter()
    res = [og.gln_forward([i], det_sd, detections_per_img=dpi)[0] for i in imgs]
    t_det = (time.perf_counter() - t) / n_img
    boxes = res[0]['boxes'][res[0]['scores'] > 0.5][:n_crop]
    if len(boxes) < n_crop:
        boxes = torch.tensor([[10., 10., 300., 400.]] * n_crop)
    t = time.perf_counter()
    embs = []
    for i in range(0, n_crop, 32):
        crops = ocrop.crop_boxes(imgs[0], boxes[i:i + 32])
        embs.append(ovgg.macvgg_forward(ocrop.scale_to_tanh(crops), enc_sd))
    t_embed = (time.perf_counter() - t) / n_crop
    q = torch.cat(embs)[:32]
    n_batches = (dpi + 31) // 32
    t = time.perf_counter()
    for _ in range(n_batches):
        omatch.nearest_neighbors_literal(gallery_emb, q, 1)
    t_match = time.perf_counter() - t
    per_image = t_det + t_embed * dpi + t_match
    return {'value': 1.0 / per_image, 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': f'oracle (fp32 torch CPU restatement): detector {n_img} images {image_size}x{image_size} at {t_det:.2f} s/image + '
                      f'crop+embed {n_crop} crops at {t_embed * 1e3:.0f} ms/crop + literal matcher {n_batches} x 32 queries x '
                      f'{len(gallery_emb)} gallery {t_match:.2f} s; extrapolated to P={dpi} proposals/image'}


def measured_peaks(dev):
    """the nominal gfx950 peaks re-measured on this box, as calibration beside the nominal figures --
    a bf16 library GEMM (torch.matmul -> hipBLASLt; 8192^3, under the same power limit as the kernels) and a device-to-device
    copy of 2 GiB (read + write bytes / time)."""
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        a @ b
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        a @ b
    e1.record(); torch.cuda.synchronize()
    gemm = 20 * 2 * 8192 ** 3 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    src = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    dst.copy_(src)
    e0.record()
    for _ in range(5):
        dst.copy_(src)
    e1.record(); torch.cuda.synchronize()
    copy = 5 * 2 * (2 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9
    return {'library_gemm_bf16_tflops': round(gemm, 1), 'device_copy_gbs': round(copy, 1),
            'nominal': {'mfma_bf16_dense_tflops': MFMA_BF16_DENSE_PEAK_TFLOPS, 'hbm_gbs': 8000.0}}


# algorithmic work per stage and image: detector 298.4 GFLOP at 800x800, embed 40.09 GFLOP per crop,
# match 2 P G D
def stage_gflop(stage, n_img, proposals, gallery):
    return {'detect': 298.4 * n_img, 'crop': 0.0, 'embed': 40.09 * proposals * n_img,
            'match': 2.0 * proposals * n_img * gallery * 1024 / 1e9}[stage]


def main():
    args = parse()
    from cvpce_amd import dist as cdist
    rank, local_rank, world = cdist.init()
    if world != args.gpus and world > 1:
        args.gpus = world
    dev = torch.device('cuda', local_rank % max(1, torch.cuda.device_count()))   # (a 1-GPU rehearsal may stack ranks on cuda:0)
    torch.cuda.set_device(dev)
    from cvpce_amd import ops, production, synthetic

    dpi = args.detections_per_img
    det = synthetic.synthetic_gln(seed=0, detections_per_img=dpi)
    enc = synthetic.synthetic_macvgg(seed=1)
    det_sd = {k: v.clone() for k, v in det.state_dict().items()} if rank == 0 else None
    enc_sd = {k: v.clone() for k, v in enc.state_dict().items()} if rank == 0 else None
    det, enc = det.to(dev), enc.to(dev)

    # gallery: embedded sharded (G/world rows per rank), assembled by ONE all_gather over RCCL/xGMI
    gal_imgs = synthetic.gallery_images(args.gallery, seed=100)
    t0 = time.perf_counter()
    embed = lambda x: enc(x.to(dev))
    s, e = cdist.shard_range(args.gallery, rank, world)
    local = torch.cat([embed(gal_imgs[i:min(i + 128, e)]) for i in range(s, e, 128)]) if e > s else torch.empty(0, 1024, device=dev)
    gallery = cdist.all_gather_rows(local, args.gallery, rank, world)
    torch.cuda.synchronize()
    t_gallery = time.perf_counter() - t0
    del gal_imgs
    mdt = torch.bfloat16 if args.match_dtype == 'bf16' else torch.float32
    clf = production.Classifier.from_embedding(enc, gallery, [f'sku_{i:05d}' for i in range(args.gallery)],
                                               device=dev, emb_device=dev, k=1, match_dtype=mdt)
    pipe = production.BatchedPipeline(det, clf, 0.5)

    images = [synthetic.shelf_image(1000 * rank + i, args.image_size, args.image_size).to(dev)
              for i in range(args.images_per_gpu)]
    torch.cuda.synchronize()

    out = None
    for _ in range(args.warmup):
        out = pipe.run(images)
    torch.cuda.synchronize()
    cdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = pipe.run(images)
    torch.cuda.synchronize()
    cdist.barrier()
    elapsed = cdist.max_over_ranks(time.perf_counter() - t0, dev)
    proposals = float(sum(out['counts_host'])) / max(1, len(images))

    roofline = None
    if not args.no_roofline and rank == 0:
        ops.PROFILE = ops.ConvProfile()
        for _ in range(args.steps):
            pipe.run(images)
        summ = ops.PROFILE.summary()
        ops.PROFILE = None
        stage_events = []                 # a separate pass: the per-launch events above slow the 140 small detector launches
        for _ in range(args.steps):
            pipe.run(images, stage_events)
        torch.cuda.synchronize()
        stages = {}
        for nm in ('detect', 'crop', 'embed', 'match'):
            ms = sum(a.elapsed_time(b) for n_, a, b in stage_events if n_ == nm) / args.steps
            gf = stage_gflop(nm, len(images), proposals, args.gallery)
            stages[nm] = {'ms_per_step': round(ms, 3), 'algorithmic_gflop': round(gf, 1),
                          'tflops': round(gf / ms, 1) if ms > 0 else None,
                          'frac_of_mfma_peak': round(gf / ms / MFMA_BF16_DENSE_PEAK_TFLOPS, 4) if ms > 0 else None}
        name, dom = max(summ.items(), key=lambda kv: kv[1]['ms'])
        achieved = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
        # HBM bytes per launch of that kernel: PMC counters cannot be read in-process, so this is the figure measured
        # by the same command under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (two passes, gfx950 FETCH x2
        # correction) and committed under profiles/; null if no profile covers the kernel
        traffic = None
        try:
            prof = json.load(open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json')))
            v = prof.get(name)            # keyed by the names ops.ConvProfile uses (tools/summarise_profiles.py)
            if v is not None:
                traffic = round((v['read_bytes_per_launch'] + v['write_bytes_per_launch']) / 1e9, 4)
        except Exception:
            traffic = None
        roofline = {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': MFMA_BF16_DENSE_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(achieved / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), 'traffic': traffic, 'traffic_unit': 'GB/launch (rocprofv3 PMC, profiles/)',
                    'kernel': name, 'launches': dom['launches'], 'avg_launch_us': round(dom['ms'] * 1e3 / dom['launches'], 2),
                    'share_of_conv_time': round(dom['ms'] / sum(v['ms'] for v in summ.values()), 4),
                    'stages': stages,
                    'all_conv_kernels': {k: {'launches': v['launches'], 'ms': round(v['ms'], 3),
                                             'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2)} for k, v in summ.items()}}

    peaks = measured_peaks(dev) if (rank == 0 and not args.no_peaks and not args.no_roofline) else None
    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        cpu = cpu_baseline(det_sd, enc_sd, dpi, gallery.float().cpu(), args.image_size)

    if rank == 0:
        total_images = world * args.images_per_gpu * args.steps
        line = {
            'metric': 'shelf images/sec end-to-end (detect+embed+match)',
            'value': round(total_images / elapsed, 3), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': f'full production path: {args.images_per_gpu} shelf images/GPU of 3x{args.image_size}x{args.image_size} '
                                   f'(SKU-110K shape) -> GLN detect (800x800 internal, detections_per_img={dpi}, conf>0.5) -> RoI crop 256x256 '
                                   f'-> MAC-VGG16 embed -> cosine NN match, gallery={args.gallery}x1024 (BASELINE configs[2]/[4] per-GPU shape)',
                       'images_per_gpu': args.images_per_gpu, 'proposals_per_image': proposals, 'gallery': args.gallery,
                       'match_dtype': args.match_dtype, 'weights': 'seeded random init, cls head calibrated (cvpce_amd/synthetic.py)',
                       'parallelism': f'dp{world} (images sharded, gallery embedded sharded + 1 all_gather, no steady-state collectives)',
                       'gallery_build_s': round(t_gallery, 3)},
        }
        if roofline is not None:
            line['roofline'] = roofline
        if peaks is not None:
            line['measured_peaks'] = peaks
        if cpu is not None:
            line['cpu_baseline'] = cpu
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
