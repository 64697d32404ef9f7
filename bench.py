#!/usr/bin/env python3
"""Headline benchmark: shelf images/sec end-to-end (detect + RoI-crop + embed + match) on MI355X.

Contract: python bench.py --gpus N --steps K --warmup W   (N>1: launched by torch.distributed.run, one rank per GPU over
RCCL; a plain `python bench.py --gpus N` starts that launcher itself as a child process).  One "step" = one pass of the
whole hot path over one batch of `--images-per-gpu` synthetic SKU-110K-shaped shelf images per GPU (weak scaling), inputs
resident in HBM when the timed region starts.  Rank 0 prints ONE JSON line.

Workloads (`--workload`):
  pipeline      (default) BASELINE configs[2]/[4] per-GPU shape: 8 x 3x2048x2048 -> detect -> crop -> embed -> match, G = 3200
  detector      BASELINE configs[1]: 4 x 3x2048x2048, GLN detector only, detections_per_img = 1000
  match-stress  BASELINE configs[3]: 200 (and 1600) queries x 10 000 gallery rows, D = 512 and 1024, bf16 distance GEMM
Extra legs of the pipeline workload (all outside the timed region of `value`):
  roofline      per-kernel HIP-event timing of the conv kernels + per-stage times
  value_with_h2d  the same K steps with the images uploaded from pinned host memory on a copy stream, batch i+1 during batch i
  value_lists_off / value_planted_boxes / value_fitted_scenes_p200   the same whole pipeline with the constant-padding work lists off,
                on 8 x 200 PLANTED proposals of 60-250 px (seed 7) and on fitted-scene proposals padded to P = 200: how
                much of `value` is a property of the random-weight detector's box shapes
  parity        tests/accuracy.py on a bounded sample: HIP pipeline (product defaults) vs the fp32 oracle (AP / AR300 / top-1)
  cpu_baseline  the oracle timed on this box's host cores
  --verify      per-image SHA-256 of (boxes, scores, matched indices) gathered to rank 0: identical for 1/2/4/8-GPU runs
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
MFMA_PEAK_CLOCK_MHZ = 2400.0           # the engine clock that peak is quoted at (same table)
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='pipeline', choices=['pipeline', 'detector', 'match-stress'])
    ap.add_argument('--images-per-gpu', type=int, default=None, help='default 8 (pipeline) / 4 (detector)')
    ap.add_argument('--image-size', type=int, default=2048)
    ap.add_argument('--gallery', type=int, default=3200)
    ap.add_argument('--detections-per-img', type=int, default=None, help='default 200 (pipeline, cli/eval.py:50) / 1000 (detector)')
    ap.add_argument('--match-dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--detector-precision', default='fp16', choices=['bf16', 'fp16'],
                    help="storage type of the detector's weights / activations: fp16 (the product default: the mode that meets north_star's 0.1 pt "
                         "tolerance, same MFMA rate) or bf16 (opt-in; the embedder and the matcher compute in bf16 in both)")
    ap.add_argument('--windows', type=int, default=3, help='timed windows of --steps steps each; value = the median window')
    ap.add_argument('--no-workloads', action='store_true', help='skip the configs[1] / configs[3] figures appended to the pipeline line')
    ap.add_argument('--verify', action='store_true', help='gather per-image result digests to rank 0 (8e: identical across world sizes)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-clocks', action='store_true', help='skip the clock / power sampling leg (amdgpu hwmon, 3 s)')
    ap.add_argument('--no-peaks', action='store_true', help='skip the measured-peak microbenchmarks (library GEMM, device copy, bare MFMA loop)')
    ap.add_argument('--no-parity', action='store_true', help='skip the bounded HIP-vs-oracle accuracy sample')
    ap.add_argument('--no-h2d', action='store_true', help='skip the H2D-inclusive leg')
    ap.add_argument('--no-precision-leg', action='store_true', help='skip the extra timed window in the other detector precision')
    ap.add_argument('--no-coheadlines', action='store_true', help='skip the lists-off / planted-box whole-pipeline windows (value_lists_off, value_planted_boxes)')
    ap.add_argument('--allow-stub', action='store_true', help='tests only: honour CVPCE_BENCH_STUB (a CPU stand-in for the HIP pipeline; the line is stamped "stub")')
    ap.add_argument('--details', default=None, help='where rank 0 writes the FULL result object (per-layer tables, stages, every co-headline object); '
                                                    'default bench_details.json beside bench.py.  The printed line is the compact form (< 4 KB) and names this file')
    return ap.parse_args()


def spawn_launcher(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the launcher as a CHILD process (never exec from a
    process that may touch the GPU) and relay its output; nothing here has initialised HIP yet."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.run(cmd, env=env).returncode


# ---------------------------------------------------------------------------------------------------------------------
# legs shared by the workloads
# ---------------------------------------------------------------------------------------------------------------------
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
    if enc_sd is None:
        return {'value': 1.0 / t_det, 'unit': 'images/s', 'cores': cores, 'kind': 'port',
                'sample': f'oracle (fp32 torch CPU restatement) detector only: {n_img} images {image_size}x{image_size} at {t_det:.2f} s/image'}
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
            'sample_short': f'oracle fp32 torch CPU: {n_img} img detect {t_det:.2f} s/img + {n_crop} crops {t_embed * 1e3:.0f} ms/crop + literal match; per image P={dpi}',
            'sample': f'oracle (fp32 torch CPU restatement): detector {n_img} images {image_size}x{image_size} at {t_det:.2f} s/image + '
                      f'crop+embed {n_crop} crops at {t_embed * 1e3:.0f} ms/crop + literal matcher {n_batches} x 32 queries x '
                      f'{len(gallery_emb)} gallery {t_match:.2f} s; extrapolated to P={dpi} proposals/image'}


def measured_peaks(dev):
    """the nominal gfx950 peaks re-measured on this box, as calibration beside the nominal figures --
    a bf16 library GEMM (torch.matmul -> hipBLASLt; 8192^3, under the same power limit as the kernels), a device-to-device
    copy of 2 GiB (read + write bytes / time), and the bare-MFMA-loop ceiling (csrc/probe.hip: register operands, random
    data, >= 2 s per MFMA shape) that bounds what any bf16 MFMA kernel can sustain on this device."""
    from cvpce_amd import ops
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
    del src, dst
    m32 = ops.probe_mfma_bf16(0, seconds=2.0)
    m16 = ops.probe_mfma_bf16(1, seconds=2.0)
    l2 = {f'{mib}MiB': round(ops.probe_l2_stream(mib, seconds=0.5), 2) for mib in (1, 2, 16)}
    return {'library_gemm_bf16_tflops': round(gemm, 1), 'device_copy_gbs': round(copy, 1),
            'l2_stream_to_registers_tbs': dict(l2, note='every workgroup walks the same buffer with 16-byte buffer loads (csrc/probe.hip): '
                                                        '1-2 MiB = a layer\'s weights resident in each XCD\'s L2, 16 MiB = beyond it'),
            'bare_mfma_loop_tflops': {'32x32x16': round(m32, 1), '16x16x32': round(m16, 1),
                                      'note': 'register operands, random data, one wave per SIMD, >= 2 s back to back (csrc/probe.hip)'},
            'nominal': {'mfma_bf16_dense_tflops': MFMA_BF16_DENSE_PEAK_TFLOPS, 'hbm_gbs': HBM_PEAK_GBS}}


class ClockSampler:
    """Shader clock and package power of the card while a leg runs, read from the amdgpu hwmon files (read-only sysfs; a host thread,
    20 ms period).  MI355X runs this workload AT its power cap: the clock the MFMA peak is quoted at (2.4 GHz) is not the clock
    the kernels get, and the median measured here prices `roofline.achieved` against the peak at the clock actually delivered."""

    def __init__(self, dev, period=0.02):
        self.period, self.samples, self._stop, self._thread = period, [], threading.Event(), None
        self.freq = self.power = None
        import glob
        pr = torch.cuda.get_device_properties(dev)         # the card of THIS process: a host has one drm node per partition
        want = f'{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.'
        for d in sorted(glob.glob('/sys/class/drm/card[0-9]*/device')):
            if not os.path.basename(os.path.realpath(d)).startswith(want):
                continue
            for h in glob.glob(d + '/hwmon/hwmon*'):
                if os.path.exists(h + '/freq1_input'):
                    self.freq = h + '/freq1_input'                      # Hz
                for nm in ('power1_input', 'power1_average'):
                    if self.power is None and os.path.exists(h + '/' + nm):
                        self.power = h + '/' + nm                       # microwatt

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return f.read().strip()
        except OSError:
            return None

    def _loop(self):
        while not self._stop.is_set():
            f, p = self._read(self.freq) if self.freq else None, self._read(self.power) if self.power else None
            try:
                self.samples.append((int(f) / 1e6 if f else None, int(p) / 1e6 if p else None))
            except ValueError:
                pass
            time.sleep(self.period)

    def __enter__(self):
        self.samples, self._stop = [], threading.Event()
        self._thread = threading.Thread(target=self._loop, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join()

    def summary(self):
        def med(xs):
            xs = sorted(x for x in xs if x is not None)
            return round(xs[len(xs) // 2], 1) if xs else None
        return {'sclk_mhz_median': med([a for a, _ in self.samples]), 'power_w_median': med([b for _, b in self.samples]),
                'samples': len(self.samples)}


# algorithmic work per stage and image: detector 298.4 GFLOP at 800x800, embed 40.09 GFLOP per crop,
# match 2 P G D
def stage_gflop(stage, n_img, proposals, gallery):
    return {'detect': 298.4 * n_img, 'crop': 0.0, 'embed': 40.09 * proposals * n_img,
            'match': 2.0 * proposals * n_img * gallery * 1024 / 1e9}[stage]


def conv_roofline(summ, stages=None):
    """`roofline` object from an ops.ConvProfile summary: the dominant kernel against the dense bf16 MFMA peak."""
    name, dom = max(summ.items(), key=lambda kv: kv[1]['ms'])
    # EXECUTED FLOPs over time: the embedder's work-list launches skip the tiles that lie in a crop's constant padding
    # (csrc/skiplist.hip), so the algorithmic FLOPs of a layer are not all performed -- the roofline fraction prices what ran
    achieved = dom.get('flops_executed', dom['flops']) / (dom['ms'] * 1e-3) / 1e12
    # HBM bytes per launch of that kernel: PMC counters cannot be read in-process, so this is the figure measured
    # by the same command under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (two passes, gfx950 FETCH x2
    # correction) and committed under profiles/; null if no profile covers the kernel
    prof, traffic_note = _pmc_traffic()     # (stamped with the sha256 of the library it was collected on: another build's figures are not reported)
    v = prof.get(name) if prof else None    # keyed by the names ops.ConvProfile uses (tools/summarise_profiles.py)
    traffic = round((v['read_bytes_per_launch'] + v['write_bytes_per_launch']) / 1e9, 4) if v is not None else None
    out = {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': MFMA_BF16_DENSE_PEAK_TFLOPS, 'unit': 'TFLOP/s',
           'frac': round(achieved / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), 'traffic': traffic, 'traffic_unit': 'GB/launch (rocprofv3 PMC, profiles/)',
           'kernel': name, 'launches': dom['launches'], 'avg_launch_us': round(dom['ms'] * 1e3 / dom['launches'], 2),
           'share_of_conv_time': round(dom['ms'] / sum(v['ms'] for v in summ.values()), 4),
           'flops': 'EXECUTED (tiles on the work lists; the skipped constant-padding tiles are not counted)',
           # the layers' ALGORITHMIC FLOPs over the same time: a throughput in units of whole crops, NOT a hardware rate (it can exceed
           # the MFMA peak where padding tiles are skipped) -- never compare it with `peak`
           'images_equivalent_tflops': round(dom['flops'] / (dom['ms'] * 1e-3) / 1e12, 2)}
    st = summ.get(name + '[strips]')
    if st and st['ms'] > 0:
        # the same kernel's STRIP template instance (tiles with 4 useful rows, three per pass): its own launches, duration and executed
        # FLOPs -- never averaged with the LIST launches above (round 5 published a 970 us mean of the two populations)
        out['strip_launches'] = {'launches': st['launches'], 'avg_launch_us': round(st['ms'] * 1e3 / st['launches'], 2), 'total_ms': round(st['ms'], 3),
                                 'executed_tflop': round(st['flops_executed'] / 1e12, 3), 'achieved': round(st['flops_executed'] / (st['ms'] * 1e-3) / 1e12, 2),
                                 'frac': round(st['flops_executed'] / (st['ms'] * 1e-3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS, 4)}
    out['total_ms'] = round(dom['ms'], 3)
    out['executed_tflop'] = round(dom.get('flops_executed', dom['flops']) / 1e12, 3)
    if traffic_note:
        out['traffic_note'] = traffic_note
    if stages is not None:
        out['stages'] = stages
    out['all_conv_kernels'] = {k: {'launches': v['launches'], 'ms': round(v['ms'], 3),
                                   'tflops': round(v.get('flops_executed', v['flops']) / (v['ms'] * 1e-3) / 1e12, 2),
                                   'frac_of_mfma_peak': round(v.get('flops_executed', v['flops']) / (v['ms'] * 1e-3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
                                   'images_equivalent_tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2)} for k, v in summ.items()}
    return out


def _pmc_traffic():
    """profiles/hbm_traffic.json (per-launch HBM bytes by kernel, rocprofv3 PMC passes summarised by tools/summarise_profiles.py), or
    (None, why) when it was collected on another build of the library than the one loaded now."""
    try:
        prof = json.load(open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json')))
        from cvpce_amd import _lib
        sha = hashlib.sha256(open(_lib.LIB_PATH, 'rb').read()).hexdigest()
        if prof.get('_library_sha256') != sha:
            return None, 'profiles/hbm_traffic.json was collected on another build of libcvpce_hip.so: not reported'
        return prof, None
    except Exception as e:                                   # noqa: BLE001 (a missing / unreadable file is "no figure", never a failure)
        return None, f'profiles/hbm_traffic.json unreadable: {type(e).__name__}'


def hbm_stages(summ_bytes, steps, copy_gbs=None):
    """`roofline.hbm_stages`: the HBM-bound stages of a step (input transform, RoI crop, the detector's 1x1 convs, the Gaussian
    subnet's thin 3x3 convs and its 1x1 tail) against the HBM roofline: ALGORITHMIC bytes per step (every input / output tensor of a
    launch once, weights included; the crop: every source pixel of every box once + the crops written), HIP-event time, GB/s, the
    fraction of the nominal 8 TB/s -- and, where profiles/hbm_traffic.json covers the kernel on THIS build, the measured PMC bytes
    (FETCH_SIZE x 2 + WRITE_SIZE, separate passes) and their ratio to the algorithmic bytes (> 1.3 is flagged: wasted re-reads)."""
    pmc, why = _pmc_traffic()
    out = {}
    for name, d in summ_bytes.items():
        ms, gb = d['ms'] / steps, d['bytes'] / steps / 1e9
        e = {'launches_per_step': round(d['launches'] / steps, 1), 'algorithmic_gb_per_step': round(gb, 4), 'ms_per_step': round(ms, 4),
             'gbs': round(gb / ms * 1e3, 1) if ms > 0 else None, 'frac_of_hbm_peak': round(gb / ms * 1e3 / HBM_PEAK_GBS, 4) if ms > 0 else None}
        v = pmc.get(name) if pmc else None
        if v is not None and v.get('launches'):
            per_launch = v['read_bytes_per_launch'] + v['write_bytes_per_launch']
            e['pmc_gb_per_launch'] = round(per_launch / 1e9, 4)
            e['algorithmic_gb_per_launch'] = round(d['bytes'] / d['launches'] / 1e9, 4)
            e['pmc_over_algorithmic'] = round(per_launch / (d['bytes'] / d['launches']), 3)
            e['pmc_gbs'] = round(per_launch / 1e9 / (d['ms'] / d['launches']) * 1e3, 1)
            e['flag_wasted_traffic'] = bool(e['pmc_over_algorithmic'] > 1.3)
        out[name] = e
    if why:
        out['_pmc_note'] = why
    out['_note'] = ('bound = hbm for every stage here; peak = 8000 GB/s nominal (MI355X_MICROARCH.md); measured device copy rate: '
                    'measured_peaks.device_copy_gbs')
    return out


def image_digest(out, i):
    """SHA-256 over everything the path returns for image i of a BatchedPipeline result (bit-level identity check)."""
    c, dc = int(out['counts_host'][i]), int(out['det_count'][i])
    h = hashlib.sha256()
    for t in (out['boxes'][i, :dc], out['scores'][i, :dc], out['labels'][i, :dc], out['indices'][i, :c]):
        h.update(t.contiguous().cpu().numpy().tobytes())
    return h.hexdigest()


def gather_digests(local, world):
    """{global image id: digest} from every rank -> rank 0 (the optional final gather of SURVEY.md 8e)."""
    if world == 1:
        return dict(local)
    import torch.distributed as dist
    parts = [None] * world
    dist.all_gather_object(parts, local)
    merged = {}
    for p in parts:
        merged.update(dict(p))
    return merged


def _sync(dev):
    if dev.type == 'cuda':
        torch.cuda.synchronize()


def timed_windows(step, steps, windows, dev, collective=True):
    """`windows` back-to-back timed regions of EXACTLY `steps` calls of `step()`, each bracketed by barrier +
    torch.cuda.synchronize() on both sides and reduced with MAX over ranks.  -> (median seconds, [seconds per window]).
    One window of ~1 s decides little on a power-limited part with +-3 % box-to-box spread; the line reports the median
    and the spread.  collective=False: a rank-0-only side leg (no barrier, no reduction: the other ranks are not there)."""
    from cvpce_amd import dist as cdist
    secs = []
    for _ in range(max(1, windows)):
        _sync(dev)
        if collective:
            cdist.barrier()
            _sync(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        _sync(dev)
        if collective:
            cdist.barrier()
        dt = time.perf_counter() - t0
        secs.append(cdist.max_over_ranks(dt, dev) if collective else dt)
    return sorted(secs)[len(secs) // 2], secs


def window_stats(secs, steps):
    ms = [round(t / steps * 1e3, 3) for t in secs]
    return {'n': len(ms), 'steps_each': steps, 'ms_per_step': ms, 'min': min(ms), 'max': max(ms), 'value': 'median window'}


# ---------------------------------------------------------------------------------------------------------------------
# workload: full pipeline (the headline metric)
# ---------------------------------------------------------------------------------------------------------------------
def run_h2d_leg(pipe, host_images, dev, steps, warmup):
    """The same pipeline fed from pinned host memory: a copy stream uploads batch i+1 while batch i runs (the reference
    moves every image with `.to(device)`, production.py:14).  -> seconds for `steps` steps, each including one upload."""
    main = torch.cuda.current_stream()
    copy_stream = torch.cuda.Stream(device=dev)
    bufs = [[torch.empty(h.shape, dtype=h.dtype, device=dev) for h in host_images] for _ in range(2)]
    ready = [torch.cuda.Event() for _ in range(2)]
    free = [torch.cuda.Event() for _ in range(2)]

    def upload(b):
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(free[b])            # the step that last read this buffer has finished
            for d, h in zip(bufs[b], host_images):
                d.copy_(h, non_blocking=True)
            ready[b].record(copy_stream)

    def step(s):
        b = s % 2
        upload(1 - b)                                   # next batch: overlaps with this step's kernels
        main.wait_event(ready[b])
        pipe.run(bufs[b])
        free[b].record(main)

    upload(0)
    for s in range(warmup):
        step(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(warmup, warmup + steps):
        step(s)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def planted_proposals(n_images, dpi, image_size, dev, seed=7):
    """PLANTED proposals -- exactly `dpi` boxes per image, both sides uniform in 60-250 px and independent, uniformly
    placed on the image_size^2 canvas, seed 7.  -> (boxes (N,dpi,4) f32, counts (N,) int32) on `dev` (BatchedPipeline.run(proposals=...))."""
    g = torch.Generator().manual_seed(seed)
    wh = 60 + 190 * torch.rand(n_images * dpi, 2, generator=g)
    xy = torch.rand(n_images * dpi, 2, generator=g) * (float(image_size) - wh)
    boxes = torch.cat((xy, xy + wh), dim=1).view(n_images, dpi, 4)
    return boxes.to(dev), torch.full((n_images,), dpi, dtype=torch.int32, device=dev)


def crop_shape_stats(boxes):
    b = boxes.reshape(-1, 4).to(torch.long).float().cpu()
    bw, bh = (b[:, 2] - b[:, 0]).clamp(min=1), (b[:, 3] - b[:, 1]).clamp(min=1)
    return {'short_over_long_mean': round(float((torch.minimum(bw, bh) / torch.maximum(bw, bh)).mean()), 4),
            'wide_fraction': round(float((bw > bh).float().mean()), 4)}


def executed_fraction(run_once):
    """One profiled pass of `run_once()` -> (executed / algorithmic conv FLOPs, {kernel: executed TFLOP/s})."""
    from cvpce_amd import ops
    ops.PROFILE = ops.ConvProfile()
    try:
        run_once()
        summ = ops.PROFILE.summary()
    finally:
        ops.PROFILE = None
    alg = sum(v['flops'] for v in summ.values())
    exe = sum(v.get('flops_executed', v['flops']) for v in summ.values())
    return round(exe / max(alg, 1.0), 4), summ


def coheadline_legs(pipe, images, ipg, dpi, image_size, steps, dev):
    """The WHOLE pipeline (detect + crop + embed + match, same images, same gallery, same kernels) timed twice more on rank 0, one window
    of `steps` steps each: (a) with the constant-padding work lists OFF -- every tile of every crop computed, the figure that does not
    depend on the boxes' shapes; (b) on PLANTED proposals, exactly `dpi` per image, 60-250 px with independent sides --
    a realistic spread of aspect ratios instead of the random-weight detector's uniform 2.6 : 1 boxes.  Each with the fraction of
    the algorithmic conv FLOPs its launches executed.  (datautils.py:232-239 is the padding rule the lists exploit.)"""
    from cvpce_amd.models import classification as C
    out = {}
    # (a) lists off
    was = C.SKIP_PADDING
    C.SKIP_PADDING = False
    try:
        for _ in range(2):
            pipe.run(images)
        t, _ = timed_windows(lambda: pipe.run(images), steps, 3, dev, collective=False)          # (median of three windows, like `value`)
        frac, summ_off = executed_fraction(lambda: pipe.run(images))
    finally:
        C.SKIP_PADDING = was
    out['lists_off'] = {'images_per_s': round(ipg * steps / t, 3), 'ms_per_step': round(t / steps * 1e3, 3), 'executed_over_algorithmic_conv_flops': frac,
                        'what': 'CVPCE_SKIP_PADDING=0: every tile of every crop is computed'}
    dom = summ_off.get('conv3x3_halo2_kernel')
    if dom:
        ach = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
        out['lists_off']['dominant_kernel'] = {'kernel': 'conv3x3_halo2_kernel', 'achieved_tflops': round(ach, 2), 'frac': round(ach / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
                                               'avg_launch_us': round(dom['ms'] * 1e3 / dom['launches'], 2), 'launches': dom['launches']}
        out['lists_off']['all_conv_kernels'] = {k: {'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 1),
                                                     'frac_of_mfma_peak': round(v['flops'] / (v['ms'] * 1e-3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS, 4)}
                                                for k, v in summ_off.items() if k in ('conv3x3_halo2_kernel', 'conv3x3_halo3_kernel', 'vgg_stem2_kernel')}
    # (b) planted proposals through the whole pipeline
    prop = planted_proposals(len(images), dpi, image_size, dev)
    for _ in range(2):
        pipe.run(images, proposals=prop)
    t, _ = timed_windows(lambda: pipe.run(images, proposals=prop), steps, 3, dev, collective=False)
    frac, _ = executed_fraction(lambda: pipe.run(images, proposals=prop))
    out['planted_boxes'] = dict({'images_per_s': round(ipg * steps / t, 3), 'ms_per_step': round(t / steps * 1e3, 3), 'proposals_per_image': dpi,
                                 'executed_over_algorithmic_conv_flops': frac,
                                 'what': f'the detector runs, then exactly {dpi} planted boxes per image (60-250 px uniform, independent sides, seed 7) replace '
                                         'its boxes for crop + embed + match'}, **crop_shape_stats(prop[0]))
    return out


def build_pipeline(args, rank, world, dev, ipg, dpi):
    """Models, the sharded gallery build with its ONE all_gather, this rank's images -> dict(pipe, clf, enc, images, host_images, ids,
    gallery, t_gallery, det_sd, enc_sd).  (tests/test_dist_cpu.py swaps this for a CPU stand-in through CVPCE_BENCH_STUB to drive
    main() under 8 gloo ranks; nothing else of bench.py is replaced there.)"""
    from cvpce_amd import dist as cdist, production, synthetic
    det = synthetic.synthetic_gln(seed=0, detections_per_img=dpi, precision=args.detector_precision)
    enc = synthetic.synthetic_macvgg(seed=1)
    det_sd = {k: v.clone() for k, v in det.state_dict().items()} if rank == 0 else None
    enc_sd = {k: v.clone() for k, v in enc.state_dict().items()} if rank == 0 else None
    det, enc = det.to(dev), enc.to(dev)

    # gallery: embedded sharded (G/world rows per rank), assembled by ONE all_gather over RCCL/xGMI
    # (every rank materialises only ITS block of the synthetic gallery: 2.5 GB of f32 images at G = 3200 otherwise, per rank)
    s, e = cdist.shard_range(args.gallery, rank, world)
    gal_imgs = synthetic.gallery_shard(s, e, seed=100)
    t0 = time.perf_counter()
    embed = lambda x: enc(x.to(dev))
    local = torch.cat([embed(gal_imgs[i:i + 128]) for i in range(0, e - s, 128)]) if e > s else torch.empty(0, 1024, device=dev)
    gallery = cdist.all_gather_rows(local, args.gallery, rank, world)
    torch.cuda.synchronize()
    t_gallery = time.perf_counter() - t0
    del gal_imgs
    mdt = torch.bfloat16 if args.match_dtype == 'bf16' else torch.float32
    clf = production.Classifier.from_embedding(enc, gallery, [f'sku_{i:05d}' for i in range(args.gallery)],
                                               device=dev, emb_device=dev, k=1, match_dtype=mdt)
    pipe = production.BatchedPipeline(det, clf, 0.5)

    # images are identified by their GLOBAL index in the job's batch of world * ipg images (contiguous blocks per rank,
    # cvpce_amd.dist.shard_images): image g is the same tensor whatever the world size
    ids = cdist.shard_images(world * ipg, rank, world)
    host_images = [synthetic.shelf_image(g, args.image_size, args.image_size) for g in ids]
    images = [h.to(dev) for h in host_images]
    torch.cuda.synchronize()
    return {'pipe': pipe, 'clf': clf, 'enc': enc, 'images': images, 'host_images': host_images, 'ids': ids, 'gallery': gallery,
            't_gallery': t_gallery, 'det_sd': det_sd, 'enc_sd': enc_sd}


ALLOW_STUB = False      # main(): --allow-stub (tests only)


def stub_factory():
    """CVPCE_BENCH_STUB='module:function' (tests only): a CPU stand-in for `build_pipeline` with the same signature and keys, so that
    main() -- argument parsing, sharding, the gallery all_gather, the timed windows' barriers and MAX reduction, the --verify gather,
    the rank-0-only tail -- can run under N gloo ranks on a box without a GPU.  The stand-in's dict may carry 'rank0_leg', a callable
    standing for the long rank-0-only side legs.  Every GPU-only leg is off in that mode."""
    spec = os.environ.get('CVPCE_BENCH_STUB')
    if not spec:
        return None
    if not ALLOW_STUB:
        sys.exit('bench.py: CVPCE_BENCH_STUB is set but --allow-stub was not passed: refusing to replace the HIP pipeline with a stand-in')
    import importlib
    mod, fn = spec.split(':')
    return getattr(importlib.import_module(mod), fn)


def run_pipeline(args, rank, local_rank, world, dev):
    from cvpce_amd import dist as cdist
    ipg = args.images_per_gpu or 8
    dpi = args.detections_per_img or 200
    stub = stub_factory()
    leg_errors = {}

    def leg(name, fn, *a, **kw):
        """A rank-0 side leg: its failure is recorded on the line (`leg_errors`), it never takes the measurement down with it."""
        try:
            return fn(*a, **kw)
        except Exception as e:                               # noqa: BLE001
            import traceback
            leg_errors[name] = f'{type(e).__name__}: {e}'
            print(f'[bench] leg {name} failed:\n{traceback.format_exc()}', file=sys.stderr, flush=True)
            return None

    if stub is not None:
        args.no_h2d = args.no_roofline = args.no_workloads = args.no_peaks = args.no_parity = args.no_cpu_baseline = True
        args.no_precision_leg = args.no_coheadlines = True
    else:
        from cvpce_amd import ops, production, synthetic
    built = (stub or build_pipeline)(args, rank, world, dev, ipg, dpi)
    pipe, images, host_images, ids = built['pipe'], built['images'], built['host_images'], built['ids']
    clf, enc, gallery, t_gallery, det_sd, enc_sd = (built.get(k) for k in ('clf', 'enc', 'gallery', 't_gallery', 'det_sd', 'enc_sd'))

    outs = [None]

    def step():
        outs[0] = pipe.run(images)

    for _ in range(args.warmup):
        step()
    coll_before = dict(cdist.STATS)
    elapsed, window_secs = timed_windows(step, args.steps, args.windows, dev)
    coll_after = dict(cdist.STATS)
    out = outs[0]
    proposals = float(sum(out['counts_host'])) / max(1, len(images))

    # ---- the legs EVERY rank takes part in come first (their barriers must not wait for rank 0's side legs) -----------------------
    verify = None
    if args.verify:
        merged = gather_digests([(g, image_digest(out, i)) for i, g in enumerate(ids)], world)
        if rank == 0:
            allh = hashlib.sha256(''.join(merged[g] for g in sorted(merged)).encode()).hexdigest()
            verify = {'images': len(merged), 'digest': allh, 'per_image': {str(g): merged[g][:16] for g in sorted(merged)}}

    h2d = None
    if not args.no_h2d:
        pinned = [h.pin_memory() for h in host_images]
        cdist.barrier()
        t_h2d = cdist.max_over_ranks(run_h2d_leg(pipe, pinned, dev, args.steps, max(1, args.warmup)), dev)
        h2d = {'value_with_h2d': round(world * ipg * args.steps / t_h2d, 3), 'ms_per_step_with_h2d': round(t_h2d / args.steps * 1e3, 3),
               'upload_mb_per_step': round(sum(h.numel() * 4 for h in host_images) / 1e6, 1),
               'how': 'pinned host staging, uploads of batch i+1 on a copy stream during batch i (double-buffered)'}
        del pinned
    collectives = cdist.collective_stats()
    if collectives is not None:
        # what the timed windows themselves issued: per window two barriers + one MAX reduction of the window's seconds, at the window
        # boundaries -- nothing between the steps (no steady-state collectives)
        collectives['in_timed_windows'] = {k: coll_after[k] - coll_before[k] for k in ('all_gather', 'all_reduce', 'barrier')}
        # derived, not asserted: what the windows issued beyond their own boundary operations, per timed step
        nw = max(1, args.windows)
        itw = collectives['in_timed_windows']
        extra = itw['all_gather'] + max(0, itw['all_reduce'] - nw) + max(0, itw['barrier'] - 2 * nw)
        collectives['data_path_collectives_per_step'] = extra / (nw * args.steps) if extra else 0
        collectives['gallery_all_gather_mb'] = round(collectives['all_gather_bytes_received'] / 1e6, 2)

    # ---- rank-0-only side legs (no collectives below this line: the other ranks are on their way out) -----------------------------
    # the same step with the detector in its OTHER storage mode (fp16 = the product default, the mode that meets the parity tolerance,
    # DESIGN.md 2a; bf16 = the opt-in): one more timed window
    def precision_leg():
        other = 'fp16' if args.detector_precision == 'bf16' else 'bf16'
        det2 = synthetic.synthetic_gln(seed=0, detections_per_img=dpi, precision=other).to(dev)
        pipe2 = production.BatchedPipeline(det2, clf, 0.5)
        for _ in range(max(2, args.warmup)):
            pipe2.run(images)
        t2, _ = timed_windows(lambda: pipe2.run(images), args.steps, 1, dev, collective=False)
        return {args.detector_precision: round(ipg * args.steps / elapsed, 3), other: round(ipg * args.steps / t2, 3),
                'note': 'images/s per GPU with the detector storing fp16 (default) / bf16; the second figure is one extra timed window of '
                        '--steps steps on rank 0 right after the headline windows, same box, same images'}

    by_precision = leg('precision', precision_leg) if (rank == 0 and not args.no_precision_leg) else None

    if rank == 0 and built.get('rank0_leg') is not None:
        built['rank0_leg']()
    cohead = None
    if rank == 0 and not args.no_coheadlines:
        cohead = leg('coheadlines', coheadline_legs, pipe, images, ipg, dpi, args.image_size, args.steps, dev)
    if cohead is not None:
        cohead['headline'] = dict({'images_per_s': round(ipg * args.steps / elapsed, 3), 'proposals_per_image': proposals}, **crop_shape_stats(
            torch.cat([out['boxes'][i, :c] for i, c in enumerate(out['counts_host'])])))

    def roofline_leg():
        ops.PROFILE = ops.ConvProfile()
        try:
            for _ in range(args.steps):
                pipe.run(images)
            summ = ops.PROFILE.summary()
            summ_bytes = ops.PROFILE.summary_bytes()
        finally:
            ops.PROFILE = None
        alg_conv = sum(v['flops'] for v in summ.values()) / args.steps / 1e9
        exe_conv = sum(v.get('flops_executed', v['flops']) for v in summ.values()) / args.steps / 1e9
        stage_events = []                 # a separate pass: the per-launch events above slow the small detector launches
        for _ in range(args.steps):
            pipe.run(images, stage_events)
        torch.cuda.synchronize()
        stages = {}
        for nm in ('detect', 'crop', 'embed', 'match'):
            ms = sum(a.elapsed_time(b) for n_, a, b in stage_events if n_ == nm) / args.steps
            gf = stage_gflop(nm, len(images), proposals, args.gallery)
            ex = gf - (alg_conv - exe_conv) if nm == 'embed' else gf       # only the embedder's launches skip tiles
            stages[nm] = {'ms_per_step': round(ms, 3), 'algorithmic_gflop': round(gf, 1), 'executed_gflop': round(ex, 1),
                          'tflops': round(ex / ms, 1) if ms > 0 else None,
                          'frac_of_mfma_peak': round(ex / ms / MFMA_BF16_DENSE_PEAK_TFLOPS, 4) if ms > 0 else None,
                          'images_equivalent_tflops': round(gf / ms, 1) if ms > 0 else None}
        roofline = conv_roofline(summ, stages)
        roofline['hbm_stages'] = hbm_stages(summ_bytes, args.steps)
        alg_step = sum(stage_gflop(nm, len(images), proposals, args.gallery) for nm in ('detect', 'embed', 'match'))
        roofline['gflop_per_step'] = {
            'algorithmic': round(alg_step, 1), 'executed': round(alg_step - (alg_conv - exe_conv), 1),
            'note': 'algorithmic = SURVEY.md 8(d) (298.4 GFLOP detector + 40.09 GFLOP per crop + matcher); executed = without the embedder tiles '
                    'that lie in the crops\' constant 0.5-padding (datautils.py:232-239) and are skipped, results bit-identical; `tflops` / `frac` '
                    'figures of this object are EXECUTED work over time; `images_equivalent_tflops` = the algorithmic work over the same time '
                    '(a throughput in whole-crop units, not a hardware rate)'}
        roofline['end_to_end'] = {'executed_tflops': round((alg_step - (alg_conv - exe_conv)) / (elapsed / args.steps * 1e3), 1),
                                  'frac_of_mfma_peak': round((alg_step - (alg_conv - exe_conv)) / (elapsed / args.steps * 1e3) / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
                                  'images_equivalent_tflops': round(alg_step / (elapsed / args.steps * 1e3), 1)}
        # shape of the crops of this step (the padding a crop carries is 1 - short / long side of its box)
        c0 = out['counts_host']
        roofline['crop_shapes'] = crop_shape_stats(torch.cat([out['boxes'][i, :c0[i]] for i in range(len(images))]))
        if not args.no_clocks:
            # the clock and the power the card delivers while it runs this step (and, for comparison, its bare MFMA loop)
            with ClockSampler(dev) as cs:
                t_end = time.perf_counter() + 1.5
                while time.perf_counter() < t_end:
                    pipe.run(images)
            clk = cs.summary()
            with ClockSampler(dev) as cs:
                ops.probe_mfma_bf16(1, seconds=1.5)
            clk_bare = cs.summary()
            if clk['sclk_mhz_median']:
                at_clock = MFMA_BF16_DENSE_PEAK_TFLOPS * clk['sclk_mhz_median'] / MFMA_PEAK_CLOCK_MHZ
                roofline['clocks'] = {'step': clk, 'bare_mfma_loop': clk_bare, 'peak_clock_mhz': MFMA_PEAK_CLOCK_MHZ,
                                      'mfma_peak_at_step_clock_tflops': round(at_clock, 1),
                                      'frac_at_step_clock': round(roofline['achieved'] / at_clock, 4),
                                      'note': 'amdgpu hwmon freq1_input / power1 sampled every 20 ms over 1.5 s of steps; `frac` above stays '
                                              'against the nominal 2.4 GHz peak'}
        return roofline

    roofline = leg('roofline', roofline_leg) if (not args.no_roofline and rank == 0) else None

    def workloads_leg():
        # BASELINE configs[1] and configs[3] on this same box (a few hundred ms of GPU time): the driver only runs the default line
        # (taken BEFORE the CPU-heavy legs: the oracle's OpenMP threads keep spinning for a while and slow the launch path)
        w = detector_workload(dev, 4, 1000, args.image_size, max(10, args.steps), max(3, args.warmup), args.detector_precision, collective=False, want_layers=True)
        workloads = {'detector_configs1': {k: v for k, v in w.items() if not k.startswith('_')}}
        workloads['match_stress_configs3'] = leg('match_stress', match_stress_cases, dev, 200, 3)
        workloads['embed_planted_boxes'] = leg('embed_planted_boxes', embed_planted_boxes, dev, enc, images[0], ipg * dpi)
        fs = leg('fitted_scenes', fitted_scenes_pipeline, dev, clf, ipg, dpi, args.image_size, max(5, args.steps // 2), args.detector_precision)
        if fs is not None:
            workloads['pipeline_fitted_scenes'] = fs
        return {k: v for k, v in workloads.items() if v is not None}

    workloads = leg('workloads', workloads_leg) if (not args.no_workloads and rank == 0) else None

    peaks = leg('peaks', measured_peaks, dev) if (rank == 0 and not args.no_peaks and not args.no_roofline) else None
    if roofline is not None and peaks is not None:
        bare = peaks['bare_mfma_loop_tflops']
        roofline['frac_of_bare_mfma_loop'] = round(roofline['achieved'] / max(bare['32x32x16'], bare['16x16x32']), 4)

    def parity_leg():
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        import accuracy                                           # tests/accuracy.py: the oracle as CHECKER (never timed, never shipped)
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        precs = (args.detector_precision,) + tuple(p for p in ('bf16', 'fp16') if p != args.detector_precision)
        rep = accuracy.run(n_images=4, image_size=args.image_size, galleries=(256,), dpi=dpi, queries=64, oracle_device='cpu',
                           match_dtypes=(args.match_dtype,), images_per_batch=4, precisions=precs)
        parity = accuracy.summary(rep)
        # the same sample with the detector whose head was FITTED on shelf scenes (tests/golden/fit_head.py): it finds the products, so
        # AP / AR300 against the TRUE boxes is non-vacuous -- north_star's "mAP within 0.1 pt" read the way the reference reports
        # it (cvpce/proposals_eval.py:19-48), per detector precision
        if os.path.exists(accuracy.FITTED_HEAD):
            repf = accuracy.run(n_images=4, image_size=args.image_size, galleries=(256,), dpi=dpi, queries=64, oracle_device='cpu',
                                match_dtypes=(args.match_dtype,), images_per_batch=4, precisions=precs, detector='fitted')
            fs = accuracy.summary(repf)
            parity['fitted_detector'] = {
                'what': 'head fitted on structured shelf scenes (tests/golden/fitted_head.pt); 4 evaluation scenes, true boxes = the pasted products',
                'true_boxes': repf['detection']['gt']['true_boxes'],
                'by_precision': {p_: {k_: v_ for k_, v_ in e_.items() if 'true_gt' in k_ or k_.startswith('pipeline_top1') or k_ in
                                      ('frac_oracle_boxes_iou90', 'ap50_area_vs_oracle', 'paired_box_diff_px_mean')}
                                 for p_, e_ in fs['by_precision'].items()}}
        parity['sample'] = ('4 structured shelf images through the whole HIP pipeline vs the whole fp32 CPU oracle; top level = the detector '
                            f'precision of this run ({args.detector_precision}: fp16 is the product default), by_precision = both detector modes (fp16 default, bf16 opt-in); '
                            '64 paired detections + 64 ground-truth crops vs a 256-product gallery; full-size figures (32 images, '
                            'G = 1000 / 3200): profiles/r05_accuracy.json')
        return parity

    parity = leg('parity', parity_leg) if (not args.no_parity and rank == 0 and world == 1) else None
    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        cpu = leg('cpu_baseline', cpu_baseline, det_sd, enc_sd, dpi, gallery.float().cpu(), args.image_size)
    if rank != 0:
        return None
    total_images = world * ipg * args.steps
    line = {
        'metric': 'shelf images/sec end-to-end (detect+embed+match)',
        'value': round(total_images / elapsed, 3), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'windows': window_stats(window_secs, args.steps),
        'config': {'workload': f'full production path: {ipg} shelf images/GPU of 3x{args.image_size}x{args.image_size} '
                               f'(SKU-110K shape) -> GLN detect (800x800 internal, detections_per_img={dpi}, conf>0.5) -> RoI crop 256x256 '
                               f'-> MAC-VGG16 embed -> cosine NN match, gallery={args.gallery}x1024 (BASELINE configs[2]/[4] per-GPU shape)',
                   'images_per_gpu': ipg, 'global_images': world * ipg, 'proposals_per_image': proposals, 'gallery': args.gallery,
                   'match_dtype': args.match_dtype, 'detector_precision': args.detector_precision,
                   'weights': 'seeded random init, cls head calibrated (cvpce_amd/synthetic.py)',
                   'parallelism': f'dp{world} (images sharded by global index, gallery embedded sharded + 1 all_gather, no steady-state collectives)',
                   'gallery_build_s': round(t_gallery, 3), 'collectives': collectives},
    }
    if h2d is not None:
        line.update(h2d)
    if cohead is not None:
        # co-headlines (same metric, same whole pipeline, same gallery): `value` above is measured on the random-weight detector's boxes, whose
        # uniform 2.6 : 1 shape lets the embedder skip ~45 % of its FLOPs as constant padding; these three do not depend on that
        line['value_lists_off'] = cohead['lists_off']['images_per_s']
        line['value_planted_boxes'] = cohead['planted_boxes']['images_per_s']
        fsp = (workloads or {}).get('pipeline_fitted_scenes', {}).get('padded_to_p')
        if fsp:
            line['value_fitted_scenes_p200'] = fsp['images_per_s']
            cohead['fitted_scenes_p200'] = fsp
        line['co_headlines'] = cohead
        if roofline is not None and 'dominant_kernel' in cohead['lists_off']:
            # the dominant kernel with every tile computed (executed = algorithmic FLOPs): the fraction that no box shape flatters
            roofline['lists_off'] = cohead['lists_off']['dominant_kernel']
    for key, val in (('value_by_detector_precision', by_precision), ('roofline', roofline), ('measured_peaks', peaks), ('parity', parity),
                     ('cpu_baseline', cpu), ('workloads', workloads), ('verify', verify)):
        if val is not None:
            line[key] = val
    if leg_errors:
        line['leg_errors'] = leg_errors
    if stub is not None:
        # a rehearsal with a CPU stand-in for the HIP pipeline (tests only): never a measurement, and the line says so
        line['stub'] = os.environ.get('CVPCE_BENCH_STUB')
        line['metric'] = 'STUB (CPU stand-in, not a measurement): ' + line['metric']
        line['data'] = 'stub'
        line['config']['weights'] = line['config']['detector_precision'] = None
    return line


def fitted_scenes_pipeline(dev, clf, ipg, dpi, image_size, steps, precision):
    """The whole pipeline on STRUCTURED shelf scenes with the detector whose head was fitted on such scenes (tests/golden/fitted_head.pt,
    a test fixture): realistic proposals -- as many confident boxes as there are products, of the products' shapes -- instead of the
    random-weight detector's 200 boxes of one shape.  Same gallery, same kernels; a side figure, never `value`."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import accuracy                                    # (only its fixture loader: nothing of the oracle is touched here)
    if not os.path.exists(accuracy.FITTED_HEAD):
        return None
    from cvpce_amd import ops, production, synthetic
    det = accuracy.fitted_detector(dpi, precision)[0].to(dev)
    products = synthetic.product_images(1024, seed=200)
    scenes = [synthetic.structured_shelf(i, image_size, image_size, products, pool=range(1000)) for i in range(ipg)]
    images = [sc[0].to(dev) for sc in scenes]
    pipe = production.BatchedPipeline(det, clf, 0.5)
    for _ in range(3):
        out = pipe.run(images)
    t, _ = timed_windows(lambda: pipe.run(images), steps, 1, dev, collective=False)
    ops.PROFILE = ops.ConvProfile()
    out = pipe.run(images)
    summ = ops.PROFILE.summary()
    ops.PROFILE = None
    counts = out['counts_host']
    bx = torch.cat([out['boxes'][i, :counts[i]] for i in range(len(images))]).to(torch.long).float()
    bw, bh = (bx[:, 2] - bx[:, 0]).clamp(min=1), (bx[:, 3] - bx[:, 1]).clamp(min=1)
    emb = {k: v for k, v in summ.items() if k in ('vgg_stem2_kernel', 'conv3x3_halo3_kernel')}
    # the same scenes padded to EXACTLY P = dpi proposals per image: the detector's confident boxes, cycled (box j of the padding = confident
    # box j mod c) -- the near-square shape distribution of real products at the headline's proposal count
    pb = out['boxes'].clone()
    for i, c in enumerate(counts):
        if 0 < c < dpi:
            pb[i, c:] = pb[i, :c].repeat((dpi + c - 1) // c, 1)[:dpi - c]
    prop = (pb, torch.full((len(images),), dpi, dtype=torch.int32, device=dev))
    for _ in range(2):
        pipe.run(images, proposals=prop)
    tp, _ = timed_windows(lambda: pipe.run(images, proposals=prop), steps, 3, dev, collective=False)
    fracp, _ = executed_fraction(lambda: pipe.run(images, proposals=prop))
    p200 = dict({'images_per_s': round(ipg * steps / tp, 3), 'ms_per_step': round(tp / steps * 1e3, 3), 'proposals_per_image': dpi,
                 'executed_over_algorithmic_conv_flops': fracp,
                 'what': 'fitted-scene proposals (near-square product boxes) padded to exactly P boxes per image by cycling the confident boxes'},
                **crop_shape_stats(pb))
    return {'padded_to_p': p200, 'images': ipg, 'images_per_s': round(ipg * steps / t, 2), 'ms_per_step': round(t / steps * 1e3, 3), 'detector_precision': precision,
            'confident_boxes_per_image': round(sum(counts) / len(counts), 1), 'products_per_image': round(sum(len(sc[1]) for sc in scenes) / len(scenes), 1),
            'short_over_long_mean': round(float((torch.minimum(bw, bh) / torch.maximum(bw, bh)).mean()), 4), 'wide_fraction': round(float((bw > bh).float().mean()), 4),
            'embed_stem_and_conv2_executed_over_algorithmic': round(sum(v['flops_executed'] for v in emb.values()) / max(1.0, sum(v['flops'] for v in emb.values())), 4),
            'note': 'structured shelf scenes + the fitted detector head (test fixture): proposals of realistic count and shape; a side figure'}


def embed_planted_boxes(dev, enc, image, n_boxes, seed=7, reps=5):
    """The embed stage on PLANTED proposals with a realistic spread of shapes (uniform boxes of 60-250 px on the
    2048^2 canvas, seed 7; width and height independent) instead of the random-weight detector's boxes, which all have one
    shape: how much of the constant-padding skipping survives a spread of aspect ratios.  Crops + extents once, then the
    embedder alone, with and without the work lists."""
    from cvpce_amd import ops
    from cvpce_amd.models import classification as C
    g = torch.Generator().manual_seed(seed)
    H, W = image.shape[1:]
    wh = 60 + 190 * torch.rand(n_boxes, 2, generator=g)
    xy = torch.rand(n_boxes, 2, generator=g) * (torch.tensor([W, H], dtype=torch.float32) - wh)
    boxes = torch.cat((xy, xy + wh), dim=1).to(dev)
    eng = enc.engine()
    crops = ops.crop_resize(image, boxes, 256, mode=2, mean=enc.input_mean, std=enc.input_std)
    ext = ops.crop_extents(boxes, None, H, W, 256)
    const = eng.const_crop(enc.input_mean, enc.input_std, 4, 256)

    def timed(**kw):
        eng.embed_packed(crops, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            eng.embed_packed(crops, **kw)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    ms_on, ms_off = timed(ext=ext, const_in=const), timed()
    ops.PROFILE = ops.ConvProfile()
    eng.embed_packed(crops, ext=ext, const_in=const)
    summ = ops.PROFILE.summary()
    ops.PROFILE = None
    alg, exe = sum(v['flops'] for v in summ.values()), sum(v['flops_executed'] for v in summ.values())
    b = boxes.to(torch.long).float().cpu()
    bw, bh = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    return {'boxes': n_boxes, 'box_px': '60-250 x 60-250 uniform, independent sides', 'short_over_long_mean': round(float((torch.minimum(bw, bh) / torch.maximum(bw, bh)).mean()), 4),
            'embed_ms_with_skipping': round(ms_on, 3), 'embed_ms_without': round(ms_off, 3), 'speedup': round(ms_off / ms_on, 4),
            'executed_over_algorithmic_flops': round(exe / alg, 4), 'executed_tflops': round(exe / ms_on / 1e9, 1)}


# ---------------------------------------------------------------------------------------------------------------------
# workload: detector only (BASELINE configs[1])
# ---------------------------------------------------------------------------------------------------------------------
def detector_layers(layers, steps, graph_ms):
    """`workloads.detector_configs1.layers`: the detector pass launch class by launch class (kernel + layer shape, launches of the step summed) --
    algorithmic FLOPs and bytes (inputs + outputs + weights once), arithmetic intensity, which roof binds it (MFMA above the ridge of
    2500 TFLOP/s / 8 TB/s = 312 FLOP/B, else HBM; 'latency' for the post-processing, which is neither) and the fraction of THAT roof
    reached.  Times are HIP events around the EAGER launches of a profiled pass (the production pass replays a hipGraph): their sum is
    given beside the graph-replayed pass time -- side branches that overlap in the graph are serial here."""
    ridge = MFMA_BF16_DENSE_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
    out = []
    for key, d in layers.items():
        ms = d['ms'] / steps
        fl, by = d['flops'] / steps, d['bytes'] / steps
        ai = fl / by if by > 0 else 0.0
        bound = 'latency' if fl == 0 and 'postprocess' in key else ('mfma' if ai >= ridge else 'hbm')
        e = {'launch_class': key, 'launches': round(d['launches'] / steps, 1), 'us': round(ms * 1e3, 1), 'gflop': round(fl / 1e9, 2), 'mb': round(by / 1e6, 1),
             'flop_per_byte': round(ai, 1), 'bound': bound}
        if ms > 0 and bound == 'mfma':
            e['tflops'] = round(fl / ms / 1e9, 1)
            e['frac_of_roof'] = round(fl / ms / 1e9 / MFMA_BF16_DENSE_PEAK_TFLOPS, 4)
        elif ms > 0 and bound == 'hbm':
            e['gbs'] = round(by / ms / 1e6, 1)
            e['frac_of_roof'] = round(by / ms / 1e6 / HBM_PEAK_GBS, 4)
        out.append(e)
    out.sort(key=lambda e: -e['us'])
    tot = sum(e['us'] for e in out)
    return {'classes': out, 'sum_of_eager_launches_us': round(tot, 1), 'graph_replayed_pass_us': round(graph_ms * 1e3, 1),
            'floor_us_at_the_roofs': round(sum((e['gflop'] / MFMA_BF16_DENSE_PEAK_TFLOPS * 1e-3 if e['bound'] == 'mfma' else e['mb'] / HBM_PEAK_GBS * 1e-3) * 1e6
                                               for e in out if e['bound'] != 'latency'), 1),
            'note': 'per launch class: algorithmic work, binding roof, achieved fraction of it; floor = every class at its roof, no overlap'}


def detector_workload(dev, ipg, dpi, image_size, steps, warmup, precision='bf16', ids=None, windows=1, want_profile=False, collective=True, want_layers=False):
    """`ipg` shelf images through the GLN detector only (transform, ResNet-50 + FPN, Gaussian branch, heads, top-k / NMS), the
    graph-replayed schedule `GLNEngine.detect` runs in production.  -> dict of figures (+ the engine / model for the caller)."""
    from cvpce_amd import ops, synthetic
    det = synthetic.synthetic_gln(seed=0, detections_per_img=dpi, precision=precision).to(dev)
    eng = det.engine()
    ids = list(range(ipg)) if ids is None else ids
    images = [synthetic.shelf_image(g, image_size, image_size).to(dev) for g in ids]
    outs = [None]

    def step():
        outs[0] = eng.detect(images, 1, dpi)

    for _ in range(warmup):
        step()
    elapsed, secs = timed_windows(step, steps, windows, dev, collective)
    ms = elapsed / steps * 1e3
    gf = 298.4 * ipg
    res = {'images': ipg, 'image_size': image_size, 'detections_per_img': dpi, 'precision': precision, 'ms_per_step': round(ms, 3),
           'images_per_s': round(ipg / ms * 1e3, 1), 'kept_per_image': float(outs[0][3].float().mean()),
           'algorithmic_gflop': round(gf, 1), 'tflops': round(gf / ms, 1), 'frac_of_mfma_peak': round(gf / ms / MFMA_BF16_DENSE_PEAK_TFLOPS, 4)}
    if windows > 1:
        res['windows'] = window_stats(secs, steps)
    if want_profile or want_layers:
        ops.PROFILE = ops.ConvProfile()
        for _ in range(steps):
            step()
        res['_conv_summary'] = ops.PROFILE.summary()
        layers = ops.PROFILE.summary_layers()
        ops.PROFILE = None
        if want_layers:
            res['layers'] = detector_layers(layers, steps, ms)
    res['_elapsed'] = elapsed
    return res


def run_detector(args, rank, local_rank, world, dev):
    from cvpce_amd import dist as cdist, synthetic
    ipg = args.images_per_gpu or 4
    dpi = args.detections_per_img or 1000
    ids = cdist.shard_images(world * ipg, rank, world)
    w = detector_workload(dev, ipg, dpi, args.image_size, args.steps, args.warmup, args.detector_precision, ids=ids, windows=args.windows,
                          want_profile=(not args.no_roofline and rank == 0))
    elapsed = w.pop('_elapsed')
    summ = w.pop('_conv_summary', None)
    roofline = None
    if summ is not None:
        roofline = conv_roofline(summ, {'detect': {k: w[k] for k in ('ms_per_step', 'algorithmic_gflop', 'tflops', 'frac_of_mfma_peak')}})
    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        det_sd = {k: v.clone() for k, v in synthetic.synthetic_gln(seed=0, detections_per_img=dpi).state_dict().items()}
        cpu = cpu_baseline(det_sd, None, dpi, None, args.image_size)
    if rank != 0:
        return None
    line = {'metric': 'shelf images/sec, GLN detector only (convs + Gaussian head + top-k/NMS)', 'value': round(world * ipg * args.steps / elapsed, 3),
            'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.detector_precision, 'data': 'synthetic',
            'windows': w.get('windows'),
            'config': {'workload': f'GLN detector only: {ipg} x 3x{args.image_size}x{args.image_size} (SKU-110K shape, 800x800 internal), '
                                   f'detections_per_img={dpi} (BASELINE configs[1])', 'images_per_gpu': ipg, 'kept_per_image': w['kept_per_image'],
                       'detector_precision': args.detector_precision,
                       'weights': 'seeded random init, cls head calibrated (cvpce_amd/synthetic.py)', 'parallelism': f'dp{world}'}}
    if roofline is not None:
        line['roofline'] = roofline
    if cpu is not None:
        line['cpu_baseline'] = cpu
    return line


# ---------------------------------------------------------------------------------------------------------------------
# workload: distance-GEMM stress (BASELINE configs[3])
# ---------------------------------------------------------------------------------------------------------------------
def match_stress_cases(dev, iters, warmup):
    """BASELINE configs[3]: P queries x 10 000 gallery rows x D, bf16 distance GEMM + fused top-1, per launch (graph-replayed)."""
    from cvpce_amd import ops
    G = 10000
    cases = []
    g = torch.Generator().manual_seed(0)
    for P, D in ((200, 512), (200, 1024), (1600, 512), (1600, 1024)):
        gal = torch.nn.functional.normalize(torch.randn(G, D, generator=g), dim=1).to(dev).to(torch.bfloat16)
        q = torch.nn.functional.normalize(torch.randn(P, D, generator=torch.Generator().manual_seed(1)), dim=1).to(dev).to(torch.bfloat16)
        gn, qn = ops.row_norms(gal), ops.row_norms(q)
        for _ in range(max(3, warmup)):
            ops.match_topk(q, gal, 1, q_norms=qn, g_norms=gn)
        torch.cuda.synchronize()
        # the launches are replayed from a hipGraph: at ~10-50 us per launch the Python -> dispatcher -> ctypes path would
        # otherwise be what the events time
        per_graph = 20
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode='thread_local'):
            for _ in range(per_graph):
                ops.match_topk(q, gal, 1, q_norms=qn, g_norms=gn)
        graph.replay()
        torch.cuda.synchronize()
        reps = max(1, iters // per_graph)
        samples = []
        for _ in range(3):                                   # median of three timed rounds: one slow replay must not decide a case
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                graph.replay()
            e1.record(); torch.cuda.synchronize()
            samples.append(e0.elapsed_time(e1) * 1e3 / (reps * per_graph))
        us = sorted(samples)[1]
        del graph
        flop = 2.0 * P * G * D
        byts = (G * D + P * D) * 2 + (G + P) * 4 + P * 8          # both operands once + norms + the (P,1) int64 result
        cases.append({'P': P, 'G': G, 'D': D, 'us_per_launch': round(us, 2), 'tflops': round(flop / us / 1e6, 1),
                      'algorithmic_gbs': round(byts / us / 1e3, 1), 'arithmetic_intensity_flop_per_byte': round(flop / byts, 1),
                      'bound': 'hbm' if flop / byts < MFMA_BF16_DENSE_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS else 'mfma',
                      'frac_of_hbm_peak': round(byts / us / 1e3 / HBM_PEAK_GBS, 4), 'frac_of_mfma_peak': round(flop / us / 1e6 / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
                      'queries_per_s': round(P / us * 1e6, 0)})
    return cases


def run_match_stress(args, rank, local_rank, world, dev):
    iters = max(20, args.steps * 10)
    cases = match_stress_cases(dev, iters, args.warmup)
    if rank != 0:
        return None
    head = cases[0]
    return {'metric': 'distance-GEMM stress: queries/s, 200 proposals x 10 000 x 512-d gallery, bf16 MFMA with fused top-1', 'value': head['queries_per_s'],
            'unit': 'queries/s', 'n_gpus': world, 'steps': iters, 'warmup': max(3, args.warmup), 'ms_per_step': round(head['us_per_launch'] / 1e3, 5),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'distance GEMM + fused top-1 (cvpce_match_topk), P = 200 queries (one image), G = 10 000 gallery rows, D = 512 '
                                   '(BASELINE configs[3]); cases: P in {200, 1600} x D in {512, 1024}', 'parallelism': f'dp{world} (replicas)'},
            'roofline': {'bound': head['bound'], 'achieved': head['algorithmic_gbs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': head['frac_of_hbm_peak'], 'traffic': None, 'kernel': 'match_kernel<bf16> + match_merge_kernel',
                         'avg_launch_us': head['us_per_launch'], 'cases': cases}}


# ---------------------------------------------------------------------------------------------------------------------
# the printed line: compact (< 4 KB); everything else goes to the details file
# ---------------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 4096
BASE_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data')
ROOFLINE_KEYS = ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'launches', 'avg_launch_us')


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(full, details_name):
    """The ONE line rank 0 prints: the contract's base keys, `config` (workload string + scalars), the co-headline values, `roofline`
    (the dominant kernel: scalars only), `cpu_baseline`, `parity` (scalars), `verify` (image count + digest) -- nothing nested deeper,
    no tables.  `full` (written to `details_name`) keeps every object."""
    line = _pick(full, BASE_KEYS)
    cfg = full.get('config', {})
    c = {'workload': str(cfg.get('workload', ''))[:400]}
    c.update(_pick(cfg, ('images_per_gpu', 'global_images', 'proposals_per_image', 'gallery', 'match_dtype', 'detector_precision', 'kept_per_image',
                         'parallelism', 'gallery_build_s')))
    if isinstance(c.get('parallelism'), str):
        c['parallelism'] = c['parallelism'].split(' ')[0]
    coll = cfg.get('collectives')
    c['collectives'] = None if coll is None else _pick(coll, ('backend', 'world_size', 'all_gather', 'data_path_collectives_per_step'))
    line['config'] = c
    for k in ('value_lists_off', 'value_planted_boxes', 'value_fitted_scenes_p200', 'value_with_h2d', 'stub'):
        if k in full:
            line[k] = full[k]
    if isinstance(full.get('windows'), dict):
        line['windows_ms_per_step'] = full['windows'].get('ms_per_step')
    r = full.get('roofline')
    if isinstance(r, dict):
        cr = _pick(r, ROOFLINE_KEYS)
        if isinstance(r.get('lists_off'), dict):
            cr['lists_off_frac'] = r['lists_off'].get('frac')
            cr['lists_off_avg_launch_us'] = r['lists_off'].get('avg_launch_us')
        if isinstance(r.get('end_to_end'), dict):
            cr['end_to_end_frac'] = r['end_to_end'].get('frac_of_mfma_peak')
        if isinstance(r.get('strip_launches'), dict):
            cr['strip_launches'] = _pick(r['strip_launches'], ('launches', 'avg_launch_us', 'frac'))
        if isinstance(r.get('stages'), dict):
            cr['stage_ms'] = {k: v.get('ms_per_step') for k, v in r['stages'].items() if isinstance(v, dict)}
        if isinstance(r.get('clocks'), dict):
            cr['sclk_mhz'] = r['clocks'].get('step', {}).get('sclk_mhz_median')
            cr['power_w'] = r['clocks'].get('step', {}).get('power_w_median')
        if 'frac_of_bare_mfma_loop' in r:
            cr['frac_of_bare_mfma_loop'] = r['frac_of_bare_mfma_loop']
        line['roofline'] = cr
    cb = full.get('cpu_baseline')
    if isinstance(cb, dict):
        line['cpu_baseline'] = dict(_pick(cb, ('value', 'unit', 'cores', 'kind')), sample=str(cb.get('sample_short') or cb.get('sample', ''))[:120])
    p = full.get('parity')
    if isinstance(p, dict):
        # random-weight detector vs the fp32 oracle: how many oracle boxes are found at IoU > 0.9, the mean corner difference of paired
        # boxes, top-1 agreement of the matcher; fitted detector (finds the pasted products): mAP / AR300 / top-1 deltas against the TRUE boxes
        cp = _pick(p, ('images', 'frac_oracle_boxes_iou90', 'paired_box_diff_px_mean', 'ap50_area_vs_oracle', 'ar300_vs_oracle'))
        g = next((v for k, v in p.items() if k.startswith('G') and isinstance(v, dict)), {})
        cp.update({'top1_agree': g.get('top1_agree'), 'top1_acc_delta_pt': g.get('top1_acc_delta_pt')})
        f = p.get('fitted_detector', {}).get('by_precision', {}).get(cfg.get('detector_precision'), {}) if isinstance(p.get('fitted_detector'), dict) else {}
        for k in ('map_delta_pt_true_gt', 'ar300_delta_pt_true_gt'):
            if k in f:
                cp['fitted_' + k] = f[k]
        t1 = [v for k, v in f.items() if k.startswith('pipeline_top1_delta_pt')]
        if t1:
            cp['fitted_top1_delta_pt'] = t1[0]
        line['parity'] = cp
    w = full.get('workloads')
    if isinstance(w, dict):
        cw = {}
        if isinstance(w.get('detector_configs1'), dict):
            cw['detector_configs1'] = _pick(w['detector_configs1'], ('images', 'ms_per_step', 'images_per_s', 'frac_of_mfma_peak'))
        if isinstance(w.get('match_stress_configs3'), list):
            cw['match_stress_configs3_us'] = {f"{e['P']}x{e['G']}x{e['D']}": e['us_per_launch'] for e in w['match_stress_configs3']}
        line['workloads'] = cw
    v = full.get('verify')
    if isinstance(v, dict):
        line['verify'] = _pick(v, ('images', 'digest'))
    if full.get('leg_errors'):
        line['leg_errors'] = {k: str(e)[:80] for k, e in full['leg_errors'].items()}
    line['details'] = details_name
    return line


def emit(full, details_path):
    """Write the full object to `details_path` (a failure to write is reported on the line, never fatal) and return the compact
    line as a JSON string of < LINE_LIMIT bytes: optional objects are dropped, largest first, if it would ever be longer."""
    name = os.path.basename(details_path)
    try:
        os.makedirs(os.path.dirname(os.path.abspath(details_path)), exist_ok=True)
        with open(details_path, 'w') as f:
            json.dump(full, f, indent=1, default=str)
    except OSError as e:
        name = f'not written ({type(e).__name__})'
    line = compact_line(full, name)
    s = json.dumps(line, separators=(',', ':'), default=str)
    for k in ('workloads', 'parity', 'windows_ms_per_step', 'leg_errors', 'verify', 'cpu_baseline', 'roofline'):
        if len(s) < LINE_LIMIT:
            break
        if k in ('cpu_baseline', 'roofline'):                # (never dropped whole: cut to the contract's own keys)
            line[k] = _pick(line[k], ROOFLINE_KEYS if k == 'roofline' else ('value', 'unit', 'cores', 'kind'))
        else:
            line.pop(k, None)
        s = json.dumps(line, separators=(',', ':'), default=str)
    assert len(s) < LINE_LIMIT, len(s)
    return s


def host_threads_for(world, cores):
    """Host threads one rank may use: its share of the node's cores (image synthesis, pinned staging and the oracle legs use torch's
    intra-op pool; N ranks on one node must not each claim every core)."""
    return max(1, cores // max(1, world))


def main():
    global ALLOW_STUB
    args = parse()
    ALLOW_STUB = args.allow_stub
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_launcher(args))
    from cvpce_amd import dist as cdist
    rank, local_rank, world = cdist.init()
    if world != args.gpus and world > 1:
        args.gpus = world
    if world > 1:
        torch.set_num_threads(host_threads_for(world, os.cpu_count() or 1))
    if stub_factory() is not None:
        dev = torch.device('cpu')                 # CVPCE_BENCH_STUB (tests/test_dist_cpu.py): the rehearsal of main() on a box without a GPU
    else:
        dev = torch.device('cuda', local_rank % max(1, torch.cuda.device_count()))   # (a 1-GPU rehearsal may stack ranks on cuda:0)
        torch.cuda.set_device(dev)
    full = {'pipeline': run_pipeline, 'detector': run_detector, 'match-stress': run_match_stress}[args.workload](args, rank, local_rank, world, dev)
    if rank == 0:
        print(emit(full, args.details or os.path.join(ROOT, 'bench_details.json')), flush=True)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        # Rank 0 runs its side legs (roofline, parity, CPU baseline: ~2 min) after the last collective; the other ranks have nothing left to
        # do.  They wait for rank 0 on the rendezvous STORE (a host-side key, no collective, nothing spinning on their GPUs) so that every
        # rank tears its process group down while all peers are still alive.
        try:
            import datetime
            store = dist.distributed_c10d._get_default_store()
            if rank == 0:
                store.set('cvpce_bench_done', '1')
            else:
                store.wait(['cvpce_bench_done'], datetime.timedelta(seconds=3600))
        except Exception:                         # noqa: BLE001 (a missing store API only loses the orderly teardown)
            pass
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
