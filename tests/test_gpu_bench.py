"""bench.py's JSON contract on small inputs: every workload prints ONE COMPACT line (< 4 KB: round 5's 23 KB line could not be parsed by the
driver) with the fields the driver reads, the roofline and cpu_baseline objects, and writes the full result object (tables, stages,
co-headline objects) to the details file the line names."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config'}


def _run(tmp_path, *args):
    """-> (the printed line, the details object)."""
    det = os.path.join(str(tmp_path), 'details.json')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *args, '--details', det], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    assert len(lines[0]) < 4096, len(lines[0])                      # the driver must be able to parse it
    line = json.loads(lines[0])
    assert line['details'] == 'details.json' and 'leg_errors' not in line, line.get('leg_errors')
    return line, json.load(open(det))


def _check_compact(c, d):
    """the compact line `c` carries the details object `d`'s headline figures unchanged, and nothing nested deeper than one level below
    its top-level objects."""
    for k in BASE - {'config'}:
        assert c[k] == d[k], k
    assert c['config']['workload'] == d['config']['workload'][:400]
    r, rd = c['roofline'], d['roofline']
    assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'} <= set(r)
    assert all(r[k] == rd[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'))
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    for v in c.values():
        if isinstance(v, dict):
            assert all(not isinstance(x, list) or len(x) <= 8 for x in v.values())
            assert all(not isinstance(y, (dict, list)) for x in v.values() if isinstance(x, dict) for y in x.values())


def test_pipeline_line_small(cuda, tmp_path):
    c, d = _run(tmp_path, '--steps', '2', '--warmup', '1', '--images-per-gpu', '2', '--image-size', '1024', '--gallery', '128', '--no-peaks')
    _check_compact(c, d)
    assert BASE <= set(c) and c['config']['detector_precision'] == 'fp16' and c['config']['collectives'] is None
    assert c['value_lists_off'] == d['value_lists_off'] and c['value_planted_boxes'] == d['value_planted_boxes'] and c['value_with_h2d'] == d['value_with_h2d']
    assert c['value_fitted_scenes_p200'] == d['value_fitted_scenes_p200']
    assert c['roofline']['kernel'] == 'conv3x3_halo2_kernel' and c['roofline']['lists_off_frac'] == d['roofline']['lists_off']['frac']
    assert c['roofline']['end_to_end_frac'] == d['roofline']['end_to_end']['frac_of_mfma_peak'] and set(c['roofline']['stage_ms']) == {'detect', 'crop', 'embed', 'match'}
    cb = c['cpu_baseline']
    assert cb['kind'] == 'port' and cb['cores'] >= 1 and cb['value'] == d['cpu_baseline']['value'] > 0 and 0 < len(cb['sample']) <= 120
    assert c['parity']['images'] == 4 and 0.9 < c['parity']['frac_oracle_boxes_iou90'] <= 1.0 and abs(c['parity']['fitted_map_delta_pt_true_gt']) <= 1.0
    assert c['workloads']['detector_configs1']['ms_per_step'] == d['workloads']['detector_configs1']['ms_per_step']
    assert len(c['workloads']['match_stress_configs3_us']) == 4
    assert BASE <= set(d) and d['unit'] == 'images/s' and d['n_gpus'] == 1 and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['dtype'] == 'bf16' and 'workload' in d['config'] and d['value'] > 0
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert set(r['stages']) == {'detect', 'crop', 'embed', 'match'}
    if 'clocks' in r:                                    # amdgpu hwmon readable on this box: the card's clock and power under the step
        k = r['clocks']
        assert 300 < k['step']['sclk_mhz_median'] <= k['peak_clock_mhz'] + 100 and k['step']['samples'] >= 10
        assert abs(k['frac_at_step_clock'] - r['achieved'] / k['mfma_peak_at_step_clock_tflops']) < 1e-3
    assert d['value_with_h2d'] > 0 and d['upload_mb_per_step'] > 0
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and 'sample' in c
    assert d['windows']['n'] == 3 and len(d['windows']['ms_per_step']) == 3 and d['windows']['min'] <= d['ms_per_step'] <= d['windows']['max']
    w = d['workloads']                                   # BASELINE configs[1] and configs[3] ride on the default (driver-run) line
    assert w['detector_configs1']['images'] == 4 and w['detector_configs1']['detections_per_img'] == 1000 and w['detector_configs1']['ms_per_step'] > 0
    assert {(c['P'], c['D']) for c in w['match_stress_configs3']} == {(200, 512), (200, 1024), (1600, 512), (1600, 1024)}
    assert set(d['parity']['by_precision']) == {'bf16', 'fp16'}
    assert d['parity']['by_precision']['fp16']['frac_oracle_boxes_iou90'] >= d['parity']['by_precision']['bf16']['frac_oracle_boxes_iou90']
    p = d['parity']
    assert p['images'] == 4 and 0.5 < p['ap50_vs_oracle'] <= 1.0 and abs(p['G256_bf16']['top1_acc_delta_pt']) <= 2.0
    # round 4: the throughput in BOTH detector storage modes, executed vs algorithmic work, the fitted-detector parity against true boxes
    v = d['value_by_detector_precision']
    assert v['bf16'] > 0 and v['fp16'] > 0 and abs(v['fp16'] - d['value']) < 1e-6          # the headline runs in the DEFAULT mode: fp16
    assert d['config']['detector_precision'] == 'fp16'
    g = r['gflop_per_step']
    assert 0 < g['executed'] <= g['algorithmic'] and r['end_to_end']['executed_tflops'] <= r['end_to_end']['images_equivalent_tflops']
    assert 'algorithmic_tflops' not in r and all('algorithmic_tflops' not in e for e in r['all_conv_kernels'].values())
    assert all(0 < e['frac_of_mfma_peak'] <= 1.0 for e in r['all_conv_kernels'].values())           # executed FLOPs cannot exceed the peak
    assert 0 < r['crop_shapes']['short_over_long_mean'] <= 1
    f = p['fitted_detector']
    assert f['true_boxes'] > 50 and set(f['by_precision']) == {'bf16', 'fp16'}
    assert f['by_precision']['fp16']['ap50_true_gt_oracle'] > 0.5 and abs(f['by_precision']['fp16']['map_delta_pt_true_gt']) <= 1.0
    e = w['embed_planted_boxes']
    assert e['embed_ms_with_skipping'] < e['embed_ms_without'] and 0 < e['executed_over_algorithmic_flops'] < 1
    assert w['pipeline_fitted_scenes']['images_per_s'] > 0 and w['pipeline_fitted_scenes']['confident_boxes_per_image'] > 5
    # round 5: co-headlines of the whole pipeline that do not depend on the random-weight detector's box shapes
    co = d['co_headlines']
    assert d['value_lists_off'] == co['lists_off']['images_per_s'] > 0 and co['lists_off']['executed_over_algorithmic_conv_flops'] == 1.0
    assert d['value_planted_boxes'] == co['planted_boxes']['images_per_s'] > 0 and co['planted_boxes']['proposals_per_image'] == 200
    assert 0.3 < co['planted_boxes']['short_over_long_mean'] < 0.9 and 0 < co['planted_boxes']['executed_over_algorithmic_conv_flops'] <= 1.0
    assert d['value_fitted_scenes_p200'] == co['fitted_scenes_p200']['images_per_s'] > 0 and co['fitted_scenes_p200']['short_over_long_mean'] > 0.6
    assert r['lists_off']['kernel'] == 'conv3x3_halo2_kernel' and 0 < r['lists_off']['frac'] <= 1.0
    # round 5: the HBM-bound stages against the HBM roofline (algorithmic bytes, HIP-event time)
    h = r['hbm_stages']
    assert {'gln_transform_batch_kernel', 'crop_resize_kernel', 'conv1x1_kernel'} <= set(h) and 'gauss_tail_kernel' not in h and 'thin3x3_kernel' not in h   # (round 6: the Gaussian subnet is ONE launch, MFMA-side)
    assert 'gauss_subnet_kernel' in r['all_conv_kernels']
    for name, e in h.items():
        if not name.startswith('_'):
            assert e['algorithmic_gb_per_step'] > 0 and e['ms_per_step'] > 0 and 0 < e['frac_of_hbm_peak'] <= 1.0, (name, e)
    assert d['config']['collectives'] is None            # one rank, no process group
    # round 5: the detector of BASELINE configs[1] launch class by launch class against the roof that binds each
    ly = w['detector_configs1']['layers']
    assert len(ly['classes']) >= 20 and ly['sum_of_eager_launches_us'] > 0 and 0 < ly['floor_us_at_the_roofs'] < ly['graph_replayed_pass_us']
    assert {e['bound'] for e in ly['classes']} >= {'mfma', 'hbm', 'latency'}
    assert all(0 < e['frac_of_roof'] <= 1.0 for e in ly['classes'] if e['bound'] != 'latency'), ly['classes']

def test_detector_and_match_stress_lines(cuda, tmp_path):
    c, d = _run(tmp_path, '--workload', 'detector', '--steps', '2', '--warmup', '1', '--images-per-gpu', '2', '--image-size', '1024', '--no-cpu-baseline')
    _check_compact(c, d)
    assert BASE <= set(d) and 'detector' in d['metric'] and d['roofline']['stages']['detect']['ms_per_step'] > 0
    cm, m = _run(tmp_path, '--workload', 'match-stress', '--steps', '2')
    _check_compact(cm, m)
    assert BASE <= set(m) and m['unit'] == 'queries/s' and m['roofline']['bound'] == 'hbm' and len(m['roofline']['cases']) == 4
    assert {(c['P'], c['D']) for c in m['roofline']['cases']} == {(200, 512), (200, 1024), (1600, 512), (1600, 1024)}
