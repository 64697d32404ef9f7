"""bench.py's printed line stays under 4 KB whatever the side legs return (round 5's 23 KB line left the round unmeasured: the driver
could not parse it); the full object goes to the details file.  CPU only: no kernels are launched."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _full(bloat):
    """a result object shaped like run_pipeline's, with `bloat` rows in every table that grew in earlier rounds."""
    row = {'launch_class': 'conv_dma_kernel 200x200 128->64 k3 s1', 'launches': 1.0, 'us': 81.8, 'gflop': 23.59, 'mb': 62.0, 'flop_per_byte': 380.3,
           'bound': 'mfma', 'tflops': 288.4, 'frac_of_roof': 0.1154}
    return {
        'metric': 'shelf images/sec end-to-end (detect+embed+match)', 'value': 280.4, 'unit': 'images/s', 'n_gpus': 1, 'steps': 20, 'warmup': 5,
        'ms_per_step': 28.53, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'windows': {'n': 3, 'ms_per_step': [28.5, 28.6, 28.4]},
        'config': {'workload': 'full production path ' + 'x' * 1000, 'images_per_gpu': 8, 'global_images': 8, 'proposals_per_image': 200.0, 'gallery': 3200,
                   'match_dtype': 'bf16', 'detector_precision': 'fp16', 'weights': 'w' * 300, 'parallelism': 'dp1 (images sharded ...)', 'gallery_build_s': 1.0,
                   'collectives': {'backend': 'nccl', 'world_size': 8, 'all_gather': 1, 'all_reduce': 9, 'barrier': 20, 'in_timed_windows': {'a': 1},
                                   'data_path_collectives_per_step': 0}},
        'value_with_h2d': 270.0, 'value_lists_off': 171.0, 'value_planted_boxes': 190.0, 'value_fitted_scenes_p200': 168.0,
        'co_headlines': {'lists_off': {'all_conv_kernels': {f'k{i}': row for i in range(bloat)}}},
        'roofline': {'bound': 'mfma', 'achieved': 1469.7, 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': 0.5879, 'traffic': 1.133, 'kernel': 'conv3x3_halo2_kernel',
                     'launches': 360, 'avg_launch_us': 1920.0, 'lists_off': {'frac': 0.63, 'avg_launch_us': 1563.9}, 'end_to_end': {'frac_of_mfma_peak': 0.57},
                     'strip_launches': {'launches': 40, 'avg_launch_us': 91.0, 'frac': 0.3, 'total_ms': 3.6},
                     'stages': {k: {'ms_per_step': 1.0, 'note': 'n' * 200} for k in ('detect', 'crop', 'embed', 'match')},
                     'all_conv_kernels': {f'k{i}': row for i in range(bloat)}, 'hbm_stages': {f'k{i}': row for i in range(bloat)},
                     'clocks': {'step': {'sclk_mhz_median': 2050.0, 'power_w_median': 1390.0}}},
        'cpu_baseline': {'value': 0.012, 'unit': 'images/s', 'cores': 16, 'kind': 'port', 'sample': 's' * 600},
        'parity': {'images': 4, 'ap50_vs_oracle': 0.99, 'by_precision': {'fp16': {f'm{i}': 0.1 for i in range(bloat)}}, 'sample': 'p' * 500},
        'workloads': {'detector_configs1': {'images': 4, 'ms_per_step': 2.4, 'images_per_s': 1600.0, 'frac_of_mfma_peak': 0.19, 'layers': {'classes': [row] * bloat}},
                      'match_stress_configs3': [{'P': p, 'G': 10000, 'D': d, 'us_per_launch': 9.9} for p in (200, 1600) for d in (512, 1024)]},
        'verify': {'images': 64, 'digest': 'd' * 64, 'per_image': {str(i): 'h' * 16 for i in range(64)}},
    }


def test_line_is_compact_and_details_complete(tmp_path):
    for bloat in (0, 40, 4000):
        full = _full(bloat)
        path = str(tmp_path / f'details_{bloat}.json')
        s = bench.emit(full, path)
        assert len(s) < 4096 and '\n' not in s
        line = json.loads(s)
        assert all(line[k] == full[k] for k in bench.BASE_KEYS)
        assert line['roofline']['frac'] == 0.5879 and line['roofline']['lists_off_frac'] == 0.63 and line['roofline']['end_to_end_frac'] == 0.57
        assert line['roofline']['strip_launches'] == {'launches': 40, 'avg_launch_us': 91.0, 'frac': 0.3}
        assert line['cpu_baseline']['kind'] == 'port' and len(line['cpu_baseline']['sample']) <= 120
        assert line['config']['collectives'] == {'backend': 'nccl', 'world_size': 8, 'all_gather': 1, 'data_path_collectives_per_step': 0}
        assert line['verify'] == {'images': 64, 'digest': 'd' * 64} and line['details'] == f'details_{bloat}.json'
        assert line['value_lists_off'] == 171.0 and line['value_fitted_scenes_p200'] == 168.0 and line['value_with_h2d'] == 270.0
        assert json.load(open(path)) == json.loads(json.dumps(full))          # nothing is lost: the file holds the whole object


def test_unwritable_details_path_does_not_lose_the_line():
    s = bench.emit(_full(3), '/proc/definitely/not/writable/details.json')
    assert json.loads(s)['details'].startswith('not written')


def test_stub_hook_refused_without_flag():
    """CVPCE_BENCH_STUB alone must not replace the HIP pipeline: bench.py exits with an error and prints no line."""
    env = dict(os.environ, CVPCE_BENCH_STUB='bench_stub:build', PYTHONPATH=os.pathsep.join([os.path.join(ROOT, 'tests'), ROOT]))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '0'], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and '--allow-stub' in r.stderr and '{"metric"' not in r.stdout
