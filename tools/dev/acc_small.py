"""dev tool: the accuracy sample of tests/test_gpu_accuracy.py (8 scenes of 1024^2) with the fitted detector, figures printed."""
import sys, os, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import accuracy
torch.set_num_threads(16)
r = accuracy.run(n_images=8, image_size=1024, galleries=(256,), dpi=200, queries=96, oracle_device='cpu', match_dtypes=('bf16', 'f32'),
                 images_per_batch=8, control_images=0, precisions=('bf16', 'fp16'), detector='fitted')
for p, rp in r['by_precision'].items():
    d = rp['detection']; g = d['gt']
    print(p, 'iou90', d['frac_oracle_boxes_iou90'], 'area_vs_orc', d['ap50_area_vs_oracle'], 'px', d['paired_box_diff_px_mean'], 'dscore', d['paired_abs_score_diff_mean'])
    print('  gt', {k: (round(v, 5) if isinstance(v, float) else v) for k, v in g.items() if k != 'note'})
    print('  pairs', {k: (m['top1_agree'], m.get('vs_true_product')) for k, m in rp['matching_pairs'].items()})
print(json.dumps(accuracy.summary(r))[:1500])
print('seconds', r['seconds'])
