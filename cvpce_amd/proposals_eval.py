"""Detector-only evaluation harness -- counterpart of /root/reference/cvpce/proposals_eval.py:9-48
(`load_gln`, `evaluate_gln_sync`; the multiprocess `evaluate_gln_async` :50-87 only changes where the CPU metric
code runs).  The per-image triple handed to the metric routine is the reference's: (target boxes, predicted boxes,
scores).  `dataset` is any of the readers in datautils.py, or any iterable of `(image (3,H,W) f32 in [0,1],
target dict with 'boxes' (T,4))`."""
import torch

from . import metrics, utils
from .models import proposals


def load_gln(save_file, trim_module_prefix, **kwargs):
    """proposals_eval.py:9-17 (checkpoint dict key `model_state_dict`, optional DDP `module.` prefix)."""
    state_dict = torch.load(save_file, map_location='cpu')['model_state_dict']
    if trim_module_prefix:
        state_dict = utils.trim_module_prefix(state_dict)
    model = proposals.gln(pretrained_backbone=False, **kwargs)
    model.load_state_dict(state_dict)
    return model.cuda().eval()


def _batches(dataset, batch_size):
    batch = []
    for item in dataset:
        batch.append(item)
        if len(batch) == batch_size:
            yield batch
            batch = []
    if batch:
        yield batch


@torch.no_grad()
def evaluate_gln_sync(model, dataset, thresholds=(.5,), batch_size=1, num_workers=2, plots=False, silent=True,
                      plot_res_reduction=1):
    predictions, targets, confidences = [], [], []
    for i, batch in enumerate(_batches(dataset, batch_size)):
        if not silent and i % 100 == 0:
            print(f'{i}...')
        result = model([img.cuda(non_blocking=True) for img, _ in batch])
        for r, (_, t) in zip(result, batch):
            predictions.append(r['boxes'].detach().cpu())
            targets.append(t['boxes'].detach().cpu())
            confidences.append(r['scores'].detach().cpu())
    res = metrics.calculate_metrics(targets, predictions, confidences, thresholds)
    return {thr: {k: v for k, v in itm.items() if k != 'raw'} for thr, itm in res.items()}


def evaluate_gln(save_file, dataset, thresholds=(.5,), batch_size=1, trim_module_prefix=True, **_ignored):
    return evaluate_gln_sync(load_gln(save_file, trim_module_prefix), dataset, thresholds, batch_size)
