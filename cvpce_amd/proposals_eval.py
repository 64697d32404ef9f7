"""Detector-only evaluation harness -- counterpart of /root/reference/cvpce/proposals_eval.py:9-91
(`load_gln`, `evaluate_gln_sync`, `evaluate_gln_async`, `evaluate_gln`; the multiprocess variant only changes where the
CPU metric code runs: at > 1 000 detector images/s the single-threaded matching loop is what the harness would wait on).  The per-image triple handed to the metric routine is the reference's: (target boxes, predicted boxes,
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


@torch.no_grad()
def evaluate_gln_async(model, dataset, thresholds=(.5,), batch_size=1, num_workers=2, num_metric_processes=4, plots=False,
                       plot_res_reduction=1, silent=True):
    """proposals_eval.py:50-87: detections are handed to `num_metric_processes` matching workers as they are produced."""
    queue, mqueue, pipe = metrics.calculate_metrics_async(processes=num_metric_processes, iou_thresholds=thresholds)
    for i, batch in enumerate(_batches(dataset, batch_size)):
        if not silent and i % 100 == 0:
            print(f'GPU: {i}...')
        result = model([img.cuda(non_blocking=True) for img, _ in batch])
        for r, (_, t) in zip(result, batch):
            queue.put((t['boxes'], r['boxes'], r['scores']))
    queue.join()
    for _ in range(num_metric_processes):
        queue.put(None)
    queue.join()
    mqueue.join()
    mqueue.put(None)
    res = pipe.recv()
    mqueue.join()
    if res is None:                               # empty dataset: what the synchronous routine returns
        res = metrics.calculate_metrics([], [], [], thresholds)
    return {thr: {k: v for k, v in itm.items() if k != 'raw'} for thr, itm in res.items()}


def evaluate_gln(save_file, dataset, thresholds=(.5,), batch_size=1, num_workers=2, num_metric_processes=4, plots=False,
                 trim_module_prefix=True, resolution_reduction=1):
    """proposals_eval.py:89-91.  num_metric_processes = 0 runs the matching in-process (evaluate_gln_sync)."""
    model = load_gln(save_file, trim_module_prefix)
    if num_metric_processes and num_metric_processes > 0:
        return evaluate_gln_async(model, dataset, thresholds, batch_size, num_workers, num_metric_processes)
    return evaluate_gln_sync(model, dataset, thresholds, batch_size)
