"""Embed+match top-k accuracy harness -- counterpart of /root/reference/cvpce/classification_eval.py:6-56.
`testset` yields `(image (3,H,W) f32 in [0,1], list of target annotations, boxes (T,4) integer xyxy)`."""
import torch

from . import ops, production, datautils


def clip_boxes_to_image(boxes, size):
    """torchvision.ops.clip_boxes_to_image: x in [0,W], y in [0,H]."""
    h, w = size
    b = boxes.clone()
    b[:, 0::2] = b[:, 0::2].clamp(min=0, max=w)
    b[:, 1::2] = b[:, 1::2].clamp(min=0, max=h)
    return b


@torch.no_grad()
def eval_dihe(encoder, sampleset, testset, batch_size, num_workers=0, k=(1,), verbose=False):
    classifier = production.Classifier(encoder, sampleset, batch_size=batch_size, num_workers=num_workers, k=max(k))
    total, correct = 0, {knn: 0 for knn in k}
    missed, confusion, per_ann = {}, {}, {}
    for i, (img, target_anns, boxes) in enumerate(testset):
        if verbose and i % 10 == 0:
            print(f'{i}...')
        boxes = clip_boxes_to_image(boxes, (img.shape[1], img.shape[2]))
        crops = ops.crop_resize(img.to(device=classifier.device, dtype=torch.float32).contiguous(),
                                boxes.to(classifier.device), datautils.CLASSIFICATION_IMAGE_SIZE, mode=0)
        pred_anns = classifier.classify(crops)
        total += len(target_anns)
        for want, got in zip(target_anns, pred_anns):
            per_ann[want] = per_ann.get(want, 0) + 1
            for knn in k:
                correct[knn] += int(want in got[:knn])
            if want != got[0]:
                missed[want] = missed.get(want, 0) + 1
                confusion.setdefault(want, {})
                confusion[want][got[0]] = confusion[want].get(got[0], 0) + 1
    accuracy = {knn: c / total for knn, c in correct.items()}
    if verbose:
        print(f'Total annotations: {total}, Correctly guessed: {correct}, Accuracy: {accuracy}')
        worst = sorted(((v / per_ann[a], v, a) for a, v in missed.items()), reverse=True)[:10]
        print('Most missed: ' + ', '.join(f'{a} ({n}, {p * 100} %)' for p, n, a in worst))
    return accuracy
