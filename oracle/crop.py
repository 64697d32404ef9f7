"""Oracle: RoI crop + pad-to-square + bilinear resize (the K9 stage).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
/root/reference/cvpce/datautils.py:232-239 and cvpce/production.py:16-20;
`ttf.resize` on a tensor in torchvision 0.9 is F.interpolate(bilinear,
align_corners=False) WITHOUT antialias ("parity unpinned", Appendix A).
"""
import torch
import torch.nn.functional as F

CLASSIFICATION_IMAGE_SIZE = 256
PAD_VALUE = 0.5


def resize_for_classification(img):
    _, h, w = img.shape
    larger = max(w, h)
    res = torch.full((3, larger, larger), PAD_VALUE, dtype=img.dtype)
    res[:, 0:h, 0:w] = img
    return F.interpolate(res[None], size=(CLASSIFICATION_IMAGE_SIZE, CLASSIFICATION_IMAGE_SIZE),
                         mode='bilinear', align_corners=False)[0]


def crop_boxes(image, boxes):
    """production.py:20 -- boxes.to(long) truncates toward zero; crop from the ORIGINAL image.

    Python slicing semantics are kept (a slice past the border is clamped by torch).
    Degenerate (zero-area) crops make the reference raise; the oracle raises too.
    """
    if not len(boxes):
        return torch.empty((0, 3, CLASSIFICATION_IMAGE_SIZE, CLASSIFICATION_IMAGE_SIZE))
    out = []
    for x1, y1, x2, y2 in boxes.to(dtype=torch.long).tolist():
        crop = image[:, y1:y2, x1:x2]
        if crop.shape[1] == 0 or crop.shape[2] == 0:
            raise ValueError(f'degenerate crop {(x1, y1, x2, y2)}')
        out.append(resize_for_classification(crop))
    return torch.stack(out)


def scale_to_tanh(t):
    """cvpce/utils.py:280-281"""
    return t * 2 - 1
