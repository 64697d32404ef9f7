"""The hot-path member of /root/reference/cvpce/datautils.py (dataset readers are out of scope)."""
import torch

from . import ops

CLASSIFICATION_IMAGE_SIZE = 256


def resize_for_classification(img):
    """datautils.py:234-239: pad (3,h,w) to a square with 0.5 (top-left anchored), bilinear -> 256x256.

    Runs the K9 HIP gather kernel (the whole image is the crop box)."""
    if not img.is_cuda:
        raise RuntimeError('resize_for_classification runs on an MI355X (HIP) device only (no CPU fallback)')
    _, h, w = img.shape
    box = torch.tensor([[0.0, 0.0, float(w), float(h)]], device=img.device)
    return ops.crop_resize(img.to(torch.float32).contiguous(), box, CLASSIFICATION_IMAGE_SIZE, mode=0)[0]
