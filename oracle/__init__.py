"""CPU oracle for the cvpce hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A literal fp32 restatement, in plain torch CPU ops (one F.conv2d per conv,
explicit F.interpolate calls), of the reference's inference path

    shelf image -> GLN detect -> RoI crop -> MACVGG embed -> cosine NN match

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import this package; nothing under cvpce_amd/ does (the product path fails
loudly when the HIP library is missing -- there is no CPU fallback).

Pinning status:
  * pinned by golden vectors made from the reference's own code
    (tests/golden/make_golden.py): Gaussian head (oracle.gln.gaussian_*),
    distance / nearest_neighbors (oracle.match), metrics (oracle.metrics).
  * PARITY UNPINNED: everything whose arithmetic lives in torchvision 0.9
    (pinned at /root/reference/environment.yml:7-8, absent offline):
    GeneralizedRCNNTransform, ResNet-50/FrozenBN, FPN + LastLevelP6P7,
    RetinaNetHead, AnchorGenerator, postprocess_detections, nms, VGG16
    features, ttf.resize / ttf.normalize.  Restated from the published
    torchvision 0.9 semantics; each constant is named.
"""
