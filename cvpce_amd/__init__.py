"""cvpce_amd -- the MI355X-native (gfx950) inference hot path of laitalaj/cvpce:
shelf image -> GLN detect -> RoI crop -> MAC-VGG16 embed -> cosine NN match.

Drop-in Python surface (same names / argument meaning as the reference):
    cvpce_amd.models.proposals      gln, gln_backbone, GaussianLayerNetwork
    cvpce_amd.models.classification MACVGG, macvgg_embedder, distance, nearest_neighbors
    cvpce_amd.production            ProposalGenerator, Classifier, PlanogramEvaluator
    cvpce_amd.datautils             resize_for_classification, CLASSIFICATION_IMAGE_SIZE
    cvpce_amd.utils                 scale_to_tanh, scale_from_tanh, trim_module_prefix
underneath: include/cvpce_amd.h (C ABI) -> cvpce_amd/csrc/*.hip (hand-written HIP kernels).
"""
from . import _lib  # noqa: F401  (raises HipLibraryMissing when the HIP library is not built)

__all__ = ['models', 'production', 'datautils', 'utils', 'ops']
