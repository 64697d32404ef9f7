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
# The HIP library is loaded by `cvpce_amd.ops` (and therefore by everything that runs a kernel: models, production, the
# eval harnesses, the CLI): importing any of those without a built libcvpce_hip.so raises `HipLibraryMissing` -- there is
# no CPU fallback.  The host-only modules (metrics, planograms, planogram_adapters, datautils readers, dist, defaults) never
# call a kernel and import without it.

__all__ = ['models', 'production', 'datautils', 'utils', 'ops']
