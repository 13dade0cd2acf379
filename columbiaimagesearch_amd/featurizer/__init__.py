"""Featurizer plugin surface of cufacesearch, computing on the MI355X.
reference: cufacesearch/cufacesearch/featurizer/generic_featurizer.py, featsio.py"""
from .generic_featurizer import GenericFeaturizer, get_feat_size, get_featurizer
from .featsio import featB64decode, featB64encode, get_feat_dtype, normfeatB64encode
from .sbhip_img_featurizer import SentiBankHIPImgFeaturizer, SentiBankNet
from .dlibhip_featurizer import DLibFaceNet, DLibHIPFeaturizer

# Version of the forward kernels' float32 SUMMATION ORDER (csrc/cnn.hip).  Descriptors of different versions agree to ~1e-6 relative but
# not bit for bit, so a descriptor near a quantiser boundary may encode differently: an index is built with ONE version (re-encode, or
# accept the boundary flips, when it changes -- INTEGRATION.md).  1: rounds 1-4 (fc layers' K split 4); 2: round 5 on (K split 8, fc tiles
# 64 x 128).  tests/test_cnn_hip_parity.py::test_feature_bits_are_pinned fails when the bits change without a bump.
FEATURE_KERNEL_VERSION = 2

__all__ = ["FEATURE_KERNEL_VERSION", "GenericFeaturizer", "get_featurizer", "get_feat_size", "get_feat_dtype", "featB64encode", "featB64decode",
           "normfeatB64encode", "SentiBankHIPImgFeaturizer", "SentiBankNet", "DLibHIPFeaturizer", "DLibFaceNet"]
