"""Featurizer plugin surface of cufacesearch, computing on the MI355X.
reference: cufacesearch/cufacesearch/featurizer/generic_featurizer.py, featsio.py"""
from .generic_featurizer import GenericFeaturizer, get_feat_size, get_featurizer
from .featsio import featB64decode, featB64encode, get_feat_dtype, normfeatB64encode
from .sbhip_img_featurizer import SentiBankHIPImgFeaturizer, SentiBankNet
from .dlibhip_featurizer import DLibFaceNet, DLibHIPFeaturizer

__all__ = ["GenericFeaturizer", "get_featurizer", "get_feat_size", "get_feat_dtype", "featB64encode", "featB64decode",
           "normfeatB64encode", "SentiBankHIPImgFeaturizer", "SentiBankNet", "DLibHIPFeaturizer", "DLibFaceNet"]
