"""Feature dtype + base64 transport, byte-compatible with cufacesearch/cufacesearch/featurizer/featsio.py."""
import base64

import numpy as np


def featB64encode(feat):
    """raw little-endian bytes of the array, base64 (reference: featsio.py:4-11)"""
    return base64.b64encode(np.ascontiguousarray(feat))


def normfeatB64encode(feat):
    """L2-normalise, then base64 (reference: featsio.py:13-22)"""
    return featB64encode(feat / np.linalg.norm(feat))


def get_feat_dtype(feat_type):
    """sbpycaffe / sbcmdline / sbhip -> float32, dlib -> float64 (reference: featsio.py:24-39)"""
    if feat_type in ("sbpycaffe", "sbcmdline", "sbhip", "float32"):
        return np.float32
    if feat_type in ("dlib", "float64"):
        return np.float64
    raise ValueError("[featsio.get_feat_dtype: error] Unkown feature type: {}".format(feat_type))


def featB64decode(feat_B64, feat_type=None):
    """reference: featsio.py:41-54"""
    return np.frombuffer(base64.b64decode(feat_B64), dtype=get_feat_dtype(feat_type))
