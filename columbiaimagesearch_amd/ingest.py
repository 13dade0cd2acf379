"""Batched ingest: CNN descriptors -> L2 normalisation -> LOPQ codes -> index, all on the GPU (BASELINE config C5).

Counterpart of the per-image chain of the reference, with the same per-element semantics:
  featurize            cufacesearch/cufacesearch/extractor/generic_extractor.py:238,248 (one image per call)
  L2 normalisation     cufacesearch/cufacesearch/featurizer/featsio.py:13-22 (feat / np.linalg.norm(feat), in the feature dtype)
  compute_codes        cufacesearch/cufacesearch/searcher/searcher_lopqhbase.py:482-524 (model.predict per feature)
  add_codes_from_dict  lopq/lopq/search.py:275-283
Only the (coarse, fine) codes -- 4 + M bytes per descriptor -- leave the GPU; images come in as preprocessed batches
(decoding / resizing stays host code).
"""
import numpy as np

from . import _lib


def l2_normalize_dev(feats):
    """Row-wise feat / ||feat|| in the tensor's dtype (zero rows stay zero: the reference would emit NaNs for them)."""
    import torch
    nrm = torch.linalg.vector_norm(feats, dim=1, keepdim=True)
    return feats / torch.where(nrm > 0, nrm, torch.ones_like(nrm))


class BatchIngest(object):
    """net: SentiBankNet / DLibFaceNet (forward_dev), model: LOPQModel[PCA] (predict_batch_dev),
    searcher: LOPQSearcherHIP or ShardedSearcher (add_codes_array)."""

    def __init__(self, net, model, searcher, feat_dtype=None):
        self.net, self.model, self.searcher = net, model, searcher
        self.feat_dtype = feat_dtype  # torch dtype the descriptors are cast to before normalisation (dlib: float64)
        self.nb_ingested = 0

    def encode_batch_dev(self, x):
        """x: preprocessed input batch on the GPU -> (coarse uint16-as-int16 [n,2], fine uint8 [n,M]) on the GPU."""
        feats = self.net.forward_dev(x)
        if self.feat_dtype is not None and feats.dtype != self.feat_dtype:
            feats = feats.to(self.feat_dtype)
        return self.model.predict_batch_dev(l2_normalize_dev(feats).contiguous())

    def ingest_batch(self, x, ids=None):
        """Encode one batch and insert it; ids default to consecutive integers.  Returns the number of new items."""
        coarse, fine = self.encode_batch_dev(x)
        n = int(coarse.shape[0])
        if ids is None:
            ids = np.arange(self.nb_ingested, self.nb_ingested + n, dtype=np.int64)
        added = self.searcher.add_codes_array(coarse.cpu().numpy().view(np.uint16), fine.cpu().numpy(), ids)
        self.nb_ingested += n
        return added
