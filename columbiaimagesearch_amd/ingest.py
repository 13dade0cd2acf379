"""Batched ingest: CNN descriptors -> L2 normalisation -> LOPQ codes -> index, all on the GPU (BASELINE config C5).

Counterpart of the per-image chain of the reference, with the same per-element semantics:
  featurize            cufacesearch/cufacesearch/extractor/generic_extractor.py:238,248 (one image per call)
  L2 normalisation     cufacesearch/cufacesearch/featurizer/featsio.py:13-22 (feat / np.linalg.norm(feat), in the feature dtype)
  compute_codes        cufacesearch/cufacesearch/searcher/searcher_lopqhbase.py:482-524 (model.predict per feature)
  add_codes_from_dict  lopq/lopq/search.py:275-283
Nothing leaves the GPU: the codes are merged into the HBM-resident index by kernels (csrc/lopq_index.hip); images come in
as preprocessed batches (decoding / resizing stays host code).
"""
import numpy as np

from . import _lib


def l2_normalize_dev(feats):
    """Row-wise feat / ||feat|| in the tensor's dtype, in place on a contiguous tensor (cis_l2_normalize_dev: one wave per
    row; zero rows stay zero: the reference would emit NaNs for them).  Returns the tensor."""
    import torch
    if not (feats.is_cuda and feats.dim() == 2 and feats.dtype in (torch.float32, torch.float64)):
        raise ValueError("feats must be a float32 / float64 [n, d] tensor on the GPU")
    if not feats.is_contiguous():
        feats = feats.contiguous()
    _lib.check(_lib.lib().cis_l2_normalize_dev(feats.data_ptr(), _lib.CIS_F32 if feats.dtype == torch.float32 else _lib.CIS_F64,
                                               int(feats.shape[0]), int(feats.shape[1]),
                                               torch.cuda.current_stream(feats.device).cuda_stream))
    return feats


class BatchIngest(object):
    """net: SentiBankNet / DLibFaceNet (forward_dev), model: LOPQModel[PCA] (predict_batch_dev),
    searcher: LOPQSearcherHIP, ShardedSearcher / GridSearcher (device insert) or any searcher with add_codes_array
    (LOPQSearcherLMDB: host key / value store)."""

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
        """Encode one batch and insert it; ids default to consecutive integers.  Returns the number of new items.
        Codes and ids stay in HBM: the insert is the device-side merge of csrc/lopq_index.hip (through the all-to-all of
        distributed.ShardedSearcher.add_codes_routed_dev when the searcher is sharded)."""
        import torch
        coarse, fine = self.encode_batch_dev(x)
        return self._insert_codes(coarse, fine, ids)

    def _insert_codes(self, coarse, fine, ids=None):
        import torch
        n = int(coarse.shape[0])
        s = self.searcher
        if not (hasattr(s, "add_codes_routed_dev") or hasattr(s, "add_codes_dev")):
            # a searcher without a device entry point (LOPQSearcherLMDB keeps its key / value store on the host): the codes go
            # through add_codes_array with the caller's ids as they are (any hashable, like the reference's add_codes)
            if ids is None:
                ids = np.arange(self.nb_ingested, self.nb_ingested + n, dtype=np.int64)
            elif hasattr(ids, "is_cuda"):
                ids = ids.cpu().numpy()
            before = s.get_nb_indexed()
            s.add_codes_array(coarse.cpu().numpy().view(np.uint16), fine.cpu().numpy(), ids)
            self.nb_ingested += n
            return int(s.get_nb_indexed() - before)
        if ids is None:
            ids = torch.arange(self.nb_ingested, self.nb_ingested + n, dtype=torch.int64, device=coarse.device)
        elif not hasattr(ids, "is_cuda"):
            try:
                ids_h = np.ascontiguousarray(ids, dtype=np.int64)
            except (TypeError, ValueError):
                ids_h = None
            if ids_h is None or ids_h.shape != (n,):
                # non-integer ids (the reference's sha1 strings): the Python mirror maps them to slots, codes take the host entry point
                added = s.add_codes_array(coarse.cpu().numpy().view(np.uint16), fine.cpu().numpy(), list(ids))
                self.nb_ingested += n
                return added
            ids = torch.as_tensor(ids_h).to(coarse.device)
        if hasattr(s, "add_codes_routed_dev"):
            added = s.add_codes_routed_dev(coarse, fine, ids)
        else:
            added = s.add_codes_dev(coarse, fine, ids)
        if isinstance(added, tuple):  # LOPQSearcherHIP.add_codes_dev -> (added, skipped); the sharded forms return the count
            added = added[0]
        self.nb_ingested += n
        return int(added)

    def ingest_batches(self, batches, ids=None, lanes=3):
        """Ingest a sequence of preprocessed batches with the CNN forwards of the NEXT batches in flight while the current one is
        encoded and inserted (round 5): `lanes` handles on the same weights (net.view()), each on its own stream, run forward +
        normalisation; encode and insert follow on the caller's stream in batch order, so ids, insertion order and dedup are those
        of calling ingest_batch batch by batch.  batches: an iterable of GPU tensors (each is kept alive, and its block recorded on the lane stream, until its
        forward has been waited for: a generator may drop them); ids: None or an iterable of per-batch id tensors / arrays.  Returns the list of per-batch new-item counts."""
        import torch
        from collections import deque
        cur = torch.cuda.current_stream()
        lanes = max(1, int(lanes))
        if not hasattr(self.net, "view"):
            lanes = 1
        if getattr(self, "_lanes", None) is None or len(self._lanes) != lanes:
            for h, _ in (getattr(self, "_lanes", None) or [])[1:]:
                h.close()
            from .streams import lane_streams   # streams on different hardware pipes, made once per process (streams.py)
            ls = lane_streams(lanes)
            self._lanes = [(self.net, ls[0])] + [(self.net.view(), ls[i]) for i in range(1, lanes)]
        ids_it = iter(ids) if ids is not None else None
        out, queue = [], deque()

        def finish():
            feats, ev, bid, _x = queue.popleft()   # (_x: the input batch, kept alive until its forward has been waited for)
            cur.wait_event(ev)
            feats.record_stream(cur)
            if self.feat_dtype is not None and feats.dtype != self.feat_dtype:
                feats = feats.to(self.feat_dtype)
            coarse, fine = self.model.predict_batch_dev(l2_normalize_dev(feats).contiguous())
            out.append(self._insert_codes(coarse, fine, bid))
        for i, x in enumerate(batches):
            net, stream = self._lanes[i % lanes]
            stream.wait_stream(cur)  # the batch was produced on the caller's stream
            with torch.cuda.stream(stream):
                feats = net.forward_dev(x)
                ev = torch.cuda.Event()
                ev.record(stream)
            # the lane stream reads x: the caching allocator must not hand its block to the caller's stream before that read is done
            # (a generator that drops x_i after yielding it would otherwise have x_{i+1} written over it)
            if hasattr(x, "record_stream"):
                x.record_stream(stream)
            queue.append((feats, ev, next(ids_it) if ids_it is not None else None, x))
            if len(queue) >= lanes:
                finish()
        while queue:
            finish()
        return out

    def close(self):
        for h, _ in (getattr(self, "_lanes", None) or [])[1:]:
            h.close()
        self._lanes = None
