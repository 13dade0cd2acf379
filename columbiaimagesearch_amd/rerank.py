"""Exact re-ranking of search results with the original features kept resident in HBM.

The reference fetches the features of the first ``rerank_nb`` results from HBase and replaces the ADC distance by the
true L2 distance (cufacesearch/cufacesearch/searcher/searcher_lopqhbase.py:864-912 and :975-1017):

    results = results[:min(rerank_nb, len(results))]
    dist    = np.linalg.norm(normed_feat - res_fts[pos])     # NOT squared; ADC distances are squared (:887,:998)
    (a result whose feature is missing keeps its ADC distance, :889-893)
    keep if not filter_near_dup or dist <= near_dup_th; only results with index < max_returned (index BEFORE the re-order)
    order = np.argsort(dists)

1M x 4096 float32 features are 16 GB: they fit the 288 GB of one MI355X, so the fetch becomes a gather in HBM.
"""
import numpy as np

from . import _lib


class ResidentFeatures(object):
    """Features [n, D] (float32 or float64 torch tensor on the GPU) + the ids of their rows."""

    def __init__(self, feats, ids=None):
        import torch
        if not (feats.is_cuda and feats.is_contiguous() and feats.dim() == 2 and feats.dtype in (torch.float32, torch.float64)):
            raise ValueError("feats must be a contiguous float32/float64 [n, D] tensor on the GPU")
        self.feats = feats
        self._row = None if ids is None else {k: i for i, k in enumerate(ids)}

    def rows_of(self, ids):
        """Feature row of every id ([nq, L] array-like; -1 where the id is unknown or negative)."""
        ids = np.asarray(ids)
        if self._row is None:
            rows = ids.astype(np.int64).copy()
            rows[(rows < 0) | (rows >= self.feats.shape[0])] = -1
            return rows
        return np.array([[self._row.get(k.item() if hasattr(k, "item") else k, -1) for k in r] for r in ids], dtype=np.int64).reshape(ids.shape)

    def distances_dev(self, q, rows):
        """True L2 distances [nq, L] (float64 tensor; NaN where rows < 0) of queries q [nq, D] to feats[rows]."""
        import torch
        if not (q.is_cuda and q.is_contiguous() and q.dtype == self.feats.dtype and q.shape[1] == self.feats.shape[1]):
            raise ValueError("q must be a contiguous tensor on the GPU with the features' dtype and width")
        rows = rows if torch.is_tensor(rows) else torch.as_tensor(np.ascontiguousarray(rows, dtype=np.int64))
        rows = rows.to(q.device).contiguous()
        nq, L = int(rows.shape[0]), int(rows.shape[1])
        out = torch.empty((nq, L), dtype=torch.float64, device=q.device)
        code = _lib.CIS_F32 if q.dtype == torch.float32 else _lib.CIS_F64
        _lib.check(_lib.lib().cis_rerank_dev(self.feats.data_ptr(), code, int(self.feats.shape[0]), int(self.feats.shape[1]),
                                             q.data_ptr(), nq, rows.data_ptr(), L, out.data_ptr(),
                                             torch.cuda.current_stream(q.device).cuda_stream))
        return out

    def rerank(self, q, ids, adc_dists, rerank_nb=None, max_returned=None, near_dup_th=None):
        """Re-rank the results of a batch: ids / adc_dists [nq, L] (ids < 0 or NaN distance = no result).
        Returns per query (ids, dists) lists in the reference's final order."""
        ids = np.asarray(ids)
        adc = np.asarray(adc_dists, dtype=np.float64)
        nq, L = ids.shape
        nb = L if rerank_nb is None else min(int(rerank_nb), L)
        valid = ~np.isnan(adc[:, :nb]) if ids.dtype.kind not in "iu" else (ids[:, :nb] >= 0)
        rows = self.rows_of(ids[:, :nb])
        rows[~valid] = -1
        true_d = self.distances_dev(q, rows).cpu().numpy()
        out = []
        for qi in range(nq):
            d = np.where(np.isnan(true_d[qi]), adc[qi, :nb], true_d[qi])
            keep = [i for i in range(nb) if valid[qi, i] and (near_dup_th is None or d[i] <= near_dup_th)
                    and (not max_returned or i < max_returned)]
            order = np.argsort(d[keep], axis=0, kind="stable") if keep else []
            out.append(([ids[qi, keep[j]] for j in order], [float(d[keep[j]]) for j in order]))
        return out
