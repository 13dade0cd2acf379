"""LOPQSearcherHIP: the reference's searcher surface over the MI355X index.

Mirror of lopq/lopq/search.py:85-382 of the reference (LOPQSearcherBase / LOPQSearcher): same method
names, arguments and return types -- ``search`` returns ``(list[Result], visited)`` with
``Result(id, code[, dist])`` -- selected in cufacesearch by ``lopq_searcher = "LOPQSearcherHIP"``
next to "LOPQSearcher" / "LOPQSearcherLMDB" (cufacesearch/searcher/searcher_lopqhbase.py:198-222).
Additions: ``search_batch`` and ``add_codes_array`` (whole arrays in, arrays out), which is where
the GPU earns its keep; the per-query ``search`` is the batch of one.

Item ids may be any hashable (the reference uses ``sha1[_bbox]`` strings,
cufacesearch/indexer/hbase_indexer_minimal.py:816): integers travel to the device as they are,
everything else is mapped to an integer slot here.
"""
import os
from collections import namedtuple
from itertools import count

import numpy as np

from .. import _lib
from . import kvlog
from .model import LOPQCode, LOPQModel, LOPQModelPCA, _code_dtype

_SLOT_BASE = 1 << 62  # device ids >= this are slots of non-integer caller ids


def _is_int(x):
    return isinstance(x, (int, np.integer)) and not isinstance(x, (bool, np.bool_))


def multisequence_batch(X, centroids, max_cells=64):
    """First `max_cells` cells of many LOPQ-space vectors at once: (cells [n,max_cells,2] int32, dists
    [n,max_cells]) in multi-sequence order (reference: lopq/lopq/search.py:13-82)."""
    C0, C1 = _lib.as_float_matrix(centroids[0]), _lib.as_float_matrix(centroids[1])
    if C0.dtype != C1.dtype:
        C0, C1 = C0.astype(np.float64), C1.astype(np.float64)
    V, h = C0.shape
    X = _lib.as_float_matrix(X, 2 * h)
    mc = int(min(max_cells, V * V))
    cells = np.empty((X.shape[0], mc, 2), dtype=np.int32)
    dists = np.empty((X.shape[0], mc), dtype=np.float64)
    dt = _lib.c_int(0)
    _lib.check(_lib.lib().cis_multisequence(_lib.ptr(X), _lib.dtype_code(X), _lib.ptr(C0), _lib.ptr(C1), _lib.dtype_code(C0),
                                            X.shape[0], V, h, mc, _lib.ptr(cells), _lib.ptr(dists), _lib.ctypes.byref(dt)))
    return cells, (dists.astype(np.float32) if dt.value == 4 else dists)


def multisequence(x, centroids):
    """Generator of (dist, (c0, c1)) in multi-sequence order, like the reference's (lopq/lopq/search.py:13-82).
    Cells are computed on the GPU in growing prefixes (64, 256, ... up to V*V) as the consumer advances."""
    V = np.asarray(centroids[0]).shape[0]
    done, want = 0, 64
    while done < V * V:
        cells, dists = multisequence_batch(np.asarray(x)[None, :], centroids, max_cells=want)
        for k in range(done, cells.shape[1]):
            yield dists[0, k], (int(cells[0, k, 0]), int(cells[0, k, 1]))
        done = cells.shape[1]
        want = min(want * 4, V * V)


class LOPQSearcherBase(object):
    """Hooks shared by every searcher (reference: lopq/lopq/search.py:85-308)."""

    def __init__(self):
        self.nb_indexed = 0
        self.verbose = 0

    def get_nb_indexed(self):
        return self.nb_indexed

    def add_codes_from_dict(self, codes_dict):
        """reference: lopq/lopq/search.py:275-283 -- {id: [coarse, fine]}"""
        ids = list(codes_dict.keys())
        self.add_codes([codes_dict[k] for k in ids], ids)

    def add_data(self, data, ids=None, num_procs=1):
        """reference: lopq/lopq/search.py:94-108 (num_procs is meaningless on the GPU)"""
        coarse, fine = self.model.predict_batch(data)
        self.add_codes_array(coarse, fine, ids)

    def _add_codes_from_one_file(self, one_file, samples_count):
        """reference: lopq/lopq/search.py:227-243 -- 'id<TAB>[[c0, c1], [f0, ...]]' lines"""
        import ast
        ids, codes = [], []
        with open(one_file, "rt") as inf:
            for line in inf:
                if line.strip():
                    one_id, one_code = line.rstrip("\n").split("\t")
                    parsed = ast.literal_eval(one_code)
                    ids.append(one_id)
                    codes.append((tuple(parsed[0]), tuple(parsed[1])))
        self.add_codes(codes, ids)
        return samples_count + len(ids)

    def add_codes_from_local(self, local_path):
        """reference: lopq/lopq/search.py:245-263 -- one file, or a directory of Spark part-* files"""
        import os
        from glob import glob
        files = [local_path] if os.path.isfile(local_path) else sorted(glob(local_path + "/part-*"))
        n = 0
        for f in files:
            n = self._add_codes_from_one_file(f, n)
        return n

    def add_codes(self, codes, ids=None):
        raise NotImplementedError()

    def get_cell(self, cell):
        raise NotImplementedError()

    # -- the two public hooks of the base class (cufacesearch never calls them; third-party code may) ----------------------
    def get_result_quota(self, x, quota=10):
        """reference: lopq/lopq/search.py:110-135 -> (retrieved items, cells visited).  x is in LOPQ space (the
        reference applies no PCA here either); cells come from the GPU multisequence, items from get_cell."""
        retrieved = []
        visited = 0
        for _, cell in multisequence(x, self.model.Cs):
            retrieved += self.get_cell(cell)
            visited += 1
            if len(retrieved) >= quota:
                break
        return retrieved, visited

    def compute_distances(self, x, items):
        """reference: lopq/lopq/search.py:137-177 -> [(dist, item)]: ADC tables (model.get_subquantizer_distances, HIP)
        memoised per coarse cluster, the M entries of an item summed left to right in float64 like the reference's sum()."""
        memo = [{}, {}]
        results = []
        for item in items:
            coarse, fine = item[1]
            c0, c1 = int(coarse[0]), int(coarse[1])
            if c0 not in memo[0]:
                memo[0][c0] = self.model.get_subquantizer_distances(x, (c0, c1), coarse_split=0)
            if c1 not in memo[1]:
                memo[1][c1] = self.model.get_subquantizer_distances(x, (c0, c1), coarse_split=1)
            tables = memo[0][c0] + memo[1][c1]
            results.append((sum([tables[i][fc] for i, fc in enumerate(fine)]), item))
        return results


class LOPQSearcherHIP(LOPQSearcherBase):
    def __init__(self, model, shard=None):
        """:param model: LOPQModel / LOPQModelPCA (this package's classes)
        :param shard: None, or (rank, world[, owner]) for a cell-sharded index (one per GPU)"""
        super(LOPQSearcherHIP, self).__init__()
        self.model = model
        self._ix = None
        self._shard = shard
        self._slot_of = {}  # non-integer id -> slot
        self._id_of = []    # slot -> id
        self._M = model.M
        self._open()

    # -- handle ----------------------------------------------------------------------------------
    def _open(self):
        out = _lib.c_void_p()
        L = _lib.lib()
        # the index borrows the cis_model* (include/cis_hip.h): hold the owning object for the index's lifetime
        self._model_handle = self.model._handle_obj()
        _lib.check(L.cis_index_create(_lib.ctypes.byref(out), self._model_handle.ptr))
        self._ix = out.value
        self._input_dim = self.model.__dict__["_dims"][0]  # of the parameters this index was built on
        if type(self).default_scan_mode:
            _lib.check(L.cis_index_set_scan_mode(self._ix, type(self).default_scan_mode))
        elif type(self).default_prefilter_only:
            _lib.check(L.cis_index_set_scan_mode(self._ix, 2))
        if self._shard is not None:
            rank, world = self._shard[0], self._shard[1]
            owner = None
            if len(self._shard) > 2 and self._shard[2] is not None:
                owner = np.ascontiguousarray(self._shard[2], dtype=np.int32)
            _lib.check(L.cis_index_set_shard(self._ix, int(rank), int(world), _lib.ptr(owner)))

    def view(self):
        """A second searcher over the SAME index in HBM with its own per-batch workspaces (cis_index_create_view): batches searched
        through the view and through this object can be in flight at once, each on its own stream -- the small kernels of one batch's
        front end fill the tail of the other's scan.  Read-only: inserts go through this object, ordered against the view's searches by
        the caller.  The view shares the id maps and must be closed before this searcher."""
        v = object.__new__(type(self))
        LOPQSearcherBase.__init__(v)
        v.model, v._shard, v._M = self.model, self._shard, self._M
        v._slot_of, v._id_of = self._slot_of, self._id_of
        v._model_handle, v._input_dim = self._model_handle, self._input_dim
        v._base = self  # keeps the base alive
        out = _lib.c_void_p()
        _lib.check(_lib.lib().cis_index_create_view(_lib.ctypes.byref(out), self._ix))
        v._ix = out.value
        v.nb_indexed = self.nb_indexed
        if getattr(self, "_views", None) is None:
            import weakref
            self._views = weakref.WeakSet()
        self._views.add(v)
        return v

    def get_nb_indexed(self):
        base = getattr(self, "_base", None)  # a view reads through to its base (inserts go through the base)
        if base is not None:
            self.nb_indexed = base.get_nb_indexed()
        return self.nb_indexed

    def close(self):
        """Destroys the device index.  Views created from this searcher are closed first (they share its storage; the library
        also refuses searches through a view whose base is gone: include/cis_hip.h:cis_index_create_view)."""
        for v in list(getattr(self, "_views", None) or ()):
            v.close()
        if self._ix:
            _lib.lib().cis_index_destroy(self._ix)
            self._ix = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- ids -------------------------------------------------------------------------------------
    def _device_ids(self, ids, n):
        if ids is None:
            return np.arange(n, dtype=np.int64)  # reference: ids = count() per call, search.py:336-337
        if isinstance(ids, np.ndarray) and ids.dtype.kind in "iu":
            out = np.ascontiguousarray(ids, dtype=np.int64)
        else:
            out = np.empty(n, dtype=np.int64)
            k = -1
            for k, item_id in zip(range(n), ids):
                if _is_int(item_id):
                    out[k] = int(item_id)
                else:
                    slot = self._slot_of.get(item_id)
                    if slot is None:
                        slot = len(self._id_of)
                        self._slot_of[item_id] = slot
                        self._id_of.append(item_id)
                    out[k] = _SLOT_BASE + slot
            if k + 1 != n:
                out = out[:k + 1]
        if out.shape[0] and ((out < 0).any()):
            raise ValueError("integer ids must be >= 0")
        return out

    def _caller_id(self, dev_id):
        dev_id = int(dev_id)
        return self._id_of[dev_id - _SLOT_BASE] if dev_id >= _SLOT_BASE else dev_id

    # -- insert ----------------------------------------------------------------------------------
    def add_codes_array(self, coarse, fine, ids=None, dedup=True):
        """Insert n codes given as arrays (coarse [n,2], fine [n,M]).  Same semantics as add_codes:
        an id already present in the same cell is skipped (reference: lopq/lopq/search.py:349-364)."""
        coarse = np.ascontiguousarray(np.asarray(coarse).reshape(-1, 2), dtype=np.uint16)
        n = coarse.shape[0]
        fine = np.ascontiguousarray(np.asarray(fine).reshape(n, self._M), dtype=np.uint8)
        dev_ids = self._device_ids(ids, n)
        m = min(n, dev_ids.shape[0])  # zip() semantics when ids is shorter
        added = _lib.c_int64(0)
        _lib.check(_lib.lib().cis_index_add(self._ix, _lib.ptr(dev_ids), _lib.ptr(coarse), _lib.ptr(fine), m,
                                            1 if dedup else 0, _lib.ctypes.byref(added)))
        self.nb_indexed = int(_lib.lib().cis_index_size(self._ix))
        return int(added.value)

    def add_codes_dev(self, coarse, fine, ids, dedup=True, cell_delta=None):
        """add_codes_array on tensors that are already in HBM (what predict_batch_dev returned): coarse [n,2] int16/uint16
        bits, fine [n,M] uint8, ids [n] int64 >= 0.  The insert is a stable merge by kernels on the current stream
        (csrc/lopq_index.hip); nothing travels through the host but the accepted count.  Returns (added, skipped):
        items with out-of-range codes or negative ids are skipped and counted ("Could not push code", :365-367).
        cell_delta: optional int64 [V*V] tensor that receives the per-cell accepted counts of this call."""
        import torch
        n = int(coarse.shape[0])
        if not (coarse.is_cuda and fine.is_cuda and ids.is_cuda and coarse.is_contiguous() and fine.is_contiguous()
                and ids.is_contiguous()):
            raise ValueError("coarse, fine and ids must be contiguous tensors on the GPU")
        if coarse.dtype not in (torch.int16, torch.uint16) or tuple(coarse.shape) != (n, 2):
            raise ValueError("coarse must be a [n, 2] tensor of 16-bit codes")
        if fine.dtype != torch.uint8 or tuple(fine.shape) != (n, self._M) or ids.dtype != torch.int64 or tuple(ids.shape) != (n,):
            raise ValueError("fine must be uint8 [n, %d] and ids int64 [n]" % self._M)
        added, bad = _lib.c_int64(0), _lib.c_int64(0)
        _lib.check(_lib.lib().cis_index_add_dev(self._ix, ids.data_ptr(), coarse.data_ptr(), fine.data_ptr(), n, 1 if dedup else 0,
                                                _lib.ctypes.byref(added), _lib.ctypes.byref(bad),
                                                None if cell_delta is None else cell_delta.data_ptr(),
                                                torch.cuda.current_stream(coarse.device).cuda_stream))
        self.nb_indexed = int(_lib.lib().cis_index_size(self._ix))
        if bad.value:
            print("Could not push {} codes (out of range for this model, or negative ids).".format(bad.value))
        return int(added.value), int(bad.value)

    def add_codes(self, codes, ids=None):
        """reference: lopq/lopq/search.py:325-369.  codes: iterable of (coarse, fine) tuples.  Items
        that cannot be pushed are reported and skipped, like the reference does (:365-367)."""
        V, K, M = self.model.V, self.model.subquantizer_clusters, self._M
        good_c, good_f, good_ids = [], [], []
        id_iter = count() if ids is None else iter(ids)
        for item_id, code in zip(id_iter, codes):
            try:
                c = (int(code[0][0]), int(code[0][1]))
                f = [int(v) for v in code[1]]
                if not (0 <= c[0] < V and 0 <= c[1] < V) or len(f) != M or min(f) < 0 or max(f) >= K:
                    raise ValueError("code out of range for this model")
                if _is_int(item_id) and int(item_id) < 0:
                    raise ValueError("negative id")
            except Exception as inst:
                print("Could not push code {}. ({}: {})".format(code, type(inst), inst))
                continue
            good_c.append(c)
            good_f.append(f)
            good_ids.append(item_id)
        if good_c:
            self.add_codes_array(np.array(good_c, dtype=np.uint16), np.array(good_f, dtype=np.uint8), good_ids)

    def get_cell(self, cell):
        """reference: lopq/lopq/search.py:372-382 -> [(id, LOPQCode), ...] in insertion order"""
        c0, c1 = int(cell[0]), int(cell[1])
        L = _lib.lib()
        n = _lib.c_int64(0)
        _lib.check(L.cis_index_get_cell(self._ix, c0, c1, 0, None, None, _lib.ctypes.byref(n)))
        cap = int(n.value)
        if cap == 0:
            return []
        ids = np.empty(cap, dtype=np.int64)
        fine = np.empty((cap, self._M), dtype=np.uint8)
        _lib.check(L.cis_index_get_cell(self._ix, c0, c1, cap, _lib.ptr(ids), _lib.ptr(fine), _lib.ctypes.byref(n)))
        ids, fine = ids[:int(n.value)], fine[:int(n.value)]  # a cell of another shard holds nothing here
        ct = _code_dtype(self.model.V)
        coarse = (ct(c0), ct(c1))
        return [(self._caller_id(i), LOPQCode(coarse, tuple(f))) for i, f in zip(ids, fine)]

    # -- search ----------------------------------------------------------------------------------
    def search_batch(self, X, quota=10, limit=None, with_codes=False):
        """Search many queries in one call.

        :returns: dict with ``ids`` [nq,L] int64 device ids (-1 padded; map with ``caller_ids``),
            ``dists`` [nq,L] float64 squared ADC distances (NaN padded), ``n_found`` [nq],
            ``visited`` [nq] and, if with_codes, ``cells`` / ``pos`` locating every result.
        """
        X = _lib.as_float_matrix(X, self._input_dim)
        nq = X.shape[0]
        L = int(quota) if limit is None else int(limit)
        L = max(L, 0)
        ids = -np.ones((nq, L), dtype=np.int64)
        dists = np.full((nq, L), np.nan)
        n_found = np.zeros(nq, dtype=np.int32)
        visited = np.zeros(nq, dtype=np.int32)
        cells = -np.ones((nq, L), dtype=np.int32) if with_codes else None
        pos = np.zeros((nq, L), dtype=np.uint32) if with_codes else None
        _lib.check(_lib.lib().cis_index_search(self._ix, _lib.ptr(X), _lib.dtype_code(X), nq, int(quota),
                                               -1 if limit is None else int(limit), _lib.ptr(ids), _lib.ptr(dists),
                                               _lib.ptr(n_found), _lib.ptr(visited), _lib.ptr(cells), _lib.ptr(pos)))
        out = {"ids": ids, "dists": dists, "n_found": n_found, "visited": visited}
        if with_codes:
            out["cells"], out["pos"] = cells, pos
        return out

    def search_batch_async(self, X, quota=10, limit=None, out=None):
        """search_batch in two halves (include/cis_hip.h:cis_index_search_async): copy-in, search and copy-out are enqueued on this
        handle's own stream and the call returns; `search_wait()` returns the result dict once it has landed.  X and the arrays of
        `out` (a dict like search_batch's result: ids, dists, n_found, visited) should live in pinned memory (`_lib.pinned_empty`) so
        that the copies are DMA transfers; they must not be touched before search_wait.  One batch in flight per searcher -- views
        (`view()`) give several."""
        X = _lib.as_float_matrix(X, self._input_dim)
        nq = X.shape[0]
        L = max(int(quota) if limit is None else int(limit), 0)
        if out is None:
            out = {"ids": _lib.pinned_empty((nq, L), np.int64), "dists": _lib.pinned_empty((nq, L), np.float64),
                   "n_found": _lib.pinned_empty((nq,), np.int32), "visited": _lib.pinned_empty((nq,), np.int32)}
        for k, dt, shp in (("ids", np.int64, (nq, L)), ("dists", np.float64, (nq, L)), ("n_found", np.int32, (nq,)), ("visited", np.int32, (nq,))):
            a = out[k]
            if a.dtype != dt or tuple(a.shape) != shp or not a.flags["C_CONTIGUOUS"]:
                raise ValueError("out[%r] must be a C-contiguous %s array of shape %r" % (k, np.dtype(dt).name, shp))
        self._async = (X, out)  # the buffers stay referenced until search_wait
        _lib.check(_lib.lib().cis_index_search_async(self._ix, _lib.ptr(X), _lib.dtype_code(X), nq, int(quota),
                                                     -1 if limit is None else int(limit), _lib.ptr(out["ids"]), _lib.ptr(out["dists"]),
                                                     _lib.ptr(out["n_found"]), _lib.ptr(out["visited"]), None, None))
        return out

    def search_wait(self):
        """Blocks until the batch of the last search_batch_async has landed; returns its result dict (None when nothing is in flight)."""
        _lib.check(_lib.lib().cis_index_search_wait(self._ix))
        pend, self._async = getattr(self, "_async", None), None
        return pend[1] if pend else None

    def caller_ids(self, dev_ids):
        """Map device ids of search_batch back to the ids given to add_codes."""
        return [self._caller_id(i) for i in dev_ids if i >= 0]

    def _codes_of(self, cells, pos):
        n = cells.shape[0]
        fine = np.empty((n, self._M), dtype=np.uint8)
        if n:
            c = np.ascontiguousarray(cells, dtype=np.int32)
            p = np.ascontiguousarray(pos, dtype=np.uint32)
            _lib.check(_lib.lib().cis_index_get_codes(self._ix, _lib.ptr(c), _lib.ptr(p), n, _lib.ptr(fine)))
        V = self.model.V
        ct = _code_dtype(V)
        return [LOPQCode((ct(int(cl) // V), ct(int(cl) % V)), tuple(f)) for cl, f in zip(cells, fine)]

    def search(self, x, quota=10, limit=None, with_dists=False):
        """reference: lopq/lopq/search.py:179-224 -> (list[Result(id, code[, dist])], visited).
        PCA is applied when the model has PCA parameters (the reference tests the exact class,
        :198, which a subclass would silently fail; SURVEY.md section 8b gotcha i)."""
        r = self.search_batch(np.asarray(x)[None, :], quota=quota, limit=limit, with_codes=True)
        n = int(r["n_found"][0])
        ids = [self._caller_id(i) for i in r["ids"][0, :n]]
        codes = self._codes_of(r["cells"][0, :n], r["pos"][0, :n])
        if with_dists:
            Result = namedtuple("Result", ["id", "code", "dist"])
            results = [Result(i, c, d) for i, c, d in zip(ids, codes, r["dists"][0, :n])]
        else:
            Result = namedtuple("Result", ["id", "code"])
            results = [Result(i, c) for i, c in zip(ids, codes)]
        return results, int(r["visited"][0])


    # -- device-resident entry points (torch tensors are only the memory/stream plumbing) ---------
    def _dev_args(self, q, quota, limit):
        import torch
        if not (q.is_cuda and q.is_contiguous() and q.dim() == 2 and q.shape[1] == self._input_dim):
            raise ValueError("q must be a contiguous [nq, %d] tensor on the GPU" % self._input_dim)
        if q.dtype not in (torch.float32, torch.float64):
            raise ValueError("q must be float32 or float64")
        L = int(quota) if limit is None else int(limit)
        code = _lib.CIS_F32 if q.dtype == torch.float32 else _lib.CIS_F64
        return max(L, 0), code, torch.cuda.current_stream(q.device).cuda_stream

    def search_batch_dev(self, q, quota=10, limit=None, out=None, with_codes=False):
        """search_batch on tensors already in HBM; returns torch tensors, does not synchronise
        (apart from the small plan read-back inside the library)."""
        import torch
        L, code, stream = self._dev_args(q, quota, limit)
        nq = q.shape[0]
        if out is None:
            out = {"ids": torch.empty((nq, L), dtype=torch.int64, device=q.device),
                   "dists": torch.empty((nq, L), dtype=torch.float64, device=q.device),
                   "n_found": torch.empty(nq, dtype=torch.int32, device=q.device),
                   "visited": torch.empty(nq, dtype=torch.int32, device=q.device)}
            if with_codes:
                out["cells"] = torch.empty((nq, L), dtype=torch.int32, device=q.device)
                out["pos"] = torch.empty((nq, L), dtype=torch.int32, device=q.device)
        cells = out["cells"].data_ptr() if "cells" in out else None
        pos = out["pos"].data_ptr() if "pos" in out else None
        _lib.check(_lib.lib().cis_index_search_dev(self._ix, q.data_ptr(), code, nq, int(quota),
                                                   -1 if limit is None else int(limit), out["ids"].data_ptr(),
                                                   out["dists"].data_ptr(), out["n_found"].data_ptr(),
                                                   out["visited"].data_ptr(), cells, pos, stream))
        return out

    def search_partial_dev(self, q, quota=10, limit=None):
        """This shard's ranked hits: (hits uint8 [nq, L, 32] viewable as cis_hit, visited [nq])."""
        import torch
        L, code, stream = self._dev_args(q, quota, limit)
        nq = q.shape[0]
        hits = torch.empty((nq, L, _lib.HIT_DTYPE.itemsize), dtype=torch.uint8, device=q.device)
        visited = torch.empty(nq, dtype=torch.int32, device=q.device)
        _lib.check(_lib.lib().cis_index_search_partial_dev(self._ix, q.data_ptr(), code, nq, int(quota),
                                                           -1 if limit is None else int(limit), hits.data_ptr(),
                                                           visited.data_ptr(), stream))
        return hits, visited

    def query_owners_dev(self, q, quota=10):
        """Routed cell-sharded search, step 0 (cis_index_query_owners_dev): (mask int64 [nq] -- bit r: rank r owns a non-empty cell
        the query visits --, visited int32 [nq]) for queries in HBM; no synchronisation."""
        import torch
        _, code, stream = self._dev_args(q, quota, 1)
        nq = q.shape[0]
        mask = torch.empty(nq, dtype=torch.int64, device=q.device)
        visited = torch.empty(nq, dtype=torch.int32, device=q.device)
        _lib.check(_lib.lib().cis_index_query_owners_dev(self._ix, q.data_ptr(), code, nq, int(quota), mask.data_ptr(),
                                                         visited.data_ptr(), stream))
        return mask, visited

    def search_partial_packed_dev(self, q, quota=10, limit=None):
        """This shard's ranked hits packed for the exchange: dict(packed int64 [nq*L, 4] (first `total` rows valid),
        cnt int32 [nq], off int64 [nq], total int64 [1], visited int32 [nq], L)."""
        import torch
        L, code, stream = self._dev_args(q, quota, limit)
        nq = q.shape[0]
        dev = q.device
        out = {"packed": torch.empty((max(nq * L, 1), 4), dtype=torch.int64, device=dev),
               "cnt": torch.empty(nq, dtype=torch.int32, device=dev), "off": torch.empty(nq, dtype=torch.int64, device=dev),
               "total": torch.empty(1, dtype=torch.int64, device=dev), "visited": torch.empty(nq, dtype=torch.int32, device=dev),
               "L": L}
        _lib.check(_lib.lib().cis_index_search_partial_packed_dev(
            self._ix, q.data_ptr(), code, nq, int(quota), -1 if limit is None else int(limit), out["packed"].data_ptr(),
            out["cnt"].data_ptr(), out["off"].data_ptr(), out["total"].data_ptr(), out["visited"].data_ptr(), stream))
        return out

    def insert_counters(self):
        """(batches inserted in place, batches that rebuilt the layout) -- see include/cis_hip.h:cis_index_insert_counters."""
        c = np.zeros(2, dtype=np.int64)
        _lib.check(_lib.lib().cis_index_insert_counters(self._ix, _lib.ptr(c)))
        return int(c[0]), int(c[1])

    def stream_counters(self):
        """(batches served by the HBM-streaming route, batches it handed back to the generic path) -- include/cis_hip.h."""
        c = np.zeros(2, dtype=np.int64)
        _lib.check(_lib.lib().cis_index_stream_counters(self._ix, _lib.ptr(c)))
        return int(c[0]), int(c[1])

    def last_stats(self):
        """Counters of the last search: candidates scanned, work items, tables, scan launches."""
        st = np.zeros(4, dtype=np.int64)
        _lib.check(_lib.lib().cis_index_last_stats(self._ix, _lib.ptr(st)))
        kind = int(_lib.lib().cis_index_last_scan_kernel(self._ix))
        return {"candidates": int(st[0]), "items": int(st[1]), "tables": int(st[2]), "scan_launches": int(st[3]),
                "scan_kernel": {0: None, 1: "k_adc_scan", 2: "k_adc_scan2", 3: "k_adc_scan3", 4: "k_adc_scan4", 5: "k_adc_stream", 6: "k_adc_scan5"}.get(kind)}


    default_prefilter_only = False  # tests: new searchers keep the float32-prefilter kernel for every batch size
    default_scan_mode = 0           # tests: 3 = new searchers run the 16-bit fixed-point kernel for every batch size

    def set_scan_mode(self, exact_only=False, prefilter_only=None, mode=None):
        """Routing of limit <= 952 (tests; results are identical on every route): exact_only forces the float64 scan
        kernel, prefilter_only the float32-prefilter kernel for every batch size (small batches otherwise take the
        all-candidates path: shorter kernel chain at low occupancy; large ones the 16-bit fixed-point kernel);
        mode = the raw cis_index_set_scan_mode value (3: the 16-bit fixed-point kernel for every batch size)."""
        if mode is None:
            if prefilter_only is None:
                prefilter_only = type(self).default_prefilter_only
            mode = 1 if exact_only else (2 if prefilter_only else type(self).default_scan_mode)
        _lib.check(_lib.lib().cis_index_set_scan_mode(self._ix, int(mode)))

    def set_profiling(self, enable=True, scan_only=False):
        """Record HIP events on the launch stream (see read_profile): around every stage, or (scan_only) just the pair
        around the scan kernel -- every recorded event is a small bubble between kernels."""
        _lib.check(_lib.lib().cis_index_set_profiling(self._ix, 0 if not enable else (1 if scan_only else 2)))

    def read_profile(self):
        """Accumulated stage times in ms since the last read (waits for the recorded events)."""
        ms = np.zeros(5, dtype=np.float64)
        n = _lib.c_int64(0)
        _lib.check(_lib.lib().cis_index_read_profile(self._ix, _lib.ptr(ms), _lib.ctypes.byref(n)))
        return {"front_ms": float(ms[0]), "tables_ms": float(ms[1]), "scan_ms": float(ms[2]),
                "merge_ms": float(ms[3]), "scan_kernel_ms": float(ms[4]), "scan_launches": int(n.value)}


def merge_hits_dev(parts, with_codes=False):
    """Merge per-shard hit lists parts [world, nq, L, 32] (uint8, on the GPU) into the final ranking
    by (dist, visit_rank, pos).  Returns dict of torch tensors like search_batch_dev."""
    import torch
    world, nq, L = int(parts.shape[0]), int(parts.shape[1]), int(parts.shape[2])
    if not (parts.is_cuda and parts.is_contiguous() and parts.dtype == torch.uint8 and parts.shape[3] == 32):
        raise ValueError("parts must be a contiguous uint8 [world, nq, L, 32] tensor on the GPU")
    dev = parts.device
    out = {"ids": torch.empty((nq, L), dtype=torch.int64, device=dev),
           "dists": torch.empty((nq, L), dtype=torch.float64, device=dev),
           "n_found": torch.zeros(nq, dtype=torch.int32, device=dev)}
    if with_codes:
        out["cells"] = torch.empty((nq, L), dtype=torch.int32, device=dev)
        out["pos"] = torch.empty((nq, L), dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().cis_merge_hits_dev(parts.data_ptr(), world, nq, L, out["ids"].data_ptr(),
                                             out["dists"].data_ptr(), out["n_found"].data_ptr(),
                                             out["cells"].data_ptr() if with_codes else None,
                                             out["pos"].data_ptr() if with_codes else None,
                                             torch.cuda.current_stream(dev).cuda_stream))
    return out


def merge_packed_dev(parts, off, cnt, nq, L, with_codes=False):
    """Merge packed per-shard hit lists: parts [world, stride, 4] int64 (cis_hit records; or flat [n, 4] with absolute offsets), off [world, nq] int64,
    cnt [world, nq] int32 -> dict like search_batch_dev.  This is what follows the all-gather over xGMI: only valid
    hits travel (about nq*L records in total instead of world*nq*L)."""
    import torch
    if parts.dim() == 2:  # one flat record buffer [n, 4], off holds absolute record offsets (the routed search's return trip)
        world, stride = int(off.shape[0]), 0
        if not (parts.is_cuda and parts.is_contiguous() and parts.dtype == torch.int64 and parts.shape[1] == 4):
            raise ValueError("parts must be a contiguous int64 [n, 4] tensor on the GPU")
    else:
        world, stride = int(parts.shape[0]), int(parts.shape[1])
        if not (parts.is_cuda and parts.is_contiguous() and parts.dtype == torch.int64 and parts.shape[2] == 4):
            raise ValueError("parts must be a contiguous int64 [world, stride, 4] tensor on the GPU")
    if not (off.is_contiguous() and off.dtype == torch.int64 and cnt.is_contiguous() and cnt.dtype == torch.int32
            and tuple(off.shape) == (world, nq) and tuple(cnt.shape) == (world, nq)):
        raise ValueError("off / cnt must be contiguous [world, nq] int64 / int32 tensors")
    dev = parts.device
    out = {"ids": torch.empty((nq, L), dtype=torch.int64, device=dev),
           "dists": torch.empty((nq, L), dtype=torch.float64, device=dev),
           "n_found": torch.zeros(nq, dtype=torch.int32, device=dev)}
    if with_codes:
        out["cells"] = torch.empty((nq, L), dtype=torch.int32, device=dev)
        out["pos"] = torch.empty((nq, L), dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().cis_merge_packed_dev(parts.data_ptr(), world, stride, off.data_ptr(), cnt.data_ptr(), nq, L,
                                               out["ids"].data_ptr(), out["dists"].data_ptr(), out["n_found"].data_ptr(),
                                               out["cells"].data_ptr() if with_codes else None,
                                               out["pos"].data_ptr() if with_codes else None,
                                               torch.cuda.current_stream(dev).cuda_stream))
    return out


class LOPQSearcherLMDB(LOPQSearcherBase):
    """LOPQSearcherLMDB (lopq/lopq/search.py:385-499) with the index on the GPU.

    What distinguishes the LMDB searcher from the dict one is kept exactly: an entry's key is the cell (2 x uint16,
    ``array('H')`` :425-427) followed by ``bytes(id)`` (:462, the py2 ``str`` of the id); ``put`` REPLACES an existing
    key (:465, last write wins -- the dict searcher keeps the first); ``get_cell`` walks the keys in byte order
    (:482-499), so items of a cell -- and therefore results with equal distances -- come in key order; ids come back
    through ``id_lambda``.  The key/value store lives on the host (a dict) and is PERSISTENT at ``lmdb_path``: in LMDB when
    the ``lmdb`` module is importable (the reference's own files; without the module an existing ``data.mdb`` is opened
    read-only through lopq/lmdb_read.py), otherwise in an append-only log of the same key / value bytes (lopq/kvlog.py) -- a restart re-opens the index (cold start tested in tests/test_reference_surfaces.py); the
    device index is rebuilt in key order before the first search after an insert."""

    def __init__(self, model, lmdb_path=None, id_lambda=int):
        super(LOPQSearcherLMDB, self).__init__()
        self.model = model
        self.lmdb_path = lmdb_path
        self.id_lambda = id_lambda
        # row store: one row per live key, in first-insertion order; a put of an existing key rewrites its row
        self._row_of = {}           # full key bytes (cell + suffix) -> row
        self._suffixes = []         # row -> key suffix bytes
        self._rows_of_cell = {}     # cell (c0, c1) -> [rows]
        self._cap = 0
        self._cells_arr = np.zeros((0, 2), dtype=np.uint16)      # row -> cell
        self._fine_arr = np.zeros((0, model.M), dtype=np.uint8)  # row -> fine codes
        self._dev = None   # LOPQSearcherHIP over the key-ordered arrays
        self._suffix_of_slot = []
        self._tail_of_cell = {}   # cell -> largest key suffix in the device index
        self._pending = []        # rows added since the device index was built (None: it must be rebuilt)
        self.env = None
        self._log = None   # kvlog.KVLog: the persistent form when the lmdb module is not installed
        if lmdb_path is not None:
            try:
                import lmdb
            except ImportError:
                lmdb = None
            has_mdb = os.path.exists(os.path.join(str(lmdb_path), "data.mdb"))       # the reference's own LMDB files
            has_log = os.path.exists(os.path.join(str(lmdb_path), kvlog.FILE_NAME))
            # never hide data: an existing LMDB index must be opened as LMDB, and two stores in one directory are ambiguous
            if has_mdb and has_log:
                raise RuntimeError("%s holds both an LMDB index (data.mdb) and a %s log: move one of them away -- refusing to pick"
                                   % (lmdb_path, kvlog.FILE_NAME))
            self._read_only = None
            if has_mdb and lmdb is None:
                # the reference's own files without py-lmdb: opened for SEARCHING through the page walker of lopq/lmdb_read.py
                # (unpinned: a restatement of LMDB's on-disk layout that stops at the first page that disagrees).  Inserts are
                # refused -- a fresh log here would shadow the stored items, and this module does not write LMDB.
                from . import lmdb_read
                for key, value in lmdb_read.Env(str(lmdb_path)).items(b"index"):
                    self._put(bytes(key), self.decode_fine_codes(value))
                self._read_only = ("%s holds an LMDB index (data.mdb) and the `lmdb` module is not installed: the index was opened "
                                   "read-only through lopq/lmdb_read.py; install py-lmdb to add items" % lmdb_path)
            elif lmdb is not None and not has_log:
                self.env = lmdb.open(self.lmdb_path, map_size=1024 * 1000000 * 32, max_dbs=1)  # :416
                self.index_db = self.env.open_db(b"index")
                with self.env.begin(db=self.index_db) as txn:
                    for key, value in txn.cursor():
                        self._put(bytes(key), self.decode_fine_codes(value))
            else:
                # cold start from the log: same keys, same values, later records replace earlier ones (put semantics); an
                # add_codes call whose write was torn by a crash is dropped as a whole, like an aborted LMDB transaction
                self._log = kvlog.KVLog(str(lmdb_path))
                for key, value in self._log.load():
                    self._put(bytes(key), self.decode_fine_codes(value))
        self.nb_indexed = len(self._suffixes)

    def _put(self, key, fine):
        """put(): a new key takes the next row, an existing key has its value replaced (last write wins, :465)."""
        row = self._row_of.get(key)
        if row is not None:
            self._pending = None  # a stored value changes: the device index is rebuilt before the next search
        elif self._pending is not None:
            self._pending.append(len(self._suffixes))
        if row is None:
            row = len(self._suffixes)
            if row >= self._cap:  # grow the row arrays geometrically
                self._cap = max(1024, 2 * self._cap)
                self._cells_arr = np.concatenate([self._cells_arr, np.zeros((self._cap - self._cells_arr.shape[0], 2), dtype=np.uint16)])
                self._fine_arr = np.concatenate([self._fine_arr, np.zeros((self._cap - self._fine_arr.shape[0], self._fine_arr.shape[1]), dtype=np.uint8)])
            self._row_of[key] = row
            self._suffixes.append(key[4:])
            cell = self.decode_cell(key[:4])
            self._cells_arr[row] = cell
            self._rows_of_cell.setdefault(cell, []).append(row)
        self._fine_arr[row] = fine

    def close(self):
        """Flush and release the persistent store (a log with many replaced keys is compacted first)."""
        if self._log is not None:
            live = self.get_nb_indexed()
            if self._log.records > 2 * max(live, 1):
                self._log.compact((key, self.encode_fine_codes(self._fine_arr[row])) for key, row in sorted(self._row_of.items()))
            self._log = None
        if self.env is not None:
            self.env.close()
            self.env = None
        if self._dev is not None:
            self._dev.close()
            self._dev = None

    @staticmethod
    def encode_cell(cell):
        import array
        return array.array("H", [int(cell[0]), int(cell[1])]).tobytes()

    @staticmethod
    def decode_cell(cell_bytes):
        import array
        a = array.array("H")
        a.frombytes(bytes(cell_bytes))
        return tuple(a.tolist())

    @staticmethod
    def encode_fine_codes(fine):
        return bytes(bytearray(int(v) for v in fine))

    @staticmethod
    def decode_fine_codes(fine_bytes):
        return tuple(bytearray(fine_bytes))

    @staticmethod
    def _key_suffix(item_id):
        return str(item_id).encode("latin1") if not isinstance(item_id, bytes) else item_id  # py2 bytes(id) == str(id)

    def _id_of_suffix(self, suffix):
        """id_lambda receives what the reference's py2 code hands it: the key suffix as a ``str`` (:489, ``key[4:]`` of
        a py2 byte string).  The production searcher passes ``id_lambda=str`` (searcher_lopqhbase.py:204-206), and
        ``str(b'sha1')`` under py3 would be the repr "b'sha1'" -- so the bytes are decoded first."""
        return self.id_lambda(suffix.decode("latin1"))

    def get_nb_indexed(self):
        self.nb_indexed = len(self._suffixes)
        return self.nb_indexed

    def add_codes(self, codes, ids=None):
        if getattr(self, "_read_only", None):
            raise ImportError(self._read_only)
        id_iter = count() if ids is None else iter(ids)
        # One call = one transaction, stored as a whole or not at all (the reference: `with env.begin(write=True)`, :459-467 -- an
        # exception inside it aborts the transaction).  The keys and values are made first (what can raise: a malformed code, an
        # id that does not encode), then written (LMDB transaction / one log transaction), and only then applied to the rows in
        # memory: a call that raises leaves memory and disk as they were.
        batch = []
        for item_id, code in zip(id_iter, codes):
            cell = (int(code[0][0]), int(code[0][1]))
            fine = tuple(int(v) for v in code[1])
            batch.append((self.encode_cell(cell) + self._key_suffix(item_id), fine))
        if self.env is not None:
            with self.env.begin(db=self.index_db, write=True) as txn:
                for key, fine in batch:
                    txn.put(key, self.encode_fine_codes(fine))
            self.env.sync()
        elif self._log is not None and batch:
            self._log.append((key, self.encode_fine_codes(fine)) for key, fine in batch)  # one transaction + fsync (the reference: env.sync(), :468)
        for key, fine in batch:
            self._put(key, fine)  # put(): an existing key is overwritten
        # the key-ordered GPU index follows before the next search: the new keys alone when they all sort behind their cells' last
        # keys (production ids arrive in time order inside an update), a rebuild otherwise (_device_index)
        self.get_nb_indexed()

    def add_codes_array(self, coarse, fine, ids=None, dedup=True):
        coarse = np.asarray(coarse).reshape(-1, 2)
        fine = np.asarray(fine).reshape(coarse.shape[0], -1)
        self.add_codes([((c[0], c[1]), tuple(f)) for c, f in zip(coarse, fine)], ids)

    def get_cell(self, cell):
        ct = _code_dtype(self.model.V)
        c = (int(cell[0]), int(cell[1]))
        rows = sorted(self._rows_of_cell.get(c, ()), key=lambda r: self._suffixes[r])  # the cursor's order: key bytes
        return [(self._id_of_suffix(self._suffixes[r]), LOPQCode((ct(c[0]), ct(c[1])), tuple(int(v) for v in self._fine_arr[r])))
                for r in rows]

    def _device_index(self):
        """The GPU index in key order: one lexicographic sort of (cell, key suffix) over the row arrays (numpy, no Python loop
        over the items) and one device-side bulk insert -- rounds 1-2 re-sorted a dict of dicts item by item on every refresh."""
        if self._dev is not None and self._pending:
            # incremental refresh: every new key must sort after the last key of its cell in the device index, then appending the new
            # keys in (cell, key) order keeps every cell in key order (the cursor's order, :482-499) -- the device insert is O(batch)
            rows = sorted(self._pending, key=lambda r: (tuple(int(v) for v in self._cells_arr[r]), self._suffixes[r]))
            ok, tails = True, dict()
            for r in rows:
                cell = tuple(int(v) for v in self._cells_arr[r])
                tail = tails.get(cell, self._tail_of_cell.get(cell))
                if tail is not None and self._suffixes[r] <= tail:
                    ok = False
                    break
                tails[cell] = self._suffixes[r]
            if ok:
                base = len(self._suffix_of_slot)
                rr = np.array(rows, dtype=np.int64)
                self._dev.add_codes_array(np.ascontiguousarray(self._cells_arr[rr]), np.ascontiguousarray(self._fine_arr[rr]),
                                          ids=np.arange(base, base + len(rows), dtype=np.int64), dedup=False)
                self._suffix_of_slot.extend(self._suffixes[r] for r in rows)
                self._tail_of_cell.update(tails)
                self._pending = []
            else:
                self._pending = None
        if self._dev is not None and self._pending is None:
            self._dev.close()
            self._dev = None
        if self._dev is None:
            n = len(self._suffixes)
            self._dev = LOPQSearcherHIP(self.model)
            self._pending = []
            self._tail_of_cell = {}
            if n:
                cells = self._cells_arr[:n]
                cell_id = cells[:, 0].astype(np.int64) * 65536 + cells[:, 1]
                # fixed-width byte strings compare like the keys do (a shorter key sorts first; ids hold no NUL bytes)
                order = np.lexsort((np.array(self._suffixes, dtype="S"), cell_id))
                self._suffix_of_slot = [self._suffixes[r] for r in order]
                for r in order:  # ascending (cell, key): the last write per cell is its tail
                    self._tail_of_cell[(int(cells[r, 0]), int(cells[r, 1]))] = self._suffixes[r]
                self._dev.add_codes_array(np.ascontiguousarray(cells[order]), np.ascontiguousarray(self._fine_arr[:n][order]),
                                          ids=np.arange(n, dtype=np.int64), dedup=False)
            else:
                self._suffix_of_slot = []
        return self._dev

    def search_batch(self, X, quota=10, limit=None, with_codes=False):
        """ids are slots of the key-ordered arrays: map with ``caller_ids``."""
        return self._device_index().search_batch(X, quota=quota, limit=limit, with_codes=with_codes)

    def caller_ids(self, dev_ids):
        return [self._id_of_suffix(self._suffix_of_slot[int(i)]) for i in dev_ids if i >= 0]

    def search(self, x, quota=10, limit=None, with_dists=False):
        dev = self._device_index()
        results, visited = dev.search(x, quota=quota, limit=limit, with_dists=with_dists)
        return [r._replace(id=self._id_of_suffix(self._suffix_of_slot[int(r.id)])) for r in results], visited


# the reference's dict searcher's name resolves to the HIP searcher so that config strings written for it keep working
LOPQSearcher = LOPQSearcherHIP

__all__ = ["LOPQSearcherBase", "LOPQSearcherHIP", "LOPQSearcher", "LOPQSearcherLMDB", "LOPQModel", "LOPQModelPCA", "LOPQCode",
           "multisequence", "multisequence_batch", "merge_hits_dev", "merge_packed_dev"]
