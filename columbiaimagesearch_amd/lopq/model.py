"""LOPQModel / LOPQModelPCA with the reference's surface, computing on the MI355X.

Mirror of lopq/lopq/model.py:447-1184 of the reference: same class names, constructor arguments,
attribute set (``Cs, Rs, mus, subquantizers, V, M, num_coarse_splits, num_fine_splits,
subquantizer_clusters`` and ``pca_P, pca_mu, renorm``), method names and return types -- so that the
cufacesearch searcher (cufacesearch/searcher/searcher_lopqhbase.py:397-524) and pickled models keep
working -- but every method that does arithmetic on vectors calls libcis_hip.so.  Batched
variants (``predict_batch`` ...) are additions; the per-vector methods are the batch of one.

Training (``fit`` / ``fit_pca``) is host-side numpy + scikit-learn for now (SURVEY.md section 8f
row 3 moves it to the GPU); it is not on the encode/search hot path.
"""
from collections import namedtuple

import numpy as np

from .. import _lib
from . import train as _train

LOPQCode = namedtuple("LOPQCode", ["coarse", "fine"])  # reference: lopq/lopq/model.py:444


def _code_dtype(n):
    # reference: predict_cluster picks the smallest unsigned type, lopq/lopq/utils.py:48-53
    return np.uint8 if n <= 256 else (np.uint16 if n <= 65536 else np.uint32)


class _Handle(object):
    """Owns a cis_model*; rebuilt when the parameter arrays are replaced.  Keeps the arrays it was built from
    alive, so that "same objects" (the cache test) cannot be fooled by a recycled id()."""

    def __init__(self, ptr, arrays, renorm):
        self.ptr, self.arrays, self.renorm = ptr, arrays, renorm

    def matches(self, arrays, renorm):
        return (self.renorm == renorm and len(self.arrays) == len(arrays)
                and all(a is b for a, b in zip(self.arrays, arrays)))

    def __del__(self):
        try:
            if self.ptr:
                _lib.lib().cis_model_destroy(self.ptr)
        except Exception:
            pass
        self.ptr = None


class LOPQModel(object):
    """reference: lopq/lopq/model.py:447-820"""

    def __init__(self, V=8, M=4, subquantizer_clusters=256, parameters=None):
        if parameters is None:
            parameters = (None, None, None, None)
        self.Cs, self.Rs, self.mus, self.subquantizers = parameters
        self._derive_hyperparameters(V, M, subquantizer_clusters)

    # -- hyper-parameters follow the parameters when those are given (model.py:479-493) ---------
    def _derive_hyperparameters(self, V, M, subquantizer_clusters):
        self.num_coarse_splits = 2 if self.Cs is None else len(self.Cs)
        self.V = V if self.Cs is None else self.Cs[0].shape[0]
        if self.subquantizers is None:
            self.num_fine_splits = M // 2
            self.M = M
            self.subquantizer_clusters = subquantizer_clusters
        else:
            self.num_fine_splits = len(self.subquantizers[0])
            self.M = self.num_fine_splits * self.num_coarse_splits
            self.subquantizer_clusters = self.subquantizers[0][0].shape[0]

    # -- pickling keeps exactly the reference's attribute set -----------------------------------
    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_hip", None)
        state.pop("_dims", None)
        return state

    # -- device handle ---------------------------------------------------------------------------
    def _pca_params(self):
        return None, None, False

    def _handle(self):
        return self._handle_obj().ptr

    def _handle_obj(self):
        """The _Handle owning the device copy of the current parameters.  A searcher keeps this object (not just
        the pointer): the index borrows the cis_model*, which must outlive it even if the model is re-fitted."""
        if self.Cs is None or self.Rs is None or self.mus is None or self.subquantizers is None:
            raise ValueError("model has no parameters yet: call fit() or pass parameters=")
        P, pmu, renorm = self._pca_params()
        arrays = [self.Cs[0], self.Cs[1], self.Rs[0], self.Rs[1], self.mus[0], self.mus[1]]
        arrays += list(self.subquantizers[0]) + list(self.subquantizers[1]) + [P, pmu]
        h = self.__dict__.get("_hip")
        if h is not None and h.matches(arrays, bool(renorm)):
            return h
        C0, C1 = np.asarray(self.Cs[0]), np.asarray(self.Cs[1])
        coarse_f32 = (C0.dtype == np.float32 and C1.dtype == np.float32)
        cdt = np.float32 if coarse_f32 else np.float64
        Cs = np.ascontiguousarray(np.stack([C0.astype(cdt), C1.astype(cdt)]))
        Rs = np.ascontiguousarray(np.stack([self.Rs[0], self.Rs[1]]), dtype=np.float64)
        mus = np.ascontiguousarray(np.stack([self.mus[0], self.mus[1]]), dtype=np.float64)
        subs = np.ascontiguousarray(
            np.stack([np.asarray(s) for half in self.subquantizers for s in half]), dtype=np.float64)
        V, hdim = C0.shape
        D = 2 * hdim
        M, K, w = subs.shape
        if Rs.shape != (2, V, hdim, hdim) or mus.shape != (2, V, hdim) or w * M != D:
            raise ValueError("inconsistent LOPQ parameter shapes")
        mu_f32 = False
        if P is not None:
            mu_f32 = np.asarray(pmu).dtype == np.float32  # x - pca_mu rounds in float32 for float32 x
            P = np.ascontiguousarray(P, dtype=np.float64)
            pmu = np.ascontiguousarray(pmu, dtype=np.float64)
            if P.shape[1] != D or pmu.shape != (P.shape[0],):
                raise ValueError("PCA parameters do not match the LOPQ dimension")
            D_in = P.shape[0]
        else:
            D_in = D
        out = _lib.c_void_p()
        L = _lib.lib()
        _lib.check(L.cis_model_create(_lib.ctypes.byref(out), D_in, D, V, M, K,
                                      _lib.CIS_F32 if coarse_f32 else _lib.CIS_F64, _lib.ptr(Cs), _lib.ptr(Rs),
                                      _lib.ptr(mus), _lib.ptr(subs), _lib.ptr(P), _lib.ptr(pmu),
                                      _lib.CIS_F32 if mu_f32 else _lib.CIS_F64, 1 if renorm else 0))
        self.__dict__["_hip"] = _Handle(out.value, arrays, bool(renorm))
        self.__dict__["_dims"] = (D_in, D)
        return self.__dict__["_hip"]

    @property
    def dim(self):
        """Dimension of the vectors LOPQ itself works on (after PCA if any)."""
        return 2 * self.Cs[0].shape[1]

    @property
    def input_dim(self):
        self._handle()
        return self.__dict__["_dims"][0]

    # -- training --------------------------------------------------------------------------------
    def fit(self, data, kmeans_coarse_iters=10, kmeans_local_iters=20, n_init=10, subquantizer_sample_ratio=1.0,
            random_state=None, verbose=False):
        """reference: lopq/lopq/model.py:495-520 (train :339-437); trains only what is missing."""
        params = _train.train(data, self.V, self.M, self.subquantizer_clusters,
                              (self.Cs, self.Rs, self.mus, self.subquantizers), kmeans_coarse_iters,
                              kmeans_local_iters, n_init, subquantizer_sample_ratio, random_state, verbose)
        self.Cs, self.Rs, self.mus, self.subquantizers = params

    def get_split_parameters(self, split):
        """reference: lopq/lopq/model.py:522-541"""
        pick = lambda p: None if p is None else p[split]
        return pick(self.Cs), pick(self.Rs), pick(self.mus), pick(self.subquantizers)

    # -- encode ----------------------------------------------------------------------------------
    def _lopq_space(self, x):
        """Vectors as LOPQ sees them: this class has no PCA, so the input itself."""
        return _lib.as_float_matrix(x, self.dim)

    def predict_batch(self, X):
        """Codes of many vectors at once: (coarse [n,2], fine [n,M]) unsigned arrays.
        Batched counterpart of compute_codes_notparallel (lopq/lopq/utils.py:203-218)."""
        h = self._handle()
        X = _lib.as_float_matrix(X, self.input_dim)
        n = X.shape[0]
        coarse = np.empty((n, 2), dtype=np.uint16)
        fine = np.empty((n, self.M), dtype=np.uint8)
        _lib.check(_lib.lib().cis_encode(h, _lib.ptr(X), _lib.dtype_code(X), n, _lib.ptr(coarse), _lib.ptr(fine)))
        return coarse.astype(_code_dtype(self.V), copy=False), fine

    def predict_batch_dev(self, X, out=None):
        """Encode a [n, D_in] float32/float64 tensor already in HBM -> (coarse int16-as-uint16 bits
        [n,2], fine uint8 [n,M]) torch tensors on the same device; asynchronous."""
        import torch
        h = self._handle()
        if not (X.is_cuda and X.is_contiguous() and X.dim() == 2 and X.shape[1] == self.input_dim):
            raise ValueError("X must be a contiguous [n, %d] tensor on the GPU" % self.input_dim)
        code = _lib.CIS_F32 if X.dtype == torch.float32 else _lib.CIS_F64
        n = X.shape[0]
        if out is None:
            out = (torch.empty((n, 2), dtype=torch.int16, device=X.device),
                   torch.empty((n, self.M), dtype=torch.uint8, device=X.device))
        _lib.check(_lib.lib().cis_encode_dev(h, X.data_ptr(), code, n, out[0].data_ptr(), out[1].data_ptr(),
                                             torch.cuda.current_stream(X.device).cuda_stream))
        return out

    def twin(self):
        """A second model object on the SAME parameter arrays with its own device handle (own encode workspaces; the parameters are
        uploaded once more: KBs to a few MB).  Encode passes of consecutive chunks through a model and its twins, each under its own
        torch stream, overlap: 254 -> 283-290 M vectors/s at the C4 shape (profiles/r05_experiments.txt).  An index keeps using the
        model it was created with."""
        import copy
        m = copy.copy(self)
        m.__dict__.pop("_hip", None)
        m.__dict__.pop("_dims", None)
        return m

    def predict(self, x):
        """reference: lopq/lopq/model.py:543-561 -> LOPQCode(coarse tuple, fine tuple)"""
        coarse, fine = self.predict_batch(np.asarray(x)[None, :])
        return LOPQCode(tuple(coarse[0]), tuple(fine[0]))

    def predict_coarse(self, x):
        """reference: lopq/lopq/model.py:563-573 (x already in LOPQ space)"""
        h = self._handle()
        X = self._lopq_space(x)
        coarse = np.empty((X.shape[0], 2), dtype=np.uint16)
        _lib.check(_lib.lib().cis_predict_coarse(h, _lib.ptr(X), _lib.dtype_code(X), X.shape[0], _lib.ptr(coarse)))
        coarse = coarse.astype(_code_dtype(self.V), copy=False)
        return tuple(coarse[0]) if np.ndim(x) == 1 else coarse

    def _coarse_arg(self, coarse_codes, n):
        c = np.ascontiguousarray(np.asarray(coarse_codes, dtype=np.uint16).reshape(n, 2))
        return c

    def predict_fine(self, x, coarse_codes=None):
        """reference: lopq/lopq/model.py:575-602"""
        h = self._handle()
        X = self._lopq_space(x)
        n = X.shape[0]
        if coarse_codes is None:
            coarse_codes = self.predict_coarse(X)
        c = self._coarse_arg(coarse_codes, n)
        fine = np.empty((n, self.M), dtype=np.uint8)
        _lib.check(_lib.lib().cis_predict_fine(h, _lib.ptr(X), _lib.dtype_code(X), n, _lib.ptr(c), _lib.ptr(fine)))
        return tuple(fine[0]) if np.ndim(x) == 1 else fine

    def project(self, x, coarse_codes, coarse_split=None):
        """reference: lopq/lopq/model.py:604-641 -> float64 locally projected residual"""
        h = self._handle()
        X = self._lopq_space(x)
        n = X.shape[0]
        c = self._coarse_arg(coarse_codes, n)
        out = np.empty((n, self.dim), dtype=np.float64)
        _lib.check(_lib.lib().cis_project(h, _lib.ptr(X), _lib.dtype_code(X), n, _lib.ptr(c), _lib.ptr(out)))
        if coarse_split is not None:
            hd = self.dim // 2
            out = out[:, coarse_split * hd:(coarse_split + 1) * hd]
        return out[0] if np.ndim(x) == 1 else out

    def reconstruct(self, codes):
        """reference: lopq/lopq/model.py:643-671"""
        h = self._handle()
        coarse = np.ascontiguousarray(np.asarray(codes[0], dtype=np.uint16).reshape(1, 2))
        fine = np.ascontiguousarray(np.asarray(codes[1], dtype=np.uint8).reshape(1, self.M))
        out = np.empty((1, self.dim), dtype=np.float64)
        _lib.check(_lib.lib().cis_reconstruct(h, _lib.ptr(coarse), _lib.ptr(fine), 1, _lib.ptr(out)))
        return out[0]

    def get_subquantizer_distances(self, x, coarse_codes, coarse_split=None):
        """reference: lopq/lopq/model.py:673-704 -> list of (K,) float64 arrays"""
        h = self._handle()
        X = self._lopq_space(x)
        if X.shape[0] != 1:
            raise ValueError("get_subquantizer_distances takes one vector")
        c = self._coarse_arg(coarse_codes, 1)
        K = self.subquantizer_clusters
        tabs = np.empty((1, self.M, K), dtype=np.float64)
        _lib.check(_lib.lib().cis_subquantizer_distances(h, _lib.ptr(X), _lib.dtype_code(X), 1, _lib.ptr(c),
                                                         _lib.ptr(tabs)))
        nf = self.num_fine_splits
        js = range(self.M) if coarse_split is None else range(coarse_split * nf, (coarse_split + 1) * nf)
        return [tabs[0, j].copy() for j in js]

    # -- cell ids --------------------------------------------------------------------------------
    def get_cell_id_for_coarse_codes(self, coarse_codes):
        """reference: lopq/lopq/model.py:706-707 (widened so that V > 16 does not overflow uint8)"""
        return int(coarse_codes[1]) + int(coarse_codes[0]) * self.V

    def get_coarse_codes_for_cell_id(self, cell_id):
        """reference: lopq/lopq/model.py:709-710"""
        return (int(cell_id) // self.V, int(cell_id) % self.V)

    # -- exchange formats ------------------------------------------------------------------------
    def export_mat(self, filename):
        """reference: lopq/lopq/model.py:712-728"""
        from scipy.io import savemat
        stack = lambda arrs: np.stack([np.asarray(a) for a in arrs])
        savemat(filename, {"Cs": stack(self.Cs), "Rs": stack(self.Rs), "mus": stack(self.mus),
                           "subs": stack([stack(half) for half in self.subquantizers]), "V": self.V, "M": self.M})

    @staticmethod
    def load_mat(filename):
        """reference: lopq/lopq/model.py:730-746"""
        from scipy.io import loadmat
        d = loadmat(filename)
        two = lambda a: (np.ascontiguousarray(a[0]), np.ascontiguousarray(a[1]))
        subs = tuple([np.ascontiguousarray(s) for s in half] for half in d["subs"])
        return LOPQModel(parameters=(two(d["Cs"]), two(d["Rs"]), two(d["mus"]), subs))

    def export_proto(self, f):
        """reference: lopq/lopq/model.py:748-786 (LOPQModelParams, float32 packed)"""
        from .proto import write_model_params
        write_model_params(self, f)

    @staticmethod
    def load_proto(filename):
        """reference: lopq/lopq/model.py:788-820"""
        from .proto import read_model_params
        return read_model_params(filename)


class LOPQModelPCA(LOPQModel):
    """reference: lopq/lopq/model.py:823-1184"""

    def __init__(self, V=8, M=4, subquantizer_clusters=256, renorm=False, parameters=None):
        if parameters is None:
            parameters = (None,) * 6
        self.Cs, self.Rs, self.mus, self.subquantizers, self.pca_P, self.pca_mu = parameters
        self.renorm = renorm
        self._derive_hyperparameters(V, M, subquantizer_clusters)

    def _pca_params(self):
        if self.pca_P is None or self.pca_mu is None:
            raise ValueError("model has no PCA parameters yet: call fit_pca() or fit()")
        return self.pca_P, self.pca_mu, self.renorm

    def fit_pca(self, data, pca_dims=256, pca_subsample=None):
        """reference: lopq/lopq/model.py:878-886"""
        if self.pca_P is not None and self.pca_mu is not None:
            raise ValueError("You are trying to retrain PCA...")
        params, _ = _train.train_pca(data, pca_dims, pca_subsample)
        self.pca_P, self.pca_mu = params["P"], params["mu"]

    def fit(self, data, pca_dims=256, kmeans_coarse_iters=10, kmeans_local_iters=20, n_init=10,
            subquantizer_sample_ratio=1.0, random_state=None, verbose=False, pca_subsample=None, apply_pca=True,
            train_pca=True):
        """reference: lopq/lopq/model.py:889-937"""
        if train_pca:
            self.fit_pca(data, pca_dims, pca_subsample)
        if apply_pca and _train.ACCUM_BACKEND == "hip":  # the projection of the training set on the GPU as well (apply_PCA, :961-978)
            pca_data = self.apply_PCA(data)
        else:
            pca_data = _train.apply_pca_host(data, self.pca_P, self.pca_mu, self.renorm) if apply_pca else data
        params = _train.train(pca_data, self.V, self.M, self.subquantizer_clusters,
                              (self.Cs, self.Rs, self.mus, self.subquantizers), kmeans_coarse_iters,
                              kmeans_local_iters, n_init, subquantizer_sample_ratio, random_state, verbose)
        self.Cs, self.Rs, self.mus, self.subquantizers = params

    def _pca_only_handle(self):
        """apply_PCA before the LOPQ parameters exist (the reference's training flow: fit_pca, then apply_PCA on every batch of
        features, then fit -- lopq/lopq/model.py:878-937, cufacesearch/searcher/searcher_lopqhbase.py:340): a device model that carries
        the PCA parameters and placeholder LOPQ parameters of the right shapes (never used by apply_PCA)."""
        P = np.asarray(self.pca_P)
        key = (id(self.pca_P), id(self.pca_mu), bool(self.renorm))
        cached = self.__dict__.get("_pca_only")
        if cached is not None and cached[0] == key:
            return cached[1]
        D = P.shape[1]
        hd = D // 2
        eye = np.eye(hd)[None]
        tmp = LOPQModelPCA(V=1, M=2, renorm=self.renorm,
                           parameters=((np.zeros((1, hd)), np.zeros((1, hd))), (eye, eye), (np.zeros((1, hd)), np.zeros((1, hd))),
                                       ([np.zeros((self.subquantizer_clusters, hd))], [np.zeros((self.subquantizer_clusters, hd))]),
                                       self.pca_P, self.pca_mu))
        self.__dict__["_pca_only"] = (key, tmp)
        return tmp

    def apply_PCA(self, x, dtype=np.float32):
        """reference: lopq/lopq/model.py:961-978 -> float32, 1-D in -> 1-D out"""
        if (self.Cs is None or self.Rs is None or self.subquantizers is None) and self.pca_P is not None:
            return self._pca_only_handle().apply_PCA(x, dtype)
        h = self._handle()
        X = _lib.as_float_matrix(x, self.input_dim)
        out = np.empty((X.shape[0], self.dim), dtype=np.float32)
        _lib.check(_lib.lib().cis_apply_pca(h, _lib.ptr(X), _lib.dtype_code(X), X.shape[0], _lib.ptr(out)))
        out = out.astype(dtype, copy=False)
        return out[0] if np.ndim(x) == 1 else out

    def export_mat(self, filename):
        raise NotImplementedError("export_mat not yet supported for LOPQModelPCA")  # model.py:1156-1164

    @staticmethod
    def load_mat(filename):
        raise NotImplementedError("load_mat not yet supported for LOPQModelPCA")  # model.py:1166-1171

    def export_proto(self, f):
        raise NotImplementedError("export_proto not yet supported for LOPQModelPCA")  # model.py:1173-1177

    @staticmethod
    def load_proto(filename):
        raise NotImplementedError("load_proto not yet supported for LOPQModelPCA")  # model.py:1179-1184
