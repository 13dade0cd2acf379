"""ctypes binding of libcis_hip.so (the C ABI declared in include/cis_hip.h).

There is no CPU fallback: if the library is missing, or no MI355X is visible when a compute entry
point is called, the failure is raised to the caller.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint8, c_uint16, c_uint32, c_void_p

import numpy as np

CIS_F32, CIS_F64 = 4, 8
CIS_OK, CIS_EINVAL, CIS_EHIP, CIS_ENOMEM, CIS_EUNSUPPORTED, CIS_ENODEVICE = 0, -1, -2, -3, -4, -5

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CIS_LIB_PATH") or os.path.join(_HERE, "lib", "libcis_hip.so")  # override: kernel A/B experiments

# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue do not overlap.  This
# package runs batches in flight on several streams (index views, CNN views, copy streams, RCCL): ask for 16 queues unless the caller
# decided otherwise (8 were one too few from four search batches in flight on: the two part streams of a dlib forward landed on one
# queue -- one batch at a time 1.72 -> 2.47 ms, three views in flight 0.61 -> 0.55 of the MFMA peak; 12 and 16 measure alike).  Read by the HIP runtime when it initialises, i.e. effective when this module is imported before the first
# torch.cuda / HIP call of the process (bench.py sets it first thing; a caller that initialises HIP earlier sets it itself).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


class cis_hit(ctypes.Structure):
    _fields_ = [("dist", c_double), ("visit_rank", c_uint32), ("pos", c_uint32), ("id", c_int64),
                ("cell", c_int32), ("reserved", c_int32)]


HIT_DTYPE = np.dtype([("dist", "<f8"), ("visit_rank", "<u4"), ("pos", "<u4"), ("id", "<i8"),
                      ("cell", "<i4"), ("reserved", "<i4")])
assert HIT_DTYPE.itemsize == ctypes.sizeof(cis_hit) == 32


class HipError(RuntimeError):
    """A HIP runtime call inside libcis_hip.so failed (or no gfx950 device is visible)."""


_PROTOS = {
    # name: (restype, argtypes)
    "cis_version": (c_int, []),
    "cis_last_error": (c_char_p, []),
    "cis_alloc_stats": (c_int, [POINTER(c_int64), POINTER(c_int64)]),
    "cis_device_count": (c_int, []),
    "cis_set_device": (c_int, [c_int]),
    "cis_selftest": (c_int, [POINTER(c_int)]),
    "cis_model_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "cis_model_destroy": (None, [c_void_p]),
    "cis_apply_pca": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "cis_encode": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "cis_encode_dev": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p]),
    "cis_predict_coarse": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "cis_project": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "cis_predict_fine": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "cis_subquantizer_distances": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "cis_reconstruct": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "cis_predict_cluster": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_void_p]),
    "cis_multisequence": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p, c_void_p,
                                  POINTER(c_int)]),
    "cis_index_create": (c_int, [POINTER(c_void_p), c_void_p]),
    "cis_index_destroy": (None, [c_void_p]),
    "cis_index_create_view": (c_int, [POINTER(c_void_p), c_void_p]),
    "cis_index_insert_counters": (c_int, [c_void_p, c_void_p]),
    "cis_index_set_shard": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "cis_index_cell_counts": (c_int, [c_void_p, c_void_p]),
    "cis_index_add_remote_counts": (c_int, [c_void_p, c_void_p]),
    "cis_index_add": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, POINTER(c_int64)]),
    "cis_index_size": (c_int64, [c_void_p]),
    "cis_index_add_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, POINTER(c_int64), POINTER(c_int64),
                                  c_void_p, c_void_p]),
    "cis_l2_normalize_dev": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p]),
    "cis_index_cell_counts_dev": (c_int, [c_void_p, c_void_p, c_void_p]),
    "cis_index_add_remote_counts_dev": (c_int, [c_void_p, c_void_p, c_void_p]),
    "cis_index_route_pack_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "cis_index_add_records_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int, POINTER(c_int64), POINTER(c_int64), c_void_p,
                                          c_void_p]),
    "cis_index_get_cell": (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, POINTER(c_int64)]),
    "cis_index_get_codes": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "cis_index_search": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p]),
    "cis_index_search_async": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    "cis_index_search_wait": (c_int, [c_void_p]),
    "cis_host_alloc": (c_int, [POINTER(c_void_p), ctypes.c_size_t]),
    "cis_host_free": (None, [c_void_p]),
    "cis_index_search_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    "cis_index_search_partial_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_void_p,
                                             c_void_p]),
    "cis_index_query_owners_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p]),
    "cis_routed_merge_tables_dev": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "cis_route_queries_dev": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cis_index_search_partial_packed_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_void_p,
                                                    c_void_p, c_void_p, c_void_p, c_void_p]),
    "cis_merge_hits_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p]),
    "cis_merge_packed_dev": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    "cis_pyramid_down2_dev": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "cis_extract_chips_dev": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "cis_exchange_offsets_dev": (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cis_rerank_dev": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "cis_kmeans": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "cis_train_gram": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "cis_train_project": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "cis_index_last_stats": (c_int, [c_void_p, c_void_p]),
    "cis_index_last_scan_kernel": (c_int, [c_void_p]),
    "cis_cnn_create": (c_int, [POINTER(c_void_p), c_int, c_void_p, c_int]),
    "cis_cnn_destroy": (None, [c_void_p]),
    "cis_cnn_create_view": (c_int, [POINTER(c_void_p), c_void_p]),
    "cis_cnn_feat_dim": (c_int, [c_int]),
    "cis_cnn_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "cis_cnn_forward_dev": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "cis_index_set_profiling": (c_int, [c_void_p, c_int]),
    "cis_index_set_scan_mode": (c_int, [c_void_p, c_int]),
    "cis_index_stream_counters": (c_int, [c_void_p, c_void_p]),
    "cis_index_read_profile": (c_int, [c_void_p, c_void_p, POINTER(c_int64)]),
}

_lib = None


def exported_symbols():
    """Names declared in include/cis_hip.h (kept in sync by tests/test_abi.py)."""
    return sorted(_PROTOS)


def _load_hip_runtime():
    """Make ONE HIP runtime global in this process before libcis_hip.so binds to it.

    libcis_hip.so is linked without a libamdhip64 dependency.  PyTorch-ROCm ships its own copy of the
    runtime; two runtimes in one process cannot share device pointers or streams, so when torch is
    installed its copy is the one we bind to (torch tensors are this package's device-memory and
    stream plumbing).  Without torch the system ROCm runtime is used.
    """
    cands = []
    try:
        import torch  # noqa: F401  (loads torch/lib/libamdhip64.so)
        cands.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except ImportError:
        pass
    cands += ["libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"]
    for c in cands:
        try:
            return ctypes.CDLL(c, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            continue
    raise ImportError("no HIP runtime (libamdhip64.so) found: install ROCm or PyTorch-ROCm")


def lib():
    """Load libcis_hip.so once.  Raises ImportError with build instructions if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C columbiaimagesearch_amd/csrc`). columbiaimagesearch_amd has no CPU fallback." % LIB_PATH)
        _load_hip_runtime()
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error():
    return lib().cis_last_error().decode("utf-8", "replace")


def check(rc):
    """Translate a C return code into the exception kinds the reference raises."""
    if rc == CIS_OK:
        return
    msg = last_error()
    if rc == CIS_EINVAL:
        raise ValueError(msg)
    if rc == CIS_EUNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == CIS_ENOMEM:
        raise MemoryError(msg)
    raise HipError(msg)


def ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_void_p)


def dtype_code(a):
    if a.dtype == np.float32:
        return CIS_F32
    if a.dtype == np.float64:
        return CIS_F64
    raise ValueError("only float32 / float64 vectors are supported, got %s" % a.dtype)


def as_float_matrix(x, cols=None):
    """C-contiguous float32/float64 2-D view of x (other dtypes are promoted to float64, as numpy
    would when the reference subtracts float64 parameters from them)."""
    a = np.asarray(x)
    if a.dtype != np.float32 and a.dtype != np.float64:
        a = a.astype(np.float64)
    if a.ndim == 1:
        a = a[None, :]
    if a.ndim != 2:
        raise ValueError("expected a vector or a matrix of vectors, got shape %r" % (a.shape,))
    if cols is not None and a.shape[1] != cols:
        raise ValueError("expected vectors of dimension %d, got %d" % (cols, a.shape[1]))
    return np.ascontiguousarray(a)


def alloc_stats():
    """(workspace allocations, bytes) since the process started -- include/cis_hip.h:cis_alloc_stats."""
    n, b = c_int64(0), c_int64(0)
    check(lib().cis_alloc_stats(ctypes.byref(n), ctypes.byref(b)))
    return int(n.value), int(b.value)


class _PinnedBlock(object):
    """Owner of one cis_host_alloc block (freed when the last array view on it goes away)."""

    def __init__(self, nbytes):
        p = c_void_p()
        check(lib().cis_host_alloc(ctypes.byref(p), int(nbytes)))
        self.ptr, self.nbytes = p.value, int(nbytes)

    def __del__(self):
        try:
            if self.ptr:
                lib().cis_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype):
    """numpy array in page-locked host memory (include/cis_hip.h:cis_host_alloc): what search_batch_async copies from / into by DMA."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
    blk = _PinnedBlock(max(n * dt.itemsize, 1))
    buf = (ctypes.c_char * blk.nbytes).from_address(blk.ptr)
    a = np.frombuffer(buf, dtype=dt, count=n).reshape(shape)
    _PINNED_OWNERS[id(buf)] = blk  # the ctypes buffer does not own the block: keep it alive as long as the buffer object lives
    import weakref
    weakref.finalize(buf, _PINNED_OWNERS.pop, id(buf), None)
    return a


_PINNED_OWNERS = {}
