"""Helpers with the names of lopq/lopq/utils.py; the bulk encoders run on the GPU."""
import struct

import numpy as np


def iterate_splits(x, splits):
    """Equal contiguous sub-vectors of a 1-D array (reference: lopq/lopq/utils.py:8-22)."""
    size = len(x) // splits
    for s in range(splits):
        yield x[s * size:(s + 1) * size], s


def concat_new_first(arrs):
    """Stack along a new first axis (reference: lopq/lopq/utils.py:25-30)."""
    return np.stack([np.asarray(a) for a in arrs], axis=0)


def predict_cluster(x, centroids):
    """Nearest centroid of one vector (or of every row of a matrix), computed on the GPU with the reference's
    arithmetic (lopq/lopq/utils.py:33-53): returns np.uint8 / uint16 / uint32 like the reference."""
    from .. import _lib
    C = _lib.as_float_matrix(centroids)
    X = _lib.as_float_matrix(x, C.shape[1])
    out = np.empty(X.shape[0], dtype=np.uint32)
    _lib.check(_lib.lib().cis_predict_cluster(_lib.ptr(X), _lib.dtype_code(X), _lib.ptr(C), _lib.dtype_code(C), X.shape[0],
                                              C.shape[0], C.shape[1], _lib.ptr(out)))
    t = np.uint8 if C.shape[0] <= 256 else (np.uint16 if C.shape[0] <= 65536 else np.uint32)
    return t(out[0]) if np.ndim(x) == 1 else out.astype(t)


def compute_codes_notparallel(data, model):
    """[model.predict(d) for d in data] as ONE batched GPU encode
    (reference: lopq/lopq/utils.py:203-218).  Returns a list of LOPQCode."""
    from .model import LOPQCode
    coarse, fine = model.predict_batch(np.asarray(data))
    return [LOPQCode(tuple(c), tuple(f)) for c, f in zip(coarse, fine)]


def compute_codes_parallel(data, model, num_procs=4):
    """Same result as compute_codes_notparallel; the process pool of the reference
    (lopq/lopq/utils.py:178-200) is replaced by the GPU batch, so num_procs is ignored."""
    return iter(compute_codes_notparallel(data, model))


_XVECS = {"f": ("f", 4, np.float64), "i": ("I", 4, np.int64), "b": ("B", 1, np.float64)}


def load_xvecs(filename, base_type="f", max_num=None):
    """Read a .fvecs/.ivecs/.bvecs file (reference: lopq/lopq/utils.py:64-100)."""
    code, size, out_type = _XVECS[base_type]
    raw = np.fromfile(filename, dtype=np.uint8)
    d = int(struct.unpack("<I", raw[:4].tobytes())[0])
    rec = 4 + d * size
    n = raw.size // rec
    if max_num is not None:
        n = min(n, max_num)
    body = raw[:n * rec].reshape(n, rec)[:, 4:]
    vals = np.frombuffer(body.tobytes(), dtype=np.dtype("<" + {"f": "f4", "I": "u4", "B": "u1"}[code]))
    return np.squeeze(vals.reshape(n, d).astype(out_type))


def save_xvecs(data, filename, base_type="f"):
    """Write the format read by load_xvecs (reference: lopq/lopq/utils.py:103-131)."""
    code, _, _ = _XVECS[base_type]
    with open(filename, "wb") as f:
        for row in data:
            row = np.atleast_1d(row)
            f.write(struct.pack("<I", len(row)))
            f.write(struct.pack("<%d%s" % (len(row), code), *row.tolist()))
