"""Host-side LOPQ training (numpy + scikit-learn), vectorised.

Restates the training recipe of the reference (lopq/lopq/model.py:19-437) without its per-sample
Python loops: covariance accumulators become matrix products.  k-means results depend on the
scikit-learn version, so training is judged by distortion/recall, not bit parity (SURVEY.md
section 8c caveat ii, section 8f row 3).  Not on the encode/search hot path.
"""
import logging

import numpy as np

logger = logging.getLogger(__name__)


def eigenvalue_allocation(num_buckets, eigenvalues):
    """Greedy balancing of eigenvalue log-products over buckets (OPQ section 3.2.4).
    reference: lopq/lopq/model.py:19-71.  Returns the permutation of dimensions."""
    eigenvalues = np.asarray(eigenvalues, dtype=np.float64)
    D = eigenvalues.shape[0]
    per_bucket = D // num_buckets
    nz = np.abs(eigenvalues[np.nonzero(eigenvalues)])
    scaled = eigenvalues / (nz.min() if nz.size else 1.0)  # keeps every |eigenvalue| >= 1
    with np.errstate(divide="ignore"):
        logs = np.log2(np.abs(scaled))
    load = np.zeros(num_buckets)
    fill = np.zeros(num_buckets, dtype=np.int64)
    perm = np.zeros((num_buckets, per_bucket), dtype=np.int64)
    for dim in np.argsort(scaled)[::-1]:
        open_buckets = np.nonzero(fill < per_bucket)[0]
        b = open_buckets[np.argmin(load[open_buckets])]
        load[b] += logs[dim]
        perm[b, fill[b]] = dim
        fill[b] += 1
    return perm.reshape(D)


def assign_clusters(data, C, chunk=8192):
    """argmin_c ||x - c||^2 with the reference's direct (x - c)^2 form (lopq/lopq/utils.py:47)."""
    out = np.empty(data.shape[0], dtype=np.int64)
    for a in range(0, data.shape[0], chunk):
        d = data[a:a + chunk, None, :] - C[None, :, :]
        out[a:a + chunk] = np.einsum("nvd,nvd->nv", d, d).argmin(axis=1)
    return out


ACCUM_BACKEND = "host"  # "hip": covariance accumulations and residual projections on the GPU (csrc/lopq_train.hip)


def _group_rows(assign, groups):
    """Stable order of the rows by group + the group offsets (what cis_train_gram / cis_train_project take)."""
    order = np.argsort(assign, kind="stable")
    off = np.concatenate([[0], np.cumsum(np.bincount(assign, minlength=groups))]).astype(np.int64)
    return order, off


def gram_hip(X, assign=None, groups=1):
    """(G [groups,d,d], S [groups,d]): per-group sums of x x^T and of x, float64, on the GPU."""
    from .. import _lib
    X = np.ascontiguousarray(X, dtype=np.float64)
    n, d = X.shape
    if assign is None:
        off = np.array([0, n], dtype=np.int64)
    else:
        order, off = _group_rows(np.asarray(assign, dtype=np.int64), groups)
        X = np.ascontiguousarray(X[order])
    G = np.empty((groups, d, d))
    S = np.empty((groups, d))
    _lib.check(_lib.lib().cis_train_gram(_lib.ptr(X), n, d, _lib.ptr(off), groups, _lib.ptr(G), _lib.ptr(S)))
    return G, S


def local_rotations(data, C, num_buckets):
    """Per-cluster residual mean and PCA rotation with balanced variance.
    reference: lopq/lopq/model.py:74-206.  Returns (R [V,d,d], mu [V,d], assignments, residuals)."""
    V, d = C.shape
    assign = assign_clusters(data, C)
    residuals = np.asarray(data - C[assign], dtype=np.float64)
    R = np.zeros((V, d, d))
    mu = np.zeros((V, d))
    grams = None
    if ACCUM_BACKEND == "hip":  # all V accumulators in one launch: the reference's per-sample np.outer loop (:142-155)
        grams, sums = gram_hip(residuals, assign, V)
        counts = np.bincount(assign, minlength=V)
    for c in range(V):
        if grams is None:
            r = residuals[assign == c]
            n = r.shape[0]
            if n > 0:
                mu[c] = r.sum(axis=0) / n
        else:
            n = int(counts[c])
            if n > 0:
                mu[c] = sums[c] / n
        if n < d:
            logger.warning("Fewer points (%d) than dimensions (%d) in rotation computation for cluster %d", n, d, c)
            eigvals, vecs = np.ones(d), np.eye(d)
        else:
            A = r.T.dot(r) if grams is None else grams[c]
            cov = (A + A.T) / (2.0 * (n - 1)) - np.outer(mu[c], mu[c])
            eigvals, vecs = np.linalg.eigh(cov)
        R[c] = vecs[:, eigenvalue_allocation(num_buckets, eigvals)].T
    return R, mu, assign, residuals


def project_to_local(residuals, assign, R, mu):
    """R[a] . (res - mu[a]) for every training residual.  reference: lopq/lopq/model.py:209-234."""
    if ACCUM_BACKEND == "hip":
        from .. import _lib
        V, d = mu.shape
        order, off = _group_rows(np.asarray(assign, dtype=np.int64), V)
        X = np.ascontiguousarray(np.asarray(residuals, dtype=np.float64)[order])
        Y = np.empty_like(X)
        Rc, mc = np.ascontiguousarray(R, dtype=np.float64), np.ascontiguousarray(mu, dtype=np.float64)
        _lib.check(_lib.lib().cis_train_project(_lib.ptr(X), X.shape[0], d, _lib.ptr(off), V, _lib.ptr(Rc), _lib.ptr(mc), _lib.ptr(Y)))
        out = np.empty_like(Y)
        out[order] = Y
        return out
    out = np.zeros(residuals.shape)
    for c in np.unique(assign):
        sel = np.nonzero(assign == c)[0]
        out[sel] = (residuals[sel] - mu[c]).dot(R[c].T)
    return out


def kmeans_pp_init(data, k, rs, sample=20000):
    """k-means++ seeding (Arthur & Vassilvitskii) on a subsample, on the host: k sequential draws."""
    n = data.shape[0]
    x = np.asarray(data[rs.choice(n, min(n, sample), replace=False)], dtype=np.float64)
    C = np.empty((k, x.shape[1]))
    C[0] = x[rs.randint(len(x))]
    d2 = ((x - C[0]) ** 2).sum(axis=1)
    for j in range(1, k):
        tot = d2.sum()
        C[j] = x[rs.randint(len(x))] if tot <= 0 else x[np.searchsorted(np.cumsum(d2), rs.rand() * tot)]
        d2 = np.minimum(d2, ((x - C[j]) ** 2).sum(axis=1))
    return C


def kmeans_hip(data, k, iters, n_init=1, random_state=None):
    """Lloyd iterations on the GPU (cis_kmeans) from k-means++ seeds drawn on the host; best of n_init by inertia.
    Counterpart of the scikit-learn call of the reference (lopq/lopq/model.py:359-372,:417-432); float32."""
    from .. import _lib
    x = np.ascontiguousarray(data, dtype=np.float32)
    n, d = x.shape
    rs = np.random.RandomState(random_state)
    best, best_inertia = None, np.inf
    for _ in range(max(int(n_init), 1)):
        C = np.ascontiguousarray(kmeans_pp_init(x, k, rs), dtype=np.float32)
        inertia = _lib.ctypes.c_double(0.0)
        _lib.check(_lib.lib().cis_kmeans(_lib.ptr(x), n, d, k, int(iters), _lib.ptr(C), None, _lib.ctypes.byref(inertia)))
        if inertia.value < best_inertia:
            best, best_inertia = C, inertia.value
    return best.astype(np.float64), best_inertia


KMEANS_BACKEND = "sklearn"  # "hip": Lloyd iterations on the GPU where the codebook fits the kernel (k * d <= 7680)


def _kmeans(data, k, iters, n_init, random_state):
    if KMEANS_BACKEND == "hip" and k * data.shape[1] <= 7680:
        return kmeans_hip(data, k, iters, n_init, random_state)[0]
    from sklearn.cluster import MiniBatchKMeans
    km = MiniBatchKMeans(n_clusters=k, init="k-means++", max_iter=iters, n_init=n_init, batch_size=10000,
                         verbose=False, random_state=random_state)
    km.fit(data)
    return km.cluster_centers_


def train_pca(data, pca_dims=256, pca_subsample=None):
    """PCA basis with variance balanced over the two halves.  reference: lopq/lopq/model.py:242-287."""
    if pca_subsample:
        data = data[:min(pca_subsample, data.shape[0])]
    n, D = data.shape
    pca_dims = min(pca_dims, D)
    X = np.asarray(data, dtype=np.float64)
    if ACCUM_BACKEND == "hip":  # summed_covar of the reference's per-sample loop (:263-267) as one tiled product
        G, S = gram_hip(X)
        mu = S[0] / n
        A = G[0] / (n - 1) - np.outer(mu, mu)
    else:
        mu = X.mean(axis=0)
        A = X.T.dot(X) / (n - 1) - np.outer(mu, mu)
    E, P = np.linalg.eigh(A)
    E, P = E[-pca_dims:], P[:, -pca_dims:]
    P = P[:, eigenvalue_allocation(2, E)]
    return {"mu": mu, "P": P, "E": E, "A": A, "c": n}, pca_dims


def apply_pca_host(x, P, mu, renorm, dtype=np.float32):
    """Training-time apply_PCA on the host (reference: lopq/lopq/model.py:961-978)."""
    y = np.dot(x - mu, P)
    if renorm:
        y = y / np.linalg.norm(y, axis=-1, keepdims=True)
    return y.astype(dtype)


def train(data, V=8, M=4, subquantizer_clusters=256, parameters=None, kmeans_coarse_iters=10,
          kmeans_local_iters=20, n_init=10, subquantizer_sample_ratio=1.0, random_state=None, verbose=False):
    """Fit whatever is missing from `parameters`.  reference: lopq/lopq/model.py:339-437."""
    Cs = Rs = mus = subs = None
    if parameters is not None:
        Cs, Rs, mus, subs = parameters
    if Rs is None or mus is None:
        Rs = mus = None
    halves = np.split(data, 2, axis=1)
    if Cs is None:
        Cs = tuple(_kmeans(h, V, kmeans_coarse_iters, n_init, random_state) for h in halves)
    fitted = None
    if Rs is None:
        fitted = [local_rotations(h, C, M // 2) for h, C in zip(halves, Cs)]
        Rs = tuple(f[0] for f in fitted)
        mus = tuple(f[1] for f in fitted)
    if subs is not None:
        return tuple(Cs), tuple(Rs), tuple(mus), subs
    N = data.shape[0]
    n_sub = int(np.floor(min(subquantizer_sample_ratio, 1.0) * N))
    sample = np.random.RandomState(random_state).choice(N, n_sub, False)
    subs = []
    for s, (h, C) in enumerate(zip(halves, Cs)):
        if fitted is not None:
            assign, residuals = fitted[s][2][sample], fitted[s][3][sample]
        else:
            hs = h[sample]
            assign = assign_clusters(hs, C)
            residuals = np.asarray(hs - C[assign], dtype=np.float64)
        projected = project_to_local(residuals, assign, Rs[s], mus[s])
        subs.append([_kmeans(p, subquantizer_clusters, kmeans_local_iters, n_init, random_state)
                     for p in np.split(projected, M // 2, axis=1)])
    return tuple(Cs), tuple(Rs), tuple(mus), tuple(subs)
