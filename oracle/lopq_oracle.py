"""CPU restatement of the reference LOPQ encode/search path.  TEST INFRASTRUCTURE ONLY.

This module is the *oracle*: a numpy restatement of the vendored ``lopq`` package of
ColumbiaImageSearch.  It is imported only by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- never by the product package ``columbiaimagesearch_amd``.

Pinning: the restatement is checked bit-for-bit (codes, cell order, candidate ids) and to 1e-9
(distances) against golden vectors produced by importing the real reference in the build
container (``tests/golden/make_golden.py``; fixtures ``tests/golden/*.npz``).

Every function cites the reference file:line it follows (paths relative to the reference root).
Two execution styles are offered where it matters:

* ``*_loop`` -- one vector at a time, same loop structure as the reference (this is the
  "reference-equivalent CPU" timed by ``bench.py``'s ``cpu_baseline``);
* batched -- the same arithmetic, vectorised over many vectors/queries.  Reductions are always
  taken over the contiguous last axis so numpy applies the very same pairwise summation per row
  as the reference's ``((x - C) ** 2).sum(axis=1)``.
"""
from __future__ import annotations

from collections import namedtuple

import numpy as np

LOPQCode = namedtuple("LOPQCode", ["coarse", "fine"])  # lopq/lopq/model.py:444


# ----------------------------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------------------------
class OracleModel(object):
    """Plain parameter holder: the attribute set of lopq.model.LOPQModel[PCA].

    Layout as lopq/lopq/model.py:461-473 (and :841 for the PCA variant):
    ``Cs`` 2 x (V,h); ``Rs`` 2 x (V,h,h); ``mus`` 2 x (V,h); ``subquantizers`` 2 x [M/2 x (K,w)];
    optional ``pca_P`` (D_in,D), ``pca_mu`` (D_in,), ``renorm``.
    """

    def __init__(self, Cs, Rs, mus, subquantizers, pca_P=None, pca_mu=None, renorm=False):
        self.Cs = [np.ascontiguousarray(c) for c in Cs]
        self.Rs = [np.ascontiguousarray(r) for r in Rs]
        self.mus = [np.ascontiguousarray(m) for m in mus]
        self.subquantizers = [[np.ascontiguousarray(s) for s in half] for half in subquantizers]
        self.pca_P = None if pca_P is None else np.ascontiguousarray(pca_P)
        self.pca_mu = None if pca_mu is None else np.ascontiguousarray(pca_mu)
        self.renorm = bool(renorm)
        # derived exactly as lopq/lopq/model.py:479-493
        self.V = self.Cs[0].shape[0]
        self.num_coarse_splits = len(self.Cs)
        self.num_fine_splits = len(self.subquantizers[0])
        self.M = self.num_fine_splits * self.num_coarse_splits
        self.subquantizer_clusters = self.subquantizers[0][0].shape[0]

    @property
    def has_pca(self):
        return self.pca_P is not None

    @staticmethod
    def from_npz(z, prefix=""):
        """Rebuild from the flat arrays stored in a golden fixture."""
        g = lambda k: z[prefix + k]
        nf = int(g("num_fine_splits"))
        subs = g("subs")  # (2, nf, K, w)
        has_pca = bool(g("has_pca"))
        return OracleModel(
            Cs=[g("Cs")[0], g("Cs")[1]],
            Rs=[g("Rs")[0], g("Rs")[1]],
            mus=[g("mus")[0], g("mus")[1]],
            subquantizers=[[subs[s, j] for j in range(nf)] for s in range(2)],
            pca_P=g("pca_P") if has_pca else None,
            pca_mu=g("pca_mu") if has_pca else None,
            renorm=bool(g("renorm")),
        )


def uint_type_for(n_clusters):
    """Return dtype chosen by predict_cluster, lopq/lopq/utils.py:48-53."""
    if n_clusters <= 256:
        return np.uint8
    if n_clusters <= 65536:
        return np.uint16
    return np.uint32


# ----------------------------------------------------------------------------------------------
# numpy's pairwise summation, spelled out (documentation + validation of the HIP exact path)
# ----------------------------------------------------------------------------------------------
def np_pairwise_sum(a):
    """Sum a 1-D contiguous float array in the order numpy's add.reduce uses.

    n < 8: sequential from the first element; n <= 128: 8 strided accumulators combined as
    ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and a sequential tail; n > 128: split at
    n2 = n/2 - (n/2 % 8) and recurse.  (behaviour 7.)  Pure Python: small
    inputs only.  The HIP exact kernels implement this very order.
    """
    t = a.dtype.type
    n = a.shape[0]
    if n < 8:
        # numpy starts from -0.0 so that sum([-0.0]) == -0.0 (pairwise sum, n < 8 branch)
        res = t(-0.0)
        for i in range(n):
            res = t(res + a[i])
        return res
    if n <= 128:
        r = [a[k] for k in range(8)]
        i = 8
        while i < n - (n % 8):
            for k in range(8):
                r[k] = t(r[k] + a[i + k])
            i += 8
        res = t(t(t(r[0] + r[1]) + t(r[2] + r[3])) + t(t(r[4] + r[5]) + t(r[6] + r[7])))
        while i < n:
            res = t(res + a[i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return t(np_pairwise_sum(a[:n2]) + np_pairwise_sum(a[n2:]))


# ----------------------------------------------------------------------------------------------
# encode
# ----------------------------------------------------------------------------------------------
def apply_pca(model, x, dtype=np.float32):
    """(x - pca_mu) . pca_P, optional row L2 renorm, cast.  lopq/lopq/model.py:961-978."""
    y = np.dot(x - model.pca_mu, model.pca_P)
    if model.renorm:
        if y.ndim > 1:
            nrm = np.linalg.norm(y, axis=1)
            y = y / nrm[:, np.newaxis]
        else:
            y = y / np.linalg.norm(y)
    return y.astype(dtype)


def _sqdist_rows(x, C):
    """((x - C) ** 2).sum(axis=1) for one vector (lopq/lopq/utils.py:47) or, for a 2-D ``x``
    of n vectors, the (n, n_c) matrix of the same row reductions (last axis contiguous)."""
    if x.ndim == 1:
        return ((x - C) ** 2).sum(axis=1)
    out = np.empty((x.shape[0], C.shape[0]), dtype=np.result_type(x.dtype, C.dtype))
    step = max(1, (1 << 22) // max(1, C.shape[0] * C.shape[1]))
    for a in range(0, x.shape[0], step):
        d = x[a:a + step, None, :] - C[None, :, :]
        np.multiply(d, d, out=d)  # == d ** 2 (numpy squares by multiplying)
        out[a:a + step] = d.sum(axis=2)
    return out


def predict_cluster(x, centroids):
    """argmin of squared distances, first minimum wins.  lopq/lopq/utils.py:33-53."""
    cid = _sqdist_rows(x, centroids).argmin(axis=-1)
    return uint_type_for(centroids.shape[0])(cid) if x.ndim == 1 else cid.astype(
        uint_type_for(centroids.shape[0]))


def split_halves(x, n):
    """Equal contiguous sub-vectors along the last axis.  lopq/lopq/utils.py:8-22."""
    w = x.shape[-1] // n
    return [x[..., s * w:(s + 1) * w] for s in range(n)]


def predict_coarse(model, x):
    """Coarse ids per half.  lopq/lopq/model.py:563-573.  1-D -> tuple, 2-D -> (n,2) array."""
    halves = split_halves(x, model.num_coarse_splits)
    ids = [predict_cluster(np.ascontiguousarray(h), model.Cs[s]) for s, h in enumerate(halves)]
    if x.ndim == 1:
        return tuple(ids)
    return np.stack(ids, axis=1)


def project(model, x, coarse, coarse_split=None):
    """R[c] . ((x_half - C[c]) - mu[c]) per half, concatenated.  lopq/lopq/model.py:604-641."""
    splits = range(model.num_coarse_splits) if coarse_split is None else [coarse_split]
    halves = split_halves(x, model.num_coarse_splits)
    out = []
    for s in splits:
        C, R, mu = model.Cs[s], model.Rs[s], model.mus[s]
        if x.ndim == 1:
            c = coarse[s]
            r = halves[s] - C[c]
            out.append(np.dot(R[c], r - mu[c]))
        else:
            c = np.asarray(coarse)[:, s].astype(np.int64)
            r = halves[s] - C[c]
            v = r - mu[c]
            pr = np.empty(v.shape, dtype=np.float64)
            for cl in np.unique(c):
                sel = np.nonzero(c == cl)[0]
                pr[sel] = np.dot(v[sel], R[cl].T)
            out.append(pr)
    return np.concatenate(out, axis=-1)


def predict_fine(model, x, coarse):
    """Fine codes from the locally projected residual.  lopq/lopq/model.py:575-602."""
    px = project(model, x, coarse)
    codes = []
    for s, half in enumerate(split_halves(px, model.num_coarse_splits)):
        for j, sub in enumerate(split_halves(half, model.num_fine_splits)):
            codes.append(predict_cluster(np.ascontiguousarray(sub), model.subquantizers[s][j]))
    if x.ndim == 1:
        return tuple(codes)
    return np.stack(codes, axis=1)


def predict(model, x):
    """One vector -> LOPQCode.  lopq/lopq/model.py:543-561 and :980-1003 (PCA variant)."""
    if model.has_pca:
        x = apply_pca(model, x)
    coarse = predict_coarse(model, x)
    return LOPQCode(coarse, predict_fine(model, x, coarse))


def compute_codes_loop(model, data):
    """[model.predict(d) for d in data] -- lopq/lopq/utils.py:203-218.  1 core, reference-shaped."""
    return [predict(model, d) for d in data]


def compute_codes(model, data):
    """Batched encode: returns (coarse (n,2) uint, fine (n,M) uint)."""
    x = apply_pca(model, data) if model.has_pca else data
    coarse = predict_coarse(model, x)
    fine = predict_fine(model, x, coarse)
    return coarse, fine


def reconstruct(model, code):
    """R[c]^T . concat(subC[j][f_j]) + mu[c] + C[c] per half.  lopq/lopq/model.py:643-671."""
    coarse, fine = code
    nf = model.num_fine_splits
    parts = []
    for s in range(model.num_coarse_splits):
        sx = np.concatenate([model.subquantizers[s][j][fine[s * nf + j]] for j in range(nf)])
        c = coarse[s]
        r = np.dot(model.Rs[s][c].transpose(), sx) + model.mus[s][c]
        parts.append(r + model.Cs[s][c])
    return np.concatenate(parts)


def subquantizer_distances(model, x, coarse, coarse_split=None):
    """ADC tables: squared distance of each projected sub-vector to its K sub-centroids.
    lopq/lopq/model.py:673-704.  Returns a list of (K,) float64 arrays."""
    px = project(model, x, coarse)
    halves = split_halves(px, model.num_coarse_splits)
    splits = range(model.num_coarse_splits) if coarse_split is None else [coarse_split]
    tabs = []
    for s in splits:
        for j, fx in enumerate(split_halves(halves[s], model.num_fine_splits)):
            tabs.append(_sqdist_rows(np.ascontiguousarray(fx), model.subquantizers[s][j]))
    return tabs


# ----------------------------------------------------------------------------------------------
# search
# ----------------------------------------------------------------------------------------------
def coarse_rank(model, x):
    """Per half: distances to the V coarse centroids and their ascending order.
    lopq/lopq/search.py:37-43.  np.argsort (quicksort) is not stable; for distinct distances the
    order is unique, ties are resolved exactly as numpy does because numpy is what runs here."""
    dists, order = [], []
    for s, cx in enumerate(split_halves(x, len(model.Cs))):
        d = _sqdist_rows(np.ascontiguousarray(cx), model.Cs[s])
        dists.append(d)
        order.append(np.argsort(d))
    return dists, order


def multisequence(model, x):
    """Yield (dist, (c0, c1)) in multi-sequence order.  lopq/lopq/search.py:13-82.

    Restated without a heap: with two splits the traversed set is a Young diagram described by
    t[i] = number of cells already taken in rank-row i.  The reference's heap holds exactly the
    cells (i, t[i]) with t[i] < V and (i == 0 or t[i-1] > t[i]) (push rule :72-82), and pops the
    minimum of the tuple (dist, (i, j)) (:63,:67), dist = (0 + d0[i]) + d1[j] (:50).
    """
    dists, order = coarse_rank(model, x)
    V = model.V
    d0 = dists[0][order[0]]
    d1 = dists[1][order[1]]
    t = [0] * V
    for _ in range(V * V):
        best = None
        for i in range(V):
            j = t[i]
            if j >= V:
                continue
            if i > 0 and t[i - 1] == 0:
                break  # t is non-increasing: every row from here on is still untouched
            if i > 0 and t[i - 1] <= j:
                continue  # (i-1, j) not taken yet, so (i, j) has not been pushed
            key = (sum([d0[i], d1[j]]), i, j)
            if best is None or key < best:
                best = key
        d, i, j = best
        t[i] += 1
        yield d, (order[0][i], order[1][j])


class OracleIndex(object):
    """Dict-of-lists index restating lopq.search.LOPQSearcher (lopq/lopq/search.py:310-382)."""

    def __init__(self, model):
        self.model = model
        self.cells = {}
        self.nb_indexed = 0

    def add_codes(self, codes, ids=None):
        """Append (id, code) to its coarse cell unless that id is already in the cell.
        lopq/lopq/search.py:325-369 (first occurrence wins, insertion order kept)."""
        if ids is None:
            ids = range(len(codes))
        seen = {}
        for item_id, code in zip(ids, codes):
            cell = (int(code[0][0]), int(code[0][1]))
            known = seen.get(cell)
            if known is None:
                known = set(i for i, _ in self.cells.get(cell, ()))
                seen[cell] = known
            if item_id in known:
                continue
            self.cells.setdefault(cell, []).append((item_id, code))
            known.add(item_id)
            self.nb_indexed += 1

    def add_codes_arrays(self, coarse, fine, ids=None):
        codes = [LOPQCode(tuple(c), tuple(f)) for c, f in zip(coarse.tolist(), fine.tolist())]
        self.add_codes(codes, ids)

    def get_cell(self, cell):
        return self.cells.get((int(cell[0]), int(cell[1])), [])

    def get_result_quota(self, x, quota=10):
        """Whole cells in multisequence order until len >= quota; empty cells count as visited.
        lopq/lopq/search.py:110-135."""
        retrieved, visited = [], 0
        for _, cell in multisequence(self.model, x):
            retrieved += self.get_cell(cell)
            visited += 1
            if len(retrieved) >= quota:
                break
        return retrieved, visited

    def compute_distances(self, x, items):
        """ADC: dist = sum_i T[i][fine_i], tables memoised per coarse id and split, summed left
        to right starting from int 0 in float64.  lopq/lopq/search.py:137-177."""
        memo = [{}, {}]
        out = []
        for item in items:
            coarse, fine = item[1]
            tabs = []
            for s in (0, 1):
                c = coarse[s]
                if c not in memo[s]:
                    memo[s][c] = subquantizer_distances(self.model, x, coarse, coarse_split=s)
                tabs += memo[s][c]
            dist = sum([tabs[i][fc] for i, fc in enumerate(fine)])
            out.append((dist, item))
        return out

    def search(self, x, quota=10, limit=None, with_dists=True):
        """PCA (if any), quota retrieval, ADC, stable sort by dist, top ``limit``.
        lopq/lopq/search.py:179-224.  Returns ([(id, code, dist)], visited)."""
        if self.model.has_pca:
            x = apply_pca(self.model, x)
        retrieved, visited = self.get_result_quota(x, quota)
        scored = self.compute_distances(x, retrieved)
        scored = sorted(scored, key=lambda d: d[0])
        if limit is None:
            limit = quota
        scored = scored[:limit]
        if with_dists:
            return [(it[0], it[1], d) for d, it in scored], visited
        return [(it[0], it[1]) for d, it in scored], visited


# ----------------------------------------------------------------------------------------------
# batched search over a CSR index (same results as OracleIndex.search, vectorised per query)
# ----------------------------------------------------------------------------------------------
class OracleCSRIndex(object):
    """Cell-contiguous arrays; candidate order inside a cell = insertion order, so results are
    identical to OracleIndex (stable sort; ties keep retrieval order, search.py:210)."""

    def __init__(self, model, coarse, fine, ids=None):
        self.model = model
        V = model.V
        coarse = np.asarray(coarse).astype(np.int64)
        cell = coarse[:, 0] * V + coarse[:, 1]
        perm = np.argsort(cell, kind="stable")
        self.fine = np.ascontiguousarray(np.asarray(fine)[perm])
        self.ids = (np.arange(len(cell), dtype=np.int64) if ids is None else np.asarray(ids))[perm]
        counts = np.bincount(cell, minlength=V * V)
        self.offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)

    def search(self, x, quota=10, limit=None):
        m = self.model
        if m.has_pca:
            x = apply_pca(m, x)
        V, nf = m.V, m.num_fine_splits
        segs, n, visited = [], 0, 0
        for _, (c0, c1) in multisequence(m, x):
            cid = int(c0) * V + int(c1)
            a, b = self.offsets[cid], self.offsets[cid + 1]
            visited += 1
            if b > a:
                segs.append((int(c0), int(c1), a, b))
                n += b - a
            if n >= quota:
                break
        memo = [{}, {}]
        dists, pos = [], []
        for c0, c1, a, b in segs:
            for s, c in ((0, c0), (1, c1)):
                if c not in memo[s]:
                    memo[s][c] = subquantizer_distances(m, x, (c0, c1), coarse_split=s)
            tabs = memo[0][c0] + memo[1][c1]
            f = self.fine[a:b]
            d = np.zeros(b - a, dtype=np.float64)
            for i in range(2 * nf):  # left-to-right float64 accumulation, search.py:173
                d = d + tabs[i][f[:, i]]
            dists.append(d)
            pos.append(np.arange(a, b))
        if not dists:
            return np.zeros(0, np.int64), np.zeros(0), visited
        dists = np.concatenate(dists)
        pos = np.concatenate(pos)
        order = np.argsort(dists, kind="stable")
        if limit is None:
            limit = quota
        order = order[:limit]
        return self.ids[pos[order]], dists[order], visited


    def search_loop(self, x, quota=10, limit=None):
        """Same result as search(), computed with the reference's per-candidate Python loop
        (compute_distances, lopq/lopq/search.py:166-175): the "reference-equivalent CPU" that
        bench.py times.  Only the visited cells are materialised as (id, code) tuples."""
        m = self.model
        if m.has_pca:
            x = apply_pca(m, x)
        V = m.V
        retrieved, visited = [], 0
        for _, (c0, c1) in multisequence(m, x):
            cid = int(c0) * V + int(c1)
            a, b = self.offsets[cid], self.offsets[cid + 1]
            coarse = (int(c0), int(c1))
            retrieved += [(self.ids[p], (coarse, tuple(self.fine[p]))) for p in range(a, b)]
            visited += 1
            if len(retrieved) >= quota:
                break
        memo = [{}, {}]
        scored = []
        for item in retrieved:
            coarse, fine = item[1]
            tabs = []
            for s in (0, 1):
                c = coarse[s]
                if c not in memo[s]:
                    memo[s][c] = subquantizer_distances(m, x, coarse, coarse_split=s)
                tabs += memo[s][c]
            scored.append((sum([tabs[i][fc] for i, fc in enumerate(fine)]), item))
        scored = sorted(scored, key=lambda d: d[0])
        if limit is None:
            limit = quota
        scored = scored[:limit]
        return (np.array([it[0] for _, it in scored], dtype=np.int64), np.array([d for d, _ in scored]), visited)


def search_exhaustive_blocked(model, coarse, fine, x, limit, ids=None, block=4_000_000):
    """OracleCSRIndex(model, coarse, fine, ids).search(x, quota=len(coarse), limit) WITHOUT grouping the rows by cell -- for indexes
    of hundreds of millions of rows, where the stable sort of the CSR build alone takes minutes.  With an exhaustive quota every
    cell is visited (lopq/lopq/search.py:128-133 stops only when the quota is reached), so the candidates are all rows, a row's
    distance is the left-to-right float64 sum of its M table entries (:166-175; tables memoised per coarse id, :151-164), and the
    stable sorted() of :210 orders equal distances by retrieval order = (visit rank of the row's cell, insertion position inside the
    cell) -- insertion position grows with the row number.  Rows are scored block by block, the best `limit` of every block are
    kept.  Returns (ids, dists, visited) like OracleCSRIndex.search; pinned against it in tests/test_oracle_golden.py."""
    m = model
    if m.has_pca:
        x = apply_pca(m, x)
    V, nf = m.V, m.num_fine_splits
    coarse = np.asarray(coarse)
    n = coarse.shape[0]
    # visit rank of every cell: an exhaustive quota walks the cells until the last non-empty one (empty ones on the way count as visited)
    rank = np.zeros(V * V, dtype=np.int64)
    for r, (_, (c0, c1)) in enumerate(multisequence(m, x)):
        rank[int(c0) * V + int(c1)] = r
    tabs0 = [subquantizer_distances(m, x, (c, 0), coarse_split=0) for c in range(V)]  # [c0][j] -> (K,)
    tabs1 = [subquantizer_distances(m, x, (0, c), coarse_split=1) for c in range(V)]
    T0 = np.stack([np.stack(t) for t in tabs0])  # [V][nf][K]
    T1 = np.stack([np.stack(t) for t in tabs1])
    best_d, best_r, best_i = [], [], []
    last_rank = -1  # the walk stops with the cell that completes the quota: the last NON-EMPTY cell in visit order
    for a in range(0, n, block):
        b = min(n, a + block)
        c0 = coarse[a:b, 0].astype(np.int64)
        c1 = coarse[a:b, 1].astype(np.int64)
        f = np.asarray(fine[a:b])
        d = np.zeros(b - a, dtype=np.float64)
        for i in range(2 * nf):  # left-to-right float64 accumulation, search.py:173
            d = d + (T0[c0, i, f[:, i]] if i < nf else T1[c1, i - nf, f[:, i]])
        rk = rank[c0 * V + c1]
        last_rank = max(last_rank, int(rk.max())) if b > a else last_rank
        k = min(limit, b - a)
        part = np.argpartition(d, k - 1)[:k] if k < b - a else np.arange(b - a)
        # everything that ties with the block's k-th distance must stay in play (the tie is decided by retrieval order)
        cut = d[part].max() if k else 0.0
        keep = np.nonzero(d <= cut)[0]
        best_d.append(d[keep]); best_r.append(rk[keep]); best_i.append(keep + a)
    d = np.concatenate(best_d); rk = np.concatenate(best_r); ii = np.concatenate(best_i)
    order = np.lexsort((ii, rk, d))[:limit]  # (dist, visit rank, row number)
    out_ids = ii[order] if ids is None else np.asarray(ids)[ii[order]]
    return out_ids.astype(np.int64), d[order], last_rank + 1


HIT_DTYPE = np.dtype([("dist", "<f8"), ("visit_rank", "<u4"), ("pos", "<u4"), ("id", "<i8"),
                      ("cell", "<i4"), ("reserved", "<i4")])  # == cis_hit (include/cis_hip.h)


def search_partial(index, x, quota, limit, owner, rank):
    """What ONE shard of a cell-sharded index contributes to a query: the traversal and the quota
    cut use the cell sizes of the whole index (lopq/lopq/search.py:128-133), only cells with
    owner[cell] == rank are scanned.  Returns (hits [limit] HIT_DTYPE padded with id = -1, visited)."""
    m = index.model
    if m.has_pca:
        x = apply_pca(m, x)
    V = m.V
    out = np.zeros(limit, dtype=HIT_DTYPE)
    out["id"] = -1
    out["dist"] = np.inf
    out["cell"] = -1
    rows, n, visited = [], 0, 0
    memo = [{}, {}]
    for _, (c0, c1) in multisequence(m, x):
        cid = int(c0) * V + int(c1)
        a, b = index.offsets[cid], index.offsets[cid + 1]
        if b > a and owner[cid] == rank:
            for s, c in ((0, int(c0)), (1, int(c1))):
                if c not in memo[s]:
                    memo[s][c] = subquantizer_distances(m, x, (int(c0), int(c1)), coarse_split=s)
            tabs = memo[0][int(c0)] + memo[1][int(c1)]
            f = index.fine[a:b]
            d = np.zeros(b - a)
            for i in range(len(tabs)):
                d = d + tabs[i][f[:, i]]
            for p in range(b - a):
                rows.append((d[p], visited, p, index.ids[a + p], cid, 0))
        visited += 1
        n += b - a
        if n >= quota:
            break
    rows.sort(key=lambda r: (r[0], r[1], r[2]))
    for i, r in enumerate(rows[:limit]):
        out[i] = r
    return out, visited


def query_owners(index, x, quota, owner):
    """Routed cell-sharded search, step 0: (bit mask of the ranks that own a non-empty cell among those the multisequence walk
    visits before the quota is reached, visited) -- the walk and cut of search_partial (lopq/lopq/search.py:128-133), no scan."""
    m = index.model
    if m.has_pca:
        x = apply_pca(m, x)
    V = m.V
    mask, n, visited = 0, 0, 0
    for _, (c0, c1) in multisequence(m, x):
        cid = int(c0) * V + int(c1)
        size = int(index.offsets[cid + 1] - index.offsets[cid])
        if size > 0:
            mask |= 1 << int(owner[cid])
        visited += 1
        n += size
        if n >= quota:
            break
    return mask, visited


def merge_partials(parts, limit):
    """Reference merge of per-shard hit lists [world][limit] by (dist, visit_rank, pos)."""
    allh = np.concatenate([p[p["id"] >= 0] for p in parts])
    order = np.lexsort((allh["pos"], allh["visit_rank"], allh["dist"]))
    return allh[order][:limit]


def recall_at(true_nn, result_ids, ks=(1, 10, 100)):
    """Fraction of queries whose true nearest neighbour is in the top-k returned ids.
    Semantics of get_recall, lopq/lopq/eval.py:92-143."""
    out = {}
    for k in ks:
        hits = sum(1 for t, r in zip(true_nn, result_ids) if t in list(r[:k]))
        out[k] = hits / float(len(true_nn))
    return out


def rerank(q, feats_by_id, results, rerank_nb, max_returned=None, near_dup_th=None):
    """Exact re-ranking restated from cufacesearch/cufacesearch/searcher/searcher_lopqhbase.py:864-912.
    results: list of (id, adc_dist) in ADC order; feats_by_id: {id: feature} (a missing id keeps its ADC distance,
    :889-893).  dist = np.linalg.norm(q - feat) in the features' dtype (:887); near-duplicate filter and the
    max_returned cut use the index BEFORE the re-order (:894-899); final order = np.argsort(dists) (:902-903)."""
    results = results[:min(rerank_nb, len(results))]
    ids, dists = [], []
    for ires, (rid, adc) in enumerate(results):
        dist = adc
        if rid in feats_by_id:
            dist = np.linalg.norm(q - feats_by_id[rid])
        if near_dup_th is None or dist <= near_dup_th:
            if not max_returned or ires < max_returned:
                ids.append(rid)
                dists.append(dist)
    order = np.argsort(dists, axis=0, kind="stable") if ids else []
    return [ids[i] for i in order], [dists[i] for i in order]


class OracleKeyOrderIndex(OracleIndex):
    """The LMDB searcher's index semantics (lopq/lopq/search.py:385-499) without LMDB: key = cell + bytes(id) (py2:
    str(id)); put() replaces an existing key (:465); get_cell walks keys in byte order (:482-499); ids come back through
    id_lambda.  **Parity unpinned**: the reference class needs the `lmdb` module, absent here, so no golden vector could
    be generated from it; this follows its source."""

    def __init__(self, model, id_lambda=int):
        OracleIndex.__init__(self, model)
        self.id_lambda = id_lambda
        self.store = {}

    def add_codes(self, codes, ids=None):
        if ids is None:
            ids = range(len(codes))
        for item_id, code in zip(ids, codes):
            cell = (int(code[0][0]), int(code[0][1]))
            self.store.setdefault(cell, {})[str(item_id).encode("latin1")] = LOPQCode(tuple(code[0]), tuple(code[1]))
        self.nb_indexed = sum(len(v) for v in self.store.values())

    def get_cell(self, cell):
        items = self.store.get((int(cell[0]), int(cell[1])), {})
        # py2: key[4:] is a str; the production caller passes id_lambda=str (searcher_lopqhbase.py:204-206)
        return [(self.id_lambda(k.decode("latin1")), items[k]) for k in sorted(items)]
