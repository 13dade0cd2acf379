"""Pin the oracle (oracle/lopq_oracle.py) to the golden vectors made from the real reference.

Bar: bit-exact coarse/fine codes, multisequence cell order, visited counts and ranked candidate
ids; float64 ADC distances to 1e-9 relative (they differ only by BLAS summation order in the
local rotation, lopq/lopq/model.py:638).
"""
import numpy as np
import pytest

from conftest import load_golden, sha1
from oracle import lopq_oracle as O

PCA_FIXTURES = ["c2", "c3", "c3b", "c4", "c3full"]
ALL = ["tiny", "c1"] + PCA_FIXTURES


def _model(z):
    return O.OracleModel.from_npz(z)


def _settings(z):
    out = []
    for k in z:
        if k.startswith("s_") and k.endswith("_ids"):
            tag = k[2:-4]
            q, l = tag.split("_")
            out.append((tag, int(q[1:]), None if l[1:] == "N" else int(l[1:])))
    return out


@pytest.mark.parametrize("name", ALL)
def test_model_shapes_and_dtypes(name):
    z, X, Q = load_golden(name)
    m = _model(z)
    assert m.M == 2 * m.num_fine_splits
    assert m.Rs[0].dtype == np.float64 and m.mus[0].dtype == np.float64
    assert m.subquantizers[0][0].dtype == np.float64
    # coarse centroids inherit the dtype of what LOPQ was trained on (float32 after apply_PCA)
    assert m.Cs[0].dtype == (np.float64 if name == "tiny" else np.float32)


def test_pairwise_sum_model_matches_numpy():
    rs = np.random.RandomState(0)
    for dt in (np.float32, np.float64):
        for n in (1, 5, 7, 8, 9, 16, 31, 32, 64, 100, 128, 129, 136, 256, 300, 2048):
            for _ in range(5):
                a = (rs.randn(n) * 3).astype(dt)
                assert O.np_pairwise_sum(a) == a.sum(), (dt, n)
                sq = ((a - a[::-1].copy()) ** 2)
                assert O.np_pairwise_sum(sq) == sq.sum()


@pytest.mark.parametrize("name", ALL)
def test_encode_codes_bit_exact(name):
    z, X, Q = load_golden(name)
    m = _model(z)
    if name == "c1":
        coarse, fine = O.compute_codes(m, X)
        assert sha1(coarse.astype(np.uint16)) == str(z["coarse_sha1"])
        assert sha1(fine.astype(np.uint8)) == str(z["fine_sha1"])
        np.testing.assert_array_equal(coarse[:4096], z["coarse_head"])
        np.testing.assert_array_equal(fine[:4096], z["fine_head"])
    else:
        n = int(z["n_index"]) if "n_index" in z else len(X)
        coarse, fine = O.compute_codes(m, X[:n])
        np.testing.assert_array_equal(coarse, z["coarse"])
        np.testing.assert_array_equal(fine, z["fine"])
    # the per-vector, reference-shaped loop gives the same codes as the batched restatement
    loop = O.compute_codes_loop(m, X[:200])
    np.testing.assert_array_equal(np.array([c.coarse for c in loop]), coarse[:200])
    np.testing.assert_array_equal(np.array([c.fine for c in loop]), fine[:200])


@pytest.mark.parametrize("name", ["c1"] + PCA_FIXTURES)
def test_near_tie_vectors(name):
    z, X, Q = load_golden(name)
    m = _model(z)
    coarse, fine = O.compute_codes(m, z["tie_X"])
    np.testing.assert_array_equal(coarse, z["tie_coarse"])
    np.testing.assert_array_equal(fine, z["tie_fine"])


@pytest.mark.parametrize("name", ALL)
def test_aux_project_tables_reconstruct(name):
    z, X, Q = load_golden(name)
    m = _model(z)
    n = z["aux_project"].shape[0]
    Xp = O.apply_pca(m, X[:n]) if m.has_pca else X[:n]
    if m.has_pca:
        np.testing.assert_array_equal(O.apply_pca(m, X[:256]), z["aux_pca"])
        single = np.stack([O.apply_pca(m, x) for x in X[:8]])
        np.testing.assert_array_equal(single, z["aux_pca_single"])
    for i in range(n):
        c = O.predict_coarse(m, Xp[i])
        np.testing.assert_allclose(O.project(m, Xp[i], c), z["aux_project"][i], rtol=1e-12, atol=1e-13)
        tabs = np.stack(O.subquantizer_distances(m, Xp[i], c))
        np.testing.assert_allclose(tabs, z["aux_tables"][i], rtol=1e-11, atol=1e-13)
        code = (c, O.predict_fine(m, Xp[i], c))
        np.testing.assert_allclose(O.reconstruct(m, code), z["aux_reconstruct"][i], rtol=1e-11, atol=1e-12)


def _build_index(name, z, X, m, csr):
    if name == "tiny":
        sel = z["sel"]
        coarse, fine = z["coarse"], z["fine"]
        idx = O.OracleIndex(m)
        codes = [O.LOPQCode(tuple(coarse[i]), tuple(fine[i])) for i in range(len(coarse))]
        ids = z["ids"].tolist()
        idx.add_codes([codes[i] for i in sel], ids)
        assert idx.nb_indexed == int(z["nb_after_first"])
        idx.add_codes([codes[i] for i in sel[:50]], ids[:50])
        assert idx.nb_indexed == int(z["nb_after_readd"])
        idx.add_codes([codes[sel[0]]] * 20, z["dup_ids"].tolist())
        assert idx.nb_indexed == int(z["nb_indexed"])
        if csr:
            # flatten in per-cell insertion order for the CSR variant
            all_ids, co, fi = [], [], []
            for cell, items in idx.cells.items():
                for iid, code in items:
                    all_ids.append(iid), co.append(code.coarse), fi.append(code.fine)
            return O.OracleCSRIndex(m, np.array(co), np.array(fi, dtype=np.uint8), np.array(all_ids))
        return idx
    if name == "c1":
        coarse, fine = O.compute_codes(m, X)
    else:
        coarse, fine = z["coarse"], z["fine"]
    if csr:
        return O.OracleCSRIndex(m, coarse, fine)
    idx = O.OracleIndex(m)
    idx.add_codes_arrays(coarse, fine)
    assert idx.nb_indexed == int(z["nb_indexed"])
    return idx


@pytest.mark.parametrize("name", ALL)
def test_multisequence_order(name):
    z, X, Q = load_golden(name)
    m = _model(z)
    cells = z["multiseq_cells"]
    Qx = np.concatenate([Q, X[z["sel"][:4]]]) if name == "tiny" else Q
    for qi in range(cells.shape[0]):
        x = O.apply_pca(m, Qx[qi]) if m.has_pca else Qx[qi]
        got = []
        for k, (d, cell) in enumerate(O.multisequence(m, x)):
            if k >= cells.shape[1]:
                break
            got.append((int(cell[0]), int(cell[1])))
            assert d == z["multiseq_dists"][qi, k]
        assert got == [tuple(c) for c in cells[qi][:len(got)].tolist()]
        if m.V * m.V <= cells.shape[1]:
            assert len(set(got)) == m.V * m.V  # every cell exactly once


@pytest.mark.parametrize("csr", [False, True])
@pytest.mark.parametrize("name", ALL)
def test_search_ids_bit_exact(name, csr):
    z, X, Q = load_golden(name)
    m = _model(z)
    idx = _build_index(name, z, X, m, csr)
    Qx = np.concatenate([Q, X[z["sel"][:4]]]) if name == "tiny" else Q
    nq = z["multiseq_cells"].shape[0]
    for tag, quota, limit in _settings(z):
        if name == "c1" and not csr and quota >= 10000:
            continue  # per-item python loop over 11k candidates x 100 queries: covered by csr=True
        for qi in range(nq):
            if csr:
                ids, dists, visited = idx.search(Qx[qi], quota=quota, limit=limit)
            else:
                res, visited = idx.search(Qx[qi], quota=quota, limit=limit)
                ids = np.array([r[0] for r in res], dtype=np.int64)
                dists = np.array([r[2] for r in res])
            n = int(z["s_%s_n" % tag][qi])
            assert visited == int(z["s_%s_visited" % tag][qi])
            assert len(ids) == n
            np.testing.assert_array_equal(ids, z["s_%s_ids" % tag][qi, :n])
            np.testing.assert_allclose(dists, z["s_%s_dists" % tag][qi, :n], rtol=1e-9, atol=1e-12)


def test_tiny_edge_semantics():
    """Behaviours 2-5 of SURVEY.md section 8a, checked on the golden data itself."""
    z, X, Q = load_golden("tiny")
    # whole-cell consumption with limit=None => limit=quota
    assert (z["s_q1_lN_n"] <= 1).all() and (z["s_q1_lN_retrieved"] >= z["s_q1_lN_n"]).all()
    # empty cells count as visited
    assert (z["s_q1_lN_visited"] >= 1).all() and z["s_q1_lN_visited"].max() > 1
    # exact ties (20 copies of one code) come back in insertion order
    ids = z["s_q100000_l40_ids"][-4]
    dd = z["s_q100000_l40_dists"][-4]
    dup = z["dup_ids"]
    first = int(z["ids"][0])
    tied = [i for i, d in zip(ids.tolist(), dd.tolist()) if d == dd[0]]
    assert tied[0] == first and tied[1:21] == dup.tolist()


@pytest.mark.parametrize("name", ["tiny", "c2", "c4"])
def test_exhaustive_blocked_search_equals_the_csr_index(name):
    """oracle.search_exhaustive_blocked (what checks the HBM-streaming route at 200 M rows, where the CSR build's sort is too slow)
    == OracleCSRIndex.search(quota = N) -- itself pinned to the reference's ranked ids above -- incl. a block of exact duplicates that
    straddles the block boundary (ties by retrieval order) and rows in a cell order that differs from the row order."""
    z, X, Q = load_golden(name)
    om = _model(z)
    coarse, fine = z["coarse"].copy(), z["fine"].copy()
    n = coarse.shape[0]
    oc, of = O.compute_codes(om, Q[:1])
    coarse[n // 2 - 40:n // 2 + 40] = oc[0]   # 80 copies of the first query's own code around the middle of the rows
    fine[n // 2 - 40:n // 2 + 40] = of[0]
    oi = O.OracleCSRIndex(om, coarse, fine)
    for qi in range(3):
        for limit in (10, 100):
            ids, dists, visited = oi.search(Q[qi], quota=n, limit=limit)
            for block in (n // 2, 997, n):  # the first splits the duplicates over two blocks
                bi, bd, bv = O.search_exhaustive_blocked(om, coarse, fine, Q[qi], limit, block=block)
                assert (bi == ids).all() and np.array_equal(bd, dists) and bv == visited
