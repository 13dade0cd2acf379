"""GPU parity: the HIP path (through the C ABI) against the golden vectors and the oracle.

Bars (north_star): coarse/fine codes, cell order (visited), candidate ids bit-exact; float64 ADC
distances within 1e-9 relative (north_star asks 1e-4); apply_PCA float32 within 1 ulp.
"""
import os
import numpy as np
import pytest

from conftest import load_golden, sha1

pytestmark = pytest.mark.gpu

PCA_FIXTURES = ["c2", "c3", "c3b", "c4", "c3full"]
ALL = ["tiny", "c1"] + PCA_FIXTURES

_ROUTE_FREE = ("encode", "alternative_kernels", "near_tie", "fine_codes", "hooks", "model_pieces", "selftest", "multisequence", "rerank", "kmeans", "training", "multisequence_plan", "select_path",
               "large_limit", "fuzz")


@pytest.fixture(autouse=True, params=["auto", "prefilter", "scan3", "scan4", "scan5", "stream", "scan7"])
def route(request):
    """Every search test runs on the three routes of limit <= 440: "auto" (small batches -- what most fixtures are -- take
    the all-candidates path: exact distances + radix select), "prefilter" (the float32-prefilter scan kernel k_adc_scan2
    whatever the batch size), "scan3" / "scan4" (the 16-bit fixed-point kernel k_adc_scan3 whatever the batch size, in its
    streaming and in its two-pass form), "scan5" (the sampled single-pass form k_adc_scan4: what large batches over short
    cells run), "stream" (the HBM-streaming route of csrc/lopq_stream.hip, scan mode 6: what a few queries over hundreds of
    thousands of candidates each take -- forced here onto the fixtures' small cells, with 1024-candidate chunks), "scan7" (scan mode
    7 = k_adc_scan5: one sampled threshold per query for the whole batch, eight queries per slot -- what large batches run by default)."""
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    if request.param != "auto" and any(k in request.node.name for k in _ROUTE_FREE):
        pytest.skip("does not depend on the search route")
    LOPQSearcherHIP.default_prefilter_only = request.param == "prefilter"
    LOPQSearcherHIP.default_scan_mode = {"scan3": 3, "scan4": 4, "scan5": 5, "stream": 6, "scan7": 7}.get(request.param, 0)
    if request.param == "stream":
        os.environ["CIS_STREAM_SEG"] = "1024"
    yield request.param
    os.environ.pop("CIS_STREAM_SEG", None)
    LOPQSearcherHIP.default_prefilter_only = False
    LOPQSearcherHIP.default_scan_mode = 0


def hip_model(z):
    from columbiaimagesearch_amd.lopq import LOPQModel, LOPQModelPCA
    nf = int(z["num_fine_splits"])
    subs = tuple([z["subs"][s, j] for j in range(nf)] for s in range(2))
    base = ((z["Cs"][0], z["Cs"][1]), (z["Rs"][0], z["Rs"][1]), (z["mus"][0], z["mus"][1]), subs)
    if bool(z["has_pca"]):
        return LOPQModelPCA(renorm=bool(z["renorm"]), parameters=base + (z["pca_P"], z["pca_mu"]))
    return LOPQModel(parameters=base)


def _settings(z):
    out = []
    for k in z:
        if k.startswith("s_") and k.endswith("_ids"):
            tag = k[2:-4]
            q, l = tag.split("_")
            out.append((tag, int(q[1:]), None if l[1:] == "N" else int(l[1:])))
    return out


@pytest.mark.parametrize("name", ALL)
def test_encode_bit_exact(name):
    z, X, Q = load_golden(name)
    m = hip_model(z)
    if name == "c1":
        coarse, fine = m.predict_batch(X)
        assert sha1(coarse.astype(np.uint16)) == str(z["coarse_sha1"])
        assert sha1(fine.astype(np.uint8)) == str(z["fine_sha1"])
    else:
        n = int(z["n_index"]) if "n_index" in z else len(X)
        coarse, fine = m.predict_batch(X[:n])
        assert (coarse != z["coarse"]).sum() == 0
        assert (fine != z["fine"]).sum() == 0
    one = m.predict(X[3])  # the reference's per-vector entry point
    assert tuple(int(c) for c in one.coarse) == tuple(int(c) for c in coarse[3])
    assert tuple(int(f) for f in one.fine) == tuple(int(f) for f in fine[3])


def test_fine_codes_prefilter_adversarial(monkeypatch):
    """Fine codes on the matrix cores (k_fine_mfma: float32 prefilter + exact re-check) against the oracle and against the
    all-pairs VALU kernel (CIS_FINE=0) on inputs built to stress the prefilter: many identical sub-centroids (the wave's list
    overflows -> all-pairs path, first index wins), vectors whose projected residual sits ON a sub-centroid or next to the
    midpoint of two (near-ties far below the float32 prefilter's resolution), huge and tiny scales, ragged counts, K not a multiple of 32,
    w = 32, 16, 8, 4."""
    from oracle import lopq_oracle as O
    from columbiaimagesearch_amd.lopq import LOPQModel
    rs = np.random.RandomState(11)
    for (V, M, K, D, scale) in [(4, 8, 256, 128, 1.0), (4, 4, 256, 128, 1e-3), (3, 16, 200, 128, 1e3), (4, 8, 40, 32, 1.0), (4, 32, 256, 128, 1.0)]:
        h, w, nf = D // 2, D // M, M // 2
        Cs = [rs.randn(V, h) * scale for _ in range(2)]
        Rs = [np.stack([np.linalg.qr(rs.randn(h, h))[0] for _ in range(V)]) for _ in range(2)]
        mus = [rs.randn(V, h) * 0.05 * scale for _ in range(2)]
        subs = [[rs.randn(K, w) * 0.6 * scale for _ in range(nf)] for _ in range(2)]
        subs[0][0][K // 2:] = subs[0][0][3]          # half of one codebook is ONE centroid: hundreds of candidates per vector
        subs[1][nf - 1][7] = subs[1][nf - 1][2]      # an exact duplicate pair
        m = LOPQModel(parameters=(tuple(Cs), tuple(Rs), tuple(mus), tuple(tuple(x) for x in subs)))
        om = O.OracleModel(Cs, Rs, mus, subs)
        n = 777
        X = rs.randn(n, D) * scale
        for i in range(300):  # projection R[c] ((x - C[c]) - mu[c]) == target  <=>  x = C[c] + mu[c] + R[c]^T target
            for s_ in range(2):
                c = i % V
                # (midpoints of two sub-centroids would be ties that the rounding of the ROTATION decides -- BLAS order in the
                # reference, k-ascending fma chains here, DESIGN.md section 3 -- so they are 1 % off the midpoint)
                tgt = np.concatenate([subs[s_][j][(i + j) % K] if i % 2 == 0 else 0.505 * subs[s_][j][i % K] + 0.495 * subs[s_][j][(i * 7 + 1) % K]
                                      for j in range(nf)])
                X[i, s_ * h:(s_ + 1) * h] = Cs[s_][c] + mus[s_][c] + Rs[s_][c].T @ tgt
        oc, of = O.compute_codes(om, X)
        for route in ("1", "0"):
            monkeypatch.setenv("CIS_FINE", route)
            coarse, fine = m.predict_batch(X)
            np.testing.assert_array_equal(coarse, oc)
            np.testing.assert_array_equal(fine, of, err_msg="CIS_FINE=%s shape %r" % (route, (V, M, K, D, scale)))
        monkeypatch.delenv("CIS_FINE")


@pytest.mark.parametrize("name", ["c1"] + PCA_FIXTURES)
def test_near_tie_vectors(name):
    z, X, Q = load_golden(name)
    m = hip_model(z)
    coarse, fine = m.predict_batch(z["tie_X"])
    np.testing.assert_array_equal(coarse, z["tie_coarse"])
    np.testing.assert_array_equal(fine, z["tie_fine"])


@pytest.mark.parametrize("name", PCA_FIXTURES)
def test_encode_pca_forms_agree(monkeypatch, name):
    """The PCA product with the loads-only fetch (k_pca_gemm_mfma_pf, round 4) against the border-checked forms (CIS_PCA_PF=0):
    same instruction and k order per output element -- the same bits, on a ragged row count that leaves partial tiles."""
    z, X, Q = load_golden(name)
    m = hip_model(z)
    Xr = np.concatenate([X, X[::-1]])[:max(1, (2 * len(X)) - 7)]
    a = m.apply_PCA(Xr)
    ca, fa = m.predict_batch(Xr)
    monkeypatch.setenv("CIS_PCA_PF", "0")
    b = m.apply_PCA(Xr)
    cb, fb = m.predict_batch(Xr)
    np.testing.assert_array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))
    np.testing.assert_array_equal(ca, cb)
    np.testing.assert_array_equal(fa, fb)


@pytest.mark.parametrize("name", PCA_FIXTURES)
def test_pca_of_a_few_rows_equals_the_batched_form(name):
    """apply_PCA of one to eight rows runs in one launch (k_pca_small, round 6); its bits are the batched MFMA form's: the same chain of
    fused multiply-adds over ascending k per output element, the same centring, norm order, division and cast.  Both input types."""
    z, X, Q = load_golden(name)
    m = hip_model(z)
    for Xs in (X[:64], X[:64].astype(np.float64) if X.dtype == np.float32 else X[:64].astype(np.float32)):
        big = np.asarray(m.apply_PCA(Xs)).view(np.uint32)          # 64 rows: the batched form
        for a, b in ((0, 1), (5, 6), (10, 13), (20, 28), (63, 64)):  # 1, 1, 3, 8, 1 rows: the small form where the model's shape allows
            got = np.asarray(m.apply_PCA(Xs[a:b])).view(np.uint32)
            np.testing.assert_array_equal(got, big[a:b])
        np.testing.assert_array_equal(np.asarray(m.apply_PCA(Xs[7])).view(np.uint32), big[7])   # a 1-D vector


@pytest.mark.parametrize("name", ALL)
def test_model_pieces(name):
    from oracle import lopq_oracle as O
    z, X, Q = load_golden(name)
    m = hip_model(z)
    om = O.OracleModel.from_npz(z)
    n = z["aux_project"].shape[0]
    if om.has_pca:
        got = m.apply_PCA(X[:256])
        ref = z["aux_pca"]
        assert got.dtype == np.float32
        # float32 output of a float64 product: equal up to 1 ulp, and almost always identical
        np.testing.assert_allclose(got, ref, rtol=2e-7, atol=1e-9)
        assert (got != ref).mean() < 1e-3
        np.testing.assert_allclose(m.apply_PCA(X[5]), ref[5], rtol=2e-7, atol=1e-9)
        Xp = ref[:n]
    else:
        Xp = X[:n]
    coarse = m.predict_coarse(Xp)
    ref_coarse = O.predict_coarse(om, Xp)
    np.testing.assert_array_equal(coarse, ref_coarse)
    np.testing.assert_allclose(m.project(Xp, coarse), z["aux_project"], rtol=1e-10, atol=1e-12)
    np.testing.assert_array_equal(m.predict_fine(Xp, coarse), O.predict_fine(om, Xp, ref_coarse))
    for i in range(n):
        tabs = np.stack(m.get_subquantizer_distances(Xp[i], coarse[i]))
        np.testing.assert_allclose(tabs, z["aux_tables"][i], rtol=1e-9, atol=1e-12)
        half = np.stack(m.get_subquantizer_distances(Xp[i], coarse[i], coarse_split=1))
        np.testing.assert_array_equal(half, tabs[m.num_fine_splits:])
        code = (tuple(coarse[i]), m.predict_fine(Xp[i], coarse[i]))
        np.testing.assert_allclose(m.reconstruct(code), z["aux_reconstruct"][i], rtol=1e-10, atol=1e-12)


def _build_searcher(name, z, X, m):
    from columbiaimagesearch_amd.lopq import LOPQCode, LOPQSearcherHIP
    s = LOPQSearcherHIP(m)
    if name == "tiny":
        coarse, fine, sel = z["coarse"], z["fine"], z["sel"]
        codes = [LOPQCode(tuple(coarse[i]), tuple(fine[i])) for i in range(len(coarse))]
        ids = z["ids"].tolist()
        s.add_codes([codes[i] for i in sel], ids)
        assert s.get_nb_indexed() == int(z["nb_after_first"])
        s.add_codes([codes[i] for i in sel[:50]], ids[:50])
        assert s.get_nb_indexed() == int(z["nb_after_readd"])
        s.add_codes([codes[sel[0]]] * 20, z["dup_ids"].tolist())
    elif name == "c1":
        coarse, fine = m.predict_batch(X)
        s.add_codes_array(coarse, fine)
    else:
        s.add_codes_array(z["coarse"], z["fine"])
    assert s.get_nb_indexed() == int(z["nb_indexed"])
    return s


@pytest.mark.parametrize("name", ALL)
def test_search_matches_reference(name):
    z, X, Q = load_golden(name)
    m = hip_model(z)
    s = _build_searcher(name, z, X, m)
    Qx = np.concatenate([Q, X[z["sel"][:4]]]) if name == "tiny" else Q
    nq = z["multiseq_cells"].shape[0]
    for tag, quota, limit in _settings(z):
        r = s.search_batch(Qx[:nq], quota=quota, limit=limit)
        np.testing.assert_array_equal(r["visited"], z["s_%s_visited" % tag])
        np.testing.assert_array_equal(r["n_found"], z["s_%s_n" % tag])
        L = z["s_%s_ids" % tag].shape[1]
        np.testing.assert_array_equal(r["ids"][:, :L], z["s_%s_ids" % tag])
        ref_d = z["s_%s_dists" % tag]
        ok = ~np.isnan(ref_d)
        np.testing.assert_allclose(r["dists"][:, :L][ok], ref_d[ok], rtol=1e-9, atol=1e-12)
        assert np.isnan(r["dists"][:, :L][~ok]).all()
    # the reference's one-query surface: Result(id, code, dist)
    tag, quota, limit = _settings(z)[0]
    res, visited = s.search(Qx[0], quota=quota, limit=limit, with_dists=True)
    n = int(z["s_%s_n" % tag][0])
    assert visited == int(z["s_%s_visited" % tag][0]) and len(res) == n
    assert [r.id for r in res] == z["s_%s_ids" % tag][0, :n].tolist()
    for r_ in res:
        items = dict((i, c) for i, c in s.get_cell(r_.code.coarse))
        assert tuple(items[r_.id].fine) == tuple(r_.code.fine)
    res2, _ = s.search(Qx[0], quota=quota, limit=limit)
    assert res2 and res2[0]._fields == ("id", "code")


def test_tiny_cells_ids_and_errors():
    from columbiaimagesearch_amd.lopq import LOPQCode, LOPQSearcherHIP
    z, X, Q = load_golden("tiny")
    m = hip_model(z)
    s = _build_searcher("tiny", z, X, m)
    coarse, fine, sel = z["coarse"], z["fine"], z["sel"]
    cell = tuple(int(c) for c in coarse[sel[0]])
    items = s.get_cell(cell)
    assert items[0][0] == int(z["ids"][0])
    assert [i for i, _ in items][-20:] == z["dup_ids"].tolist()  # insertion order
    assert all(tuple(int(c) for c in code.coarse) == cell for _, code in items)
    assert s.get_cell((1, 2)) == []  # never filled
    # string ids (sha1_bbox style) and dict insertion
    s2 = LOPQSearcherHIP(m)
    d = {"sha1_%d_0_0_10_10" % i: [tuple(coarse[i]), tuple(fine[i])] for i in sel[:40]}
    s2.add_codes_from_dict(d)
    s2.add_codes_from_dict(d)  # second time: all duplicates
    assert s2.get_nb_indexed() == 40
    res, _ = s2.search(X[sel[0]], quota=5, limit=3, with_dists=True)
    assert res[0].id == "sha1_%d_0_0_10_10" % sel[0]
    # a broken code is reported and skipped, the rest goes in (reference search.py:365-367)
    s3 = LOPQSearcherHIP(m)
    s3.add_codes([LOPQCode((0, 0), (1, 2, 3, 4)), LOPQCode((9, 9), (1, 2, 3, 4)), LOPQCode((0, 1), (1, 2, 3))])
    assert s3.get_nb_indexed() == 1
    # empty index: every cell is visited, nothing found
    s4 = LOPQSearcherHIP(m)
    r = s4.search_batch(Q[:2], quota=10, limit=5)
    assert (r["visited"] == m.V * m.V).all() and (r["n_found"] == 0).all()
    with pytest.raises(ValueError):
        s4.search_batch(np.zeros((1, 5)), quota=1)


def test_public_hooks_get_result_quota_and_compute_distances():
    """search.py:110-177 as callable hooks on the HIP searcher: same retrieved items / visited count / distances as the
    oracle's dict index, and search() == sorted(compute_distances(get_result_quota()))[:limit]."""
    from oracle import lopq_oracle as O
    z, X, Q = load_golden("c1")
    m = hip_model(z)
    s = _build_searcher("c1", z, X, m)
    oi = O.OracleIndex(O.OracleModel.from_npz(z))
    coarse, fine = m.predict_batch(X)
    oi.add_codes_arrays(coarse, fine)
    for qi in range(3):
        x = Q[qi]
        items, visited = s.get_result_quota(x, quota=300)
        want_items, want_visited = oi.get_result_quota(x, quota=300)
        assert visited == want_visited and [i for i, _ in items] == [i for i, _ in want_items]
        scored = s.compute_distances(x, items)
        want = oi.compute_distances(x, want_items)
        np.testing.assert_allclose([d for d, _ in scored], [d for d, _ in want], rtol=1e-9)
        res, vis = s.search(x, quota=300, limit=20, with_dists=True)
        top = sorted(scored, key=lambda t: t[0])[:20]
        assert vis == visited and [r.id for r in res] == [it[0] for _, it in top]


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", ["tiny", "c2"])
def test_cell_sharded_search_equals_single(name, world):
    """Each shard scans its own cells with the replicated cell-size table; merging the partial lists
    by (dist, visit_rank, pos) reproduces the single-index result exactly."""
    import torch
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    from columbiaimagesearch_amd.lopq.search import merge_hits_dev
    z, X, Q = load_golden(name)
    m = hip_model(z)
    single = _build_searcher(name, z, X, m)
    if name == "tiny":
        coarse, fine = z["coarse"][z["sel"]], z["fine"][z["sel"]]
        ids = z["ids"]
    else:
        coarse, fine, ids = z["coarse"], z["fine"], None
    shards = []
    for r in range(world):
        s = LOPQSearcherHIP(m, shard=(r, world))
        s.add_codes_array(coarse, fine, ids)
        if name == "tiny":
            s.add_codes_array(np.repeat(coarse[:1], 20, 0), np.repeat(fine[:1], 20, 0), z["dup_ids"])
        assert s.get_nb_indexed() == single.get_nb_indexed()
        shards.append(s)
    q = torch.as_tensor(Q).cuda().contiguous()
    for quota, limit in [(10, 10), (300, 40)]:
        ref = single.search_batch(Q, quota=quota, limit=limit)
        parts, vis = [], []
        for s in shards:
            h, v = s.search_partial_dev(q, quota=quota, limit=limit)
            parts.append(h)
            vis.append(v.cpu().numpy())
        out = merge_hits_dev(torch.stack(parts).contiguous())
        torch.cuda.synchronize()
        for v in vis:
            np.testing.assert_array_equal(v, ref["visited"])
        np.testing.assert_array_equal(out["ids"].cpu().numpy(), ref["ids"])
        np.testing.assert_array_equal(out["n_found"].cpu().numpy(), ref["n_found"])
        a, b = out["dists"].cpu().numpy(), ref["dists"]
        np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
        np.testing.assert_array_equal(a[~np.isnan(a)], b[~np.isnan(b)])
        # the packed exchange format (what ShardedSearcher sends over RCCL): valid hits only, padded to the largest shard
        from columbiaimagesearch_amd.lopq.search import merge_packed_dev
        from ref_merge import pack_hits_dev
        packed = [pack_hits_dev(h) for h in parts]
        stride = max(max(int(pk.shape[0]) for pk, _ in packed), 1)
        buf = torch.zeros((world, stride, 4), dtype=torch.int64, device="cuda")
        for r, (pk, _) in enumerate(packed):
            buf[r, :pk.shape[0]] = pk
        cnt = torch.stack([c for _, c in packed]).contiguous()
        off = (torch.cumsum(cnt, dim=1, dtype=torch.int64) - cnt).contiguous()
        out2 = merge_packed_dev(buf, off, cnt, len(Q), limit)
        torch.cuda.synchronize()
        for sh, (pk, c) in zip(shards, packed):  # the library's own packing == packing the dense partial list
            pp = sh.search_partial_packed_dev(q, quota=quota, limit=limit)
            tot = int(pp["total"].item())
            assert tot == int(pk.shape[0])
            assert torch.equal(pp["packed"][:tot], pk) and torch.equal(pp["cnt"], c)
            assert torch.equal(pp["off"], torch.cumsum(c, 0, dtype=torch.int64) - c)
        np.testing.assert_array_equal(out2["ids"].cpu().numpy(), ref["ids"])
        np.testing.assert_array_equal(out2["n_found"].cpu().numpy(), ref["n_found"])
        a2 = out2["dists"].cpu().numpy()
        np.testing.assert_array_equal(np.isnan(a2), np.isnan(b))
        np.testing.assert_array_equal(a2[~np.isnan(a2)], b[~np.isnan(b)])


@pytest.mark.gpu
def test_sharded_any_limit_merge_equals_single():
    """limit above the 512 records per query of the one-wave merge kernel (and above the 3072 of the dense one): the
    shards' packed lists are merged by cis_merge_packed_dev at any limit -- same result as one index."""
    import torch
    from ref_merge import merge_packed_sorted
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    z, X, Q = load_golden("c2")
    m = hip_model(z)
    single = _build_searcher("c2", z, X, m)
    world = 3
    shards = []
    for r in range(world):
        s = LOPQSearcherHIP(m, shard=(r, world))
        s.add_codes_array(z["coarse"], z["fine"], None)
        shards.append(s)
    q = torch.as_tensor(Q[:9]).cuda().contiguous()
    for quota, limit in [(5000, 700), (20000, 4000), (2500, None)]:
        ref = single.search_batch(Q[:9], quota=quota, limit=limit)
        L = ref["ids"].shape[1]
        pp = [s.search_partial_packed_dev(q, quota=quota, limit=limit) for s in shards]
        torch.cuda.synchronize()
        stride = max(max(int(p["total"].item()) for p in pp), 1)
        buf = torch.zeros((world, stride, 4), dtype=torch.int64, device="cuda")
        for r, p in enumerate(pp):
            t = int(p["total"].item())
            buf[r, :t] = p["packed"][:t]
        cnt = torch.stack([p["cnt"] for p in pp]).contiguous()
        off = torch.stack([p["off"] for p in pp]).contiguous()
        from columbiaimagesearch_amd.lopq.search import merge_packed_dev
        # the product merge (HIP: one wave per query up to 3072 records, places by binary search above) and the test-side
        # restatement with three stable torch sorts, both against the single index
        for out in (merge_packed_dev(buf, off, cnt, 9, L, with_codes=True), merge_packed_sorted(buf, off, cnt, 9, L)):
            np.testing.assert_array_equal(out["ids"].cpu().numpy(), ref["ids"])
            np.testing.assert_array_equal(out["n_found"].cpu().numpy(), ref["n_found"])
            a, b = out["dists"].cpu().numpy(), ref["dists"]
            np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
            np.testing.assert_array_equal(a[~np.isnan(a)], b[~np.isnan(b)])


def test_device_entry_points_match_host_entry_points():
    import torch
    z, X, Q = load_golden("c2")
    m = hip_model(z)
    s = _build_searcher("c2", z, X, m)
    ref = s.search_batch(Q, quota=1000, limit=100)
    out = s.search_batch_dev(torch.as_tensor(Q).cuda(), quota=1000, limit=100)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out["ids"].cpu().numpy(), ref["ids"])
    np.testing.assert_array_equal(out["visited"].cpu().numpy(), ref["visited"])
    st = s.last_stats()
    assert st["candidates"] >= int(z["s_q1000_l100_retrieved"].sum()) > 0 and st["items"] > 0
    c, f = m.predict_batch_dev(torch.as_tensor(X[:5000]).cuda())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(c.cpu().numpy().view(np.uint16), z["coarse"][:5000])
    np.testing.assert_array_equal(f.cpu().numpy(), z["fine"][:5000])


@pytest.mark.parametrize("name", ["c1", "c2", "c3", "c4"])
def test_fast_and_exact_scan_kernels_agree(name):
    """The float32-prefilter scan (exact float64 re-scoring) and the float64 scan return identical
    ids AND bit-identical distances; M = 4, 8, 16 cover the three lane-rotation layouts."""
    z, X, Q = load_golden(name)
    m = hip_model(z)
    s = _build_searcher(name, z, X, m)
    for quota, limit in [(10, 10), (1000, 100), (10000, 100), (20000, 300), (100000, 37)]:
        s.set_scan_mode(exact_only=False)
        a = s.search_batch(Q, quota=quota, limit=limit)
        s.set_scan_mode(exact_only=True)
        b = s.search_batch(Q, quota=quota, limit=limit)
        s.set_scan_mode(exact_only=False)
        for k in ("ids", "n_found", "visited"):
            np.testing.assert_array_equal(a[k], b[k])
        np.testing.assert_array_equal(a["dists"].view(np.uint64), b["dists"].view(np.uint64))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c2", "c3", "c4"])
def test_prefilter_scan_with_16_register_regions_matches_exact_kernel_and_oracle(name):
    """limit 441 ... 952: the float32-prefilter kernel with 16 region registers per lane (one query per workgroup) and the
    2048-key survivor merge, against the float64 scan kernel (bit-identical) and the oracle; 953 is back on the all-candidates
    path.  M = 8 and 16."""
    from oracle import lopq_oracle as O
    z, X, Q = load_golden(name)
    m = hip_model(z)
    s = _build_searcher(name, z, X, m)
    om = O.OracleModel.from_npz(z)
    oi = O.OracleCSRIndex(om, z["coarse"], z["fine"])
    for quota, limit in [(5000, 441), (100000, 700), (100000, 952), (100000, 953)]:
        s.set_scan_mode(prefilter_only=True)
        a = s.search_batch(Q, quota=quota, limit=limit)
        kern = s.last_stats()["scan_kernel"]
        assert kern == ("k_adc_scan2" if limit <= 952 else None), (limit, kern)
        s.set_scan_mode(exact_only=True)
        b = s.search_batch(Q, quota=quota, limit=limit)
        s.set_scan_mode(exact_only=False)
        for k in ("ids", "n_found", "visited"):
            np.testing.assert_array_equal(a[k], b[k])
        np.testing.assert_array_equal(a["dists"].view(np.uint64), b["dists"].view(np.uint64))
        for qi in range(0, len(Q), max(1, len(Q) // 6)):
            ids, dists, visited = oi.search(Q[qi], quota=quota, limit=limit)
            n = len(ids)
            assert a["n_found"][qi] == n and a["visited"][qi] == visited
            np.testing.assert_array_equal(a["ids"][qi, :n], ids)
            np.testing.assert_allclose(a["dists"][qi, :n], dists, rtol=1e-9, atol=1e-12)


def test_many_exact_ties_fall_back_to_exact_kernel():
    """Thousands of identical codes in one cell tie exactly in float32 and float64: the fast kernel
    must hand those work items to the exact kernel, and ties come back in insertion order."""
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    z, X, Q = load_golden("c2")
    m = hip_model(z)
    coarse, fine = z["coarse"], z["fine"]
    s = LOPQSearcherHIP(m)
    n_dup = 5000
    dc = np.repeat(coarse[:1], n_dup, 0)
    df = np.repeat(fine[:1], n_dup, 0)
    s.add_codes_array(np.concatenate([coarse[:3000], dc, coarse[3000:6000]]),
                      np.concatenate([fine[:3000], df, fine[3000:6000]]),
                      np.arange(3000 + n_dup + 3000, dtype=np.int64) + 10)
    q = X[:1]  # the duplicated vector itself
    s.set_scan_mode(prefilter_only=True)
    a = s.search_batch(q, quota=50000, limit=100)
    s.set_scan_mode(exact_only=False, prefilter_only=False)  # the all-candidates path (a batch of one): tie crowd cut by retrieval order
    c = s.search_batch(q, quota=50000, limit=100)
    np.testing.assert_array_equal(a["ids"], c["ids"])
    np.testing.assert_array_equal(a["dists"].view(np.uint64), c["dists"].view(np.uint64))
    s.set_scan_mode(exact_only=True)
    b = s.search_batch(q, quota=50000, limit=100)
    np.testing.assert_array_equal(a["ids"], b["ids"])
    np.testing.assert_array_equal(a["dists"].view(np.uint64), b["dists"].view(np.uint64))
    for mode in (3, 4, 5):  # 16-bit fixed-point kernel: the crowd of equal sums goes through the exact compaction (4: via the two-pass
        s.set_scan_mode(mode=mode)  # form's crowd fall-through; 5: the sampled form's list overflow -> two-pass -> streaming)
        for qq in (q, np.concatenate([q, X[1:8]])):  # alone in its slot, and sharing slots with other queries
            d = s.search_batch(qq, quota=50000, limit=100)
            np.testing.assert_array_equal(a["ids"][0], d["ids"][0])
            np.testing.assert_array_equal(a["dists"][0].view(np.uint64), d["dists"][0].view(np.uint64))
    # item 0 (id 10) and the duplicates (ids 3010...) share the best distance; insertion order within the cell
    best = a["dists"][0, 0]
    tied = a["ids"][0][a["dists"][0] == best]
    assert tied[0] == 10 and (np.diff(tied) > 0).all() and len(tied) == 100


def test_device_selftest_of_wave_primitives():
    """DPP / permlane lane exchanges == __shfl_xor, and the in-register bitonic sort sorts."""
    import ctypes
    from columbiaimagesearch_amd import _lib
    n = ctypes.c_int(-1)
    _lib.check(_lib.lib().cis_selftest(ctypes.byref(n)))
    assert n.value == 0


def test_exhaustive_quota_splits_batches_and_matches_oracle():
    """quota = N (every query visits every cell): the workspace budget forces the batch to be split and
    re-planned; results still equal the oracle's."""
    from oracle import lopq_oracle as O
    z, X, Q = load_golden("c2")
    m = hip_model(z)
    s = _build_searcher("c2", z, X, m)
    rs = np.random.RandomState(5)
    Qb = np.concatenate([Q, X[rs.choice(len(X), 4096 - len(Q), replace=False)]])  # 4096 queries x 256 cells
    quota = len(z["coarse"]) + 1  # can never be filled: every one of the V*V cells is visited
    os.environ["CIS_WORKSPACE_GB"] = "2"  # the default budget (24 GB) would take this batch whole
    try:
        r = s.search_batch(Qb, quota=quota, limit=100)
        r2 = s.search_batch(Qb[:3000], quota=quota, limit=100)  # starts from the remembered sub-batch size
    finally:
        del os.environ["CIS_WORKSPACE_GB"]
    assert (r["visited"] == m.V * m.V).all() and (r["n_found"] == 100).all()
    np.testing.assert_array_equal(r2["ids"], r["ids"][:3000])
    whole = s.search_batch(Qb[:512], quota=quota, limit=100)
    np.testing.assert_array_equal(whole["ids"], r["ids"][:512])
    om = O.OracleModel.from_npz(z)
    oi = O.OracleCSRIndex(om, z["coarse"], z["fine"])
    for qi in list(range(8)) + [2000, 4095]:
        ids, dists, visited = oi.search(Qb[qi], quota=quota, limit=100)
        assert visited == m.V * m.V
        np.testing.assert_array_equal(r["ids"][qi], ids)
        np.testing.assert_allclose(r["dists"][qi], dists, rtol=1e-9)


@pytest.mark.parametrize("name", ["tiny", "c1", "c2"])
def test_multisequence_and_predict_cluster_functions(name):
    """module-level multisequence(x, centroids) and utils.predict_cluster(x, centroids), the reference's names."""
    from columbiaimagesearch_amd.lopq import multisequence
    from columbiaimagesearch_amd.lopq.search import multisequence_batch
    from columbiaimagesearch_amd.lopq.utils import predict_cluster
    from oracle import lopq_oracle as O
    z, X, Q = load_golden(name)
    om = O.OracleModel.from_npz(z)
    m = hip_model(z)
    Qx = np.concatenate([Q, X[z["sel"][:4]]]) if name == "tiny" else Q
    Xq = O.apply_pca(om, Qx) if om.has_pca else Qx
    nq, nc = z["multiseq_cells"].shape[:2]
    cells, dists = multisequence_batch(Xq[:nq], m.Cs, max_cells=nc)
    np.testing.assert_array_equal(cells, z["multiseq_cells"][:, :cells.shape[1]])
    np.testing.assert_array_equal(dists, z["multiseq_dists"][:, :cells.shape[1]].astype(dists.dtype))
    gen = multisequence(Xq[0], m.Cs)
    total = m.V * m.V
    got = [next(gen) for _ in range(min(total, 70))]  # crosses the first 64-cell prefix
    assert [c for _, c in got[:nc]] == [tuple(c) for c in z["multiseq_cells"][0][:min(nc, len(got))].tolist()][:len(got[:nc])]
    if total <= 70:
        assert len(set(c for _, c in got)) == total
    # predict_cluster against coarse centroids and against a fine sub-quantizer
    h = m.Cs[0].shape[1]
    ids = predict_cluster(np.ascontiguousarray(Xq[:, :h]), m.Cs[0])
    np.testing.assert_array_equal(ids, O.predict_cluster(np.ascontiguousarray(Xq[:, :h]), om.Cs[0]))
    one = predict_cluster(np.ascontiguousarray(Xq[0, :h]), m.Cs[0])
    assert one == ids[0] and one.dtype == np.uint8


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_exact_rerank_matches_reference_semantics(dtype):
    """True-L2 re-ranking with resident features == the restated searcher_lopqhbase.py:864-912 loop (missing features
    keep the ADC distance, near-duplicate threshold, max_returned on the pre-sort index)."""
    import torch
    from oracle import lopq_oracle as O
    from columbiaimagesearch_amd.rerank import ResidentFeatures
    rs = np.random.RandomState(3)
    n, D, nq, L = 500, 96, 7, 20
    feats = rs.randn(n, D).astype(dtype)
    feats /= np.linalg.norm(feats, axis=1, keepdims=True)
    Q = (feats[rs.randint(0, n, nq)] + 0.05 * rs.randn(nq, D)).astype(dtype)
    ids = np.stack([rs.choice(n + 40, L, replace=False) for _ in range(nq)]).astype(np.int64)  # some ids have no feature
    ids[2, 15:] = -1
    adc = np.sort(rs.rand(nq, L), axis=1)
    adc[2, 15:] = np.nan
    rf = ResidentFeatures(torch.as_tensor(feats).cuda().contiguous())
    for kw in [dict(rerank_nb=12), dict(rerank_nb=20, max_returned=8), dict(rerank_nb=20, near_dup_th=1.0)]:
        got = rf.rerank(torch.as_tensor(Q).cuda().contiguous(), ids, adc, **kw)
        for qi in range(nq):
            res = [(int(ids[qi, i]), float(adc[qi, i])) for i in range(L) if ids[qi, i] >= 0]
            fb = {int(i): feats[i] for i in ids[qi] if 0 <= i < n}
            eids, ed = O.rerank(Q[qi], fb, res, kw["rerank_nb"], kw.get("max_returned"), kw.get("near_dup_th"))
            gids, gd = got[qi]
            assert [int(i) for i in gids] == [int(i) for i in eids]
            np.testing.assert_allclose(gd, np.asarray(ed, dtype=np.float64), rtol=2e-6 if dtype == np.float32 else 1e-13)


@pytest.mark.gpu
def test_lmdb_order_searcher_matches_key_order_restatement():
    """LOPQSearcherLMDB semantics on the GPU index: byte order of str(id) inside a cell decides ties, a re-added id
    replaces its code (last write wins), ids come back through id_lambda."""
    from oracle import lopq_oracle as O
    from columbiaimagesearch_amd.lopq import LOPQSearcherLMDB
    z, X, Q = load_golden("c2")
    m = hip_model(z)
    om = O.OracleModel.from_npz(z)
    n = 3000
    coarse, fine = z["coarse"][:n], z["fine"][:n]
    codes = [((int(c[0]), int(c[1])), tuple(int(v) for v in f)) for c, f in zip(coarse, fine)]
    ids = [(i * 7919) % 10007 for i in range(n)]  # distinct, not monotone: "10" sorts before "9" as bytes
    s = LOPQSearcherLMDB(m)
    o = O.OracleKeyOrderIndex(om)
    s.add_codes(codes, ids); o.add_codes(codes, ids)
    # duplicates of one code in one cell (ties) and an overwrite of an existing id with another code
    dup = [codes[0]] * 12
    dup_ids = [5, 40, 300, 31, 299, 1000003, 20, 2, 100, 1, 11, 3]
    s.add_codes(dup, dup_ids); o.add_codes(dup, dup_ids)
    s.add_codes([codes[1]], [ids[0]]); o.add_codes([codes[1]], [ids[0]])
    assert s.get_nb_indexed() == o.nb_indexed
    cell = codes[0][0]
    assert [(i, tuple(int(v) for v in c[1])) for i, c in s.get_cell(cell)] == [(i, tuple(c[1])) for i, c in o.get_cell(cell)]
    for qi in range(8):
        for quota, limit in [(10, 10), (400, 50)]:
            got, vis = s.search(Q[qi], quota=quota, limit=limit, with_dists=True)
            res, evis = o.search(Q[qi], quota=quota, limit=limit)
            assert vis == evis and [r.id for r in got] == [r[0] for r in res]
            np.testing.assert_allclose([r.dist for r in got], [r[2] for r in res], rtol=1e-9)


@pytest.mark.gpu
def test_large_limit_sorted_path_matches_oracle():
    """limit above the LDS top-k capacity (here limit=None => limit=quota=5000, search.py:213-214): every candidate is
    scored exactly and a stable segmented sort ranks them; ties (duplicates) keep retrieval order."""
    from oracle import lopq_oracle as O
    z, X, Q = load_golden("c2")
    m = hip_model(z)
    s = _build_searcher("c2", z, X, m)
    dup_c, dup_f = np.repeat(z["coarse"][:1], 40, 0), np.repeat(z["fine"][:1], 40, 0)
    s.add_codes_array(dup_c, dup_f, ids=np.arange(900000, 900040), dedup=False)
    om = O.OracleModel.from_npz(z)
    oi = O.OracleCSRIndex(om, np.concatenate([z["coarse"], dup_c]), np.concatenate([z["fine"], dup_f]),
                          ids=np.concatenate([np.arange(len(z["coarse"])), np.arange(900000, 900040)]))
    nq = 6
    Qx = np.concatenate([Q[:nq - 1], X[:1]])  # the last query sits on the duplicated vector
    for quota, limit in [(5000, None), (20000, 4000)]:
        r = s.search_batch(Qx, quota=quota, limit=limit)
        L = quota if limit is None else limit
        assert r["ids"].shape == (nq, L)
        for qi in range(nq):
            ids, dists, visited = oi.search(Qx[qi], quota=quota, limit=limit)
            n = len(ids)
            assert r["n_found"][qi] == n and r["visited"][qi] == visited
            np.testing.assert_array_equal(r["ids"][qi, :n], ids)
            np.testing.assert_allclose(r["dists"][qi, :n], dists, rtol=1e-9)
            assert (r["ids"][qi, n:] == -1).all()


@pytest.mark.gpu
def test_select_path_matches_oracle_and_exact_kernel():
    """limit above the float32-prefilter kernel's 440: every candidate is scored exactly, a per-query radix select finds
    the cut of the stable ranking, the selected pairs are ranked in LDS (limit <= 3072) or by the segmented sort (above).
    Covers: cut inside a crowd of > 1024 exact ties (retrieval order decides), cut inside a small tie group, segments
    shorter than the limit, both select modes, and the full-sort fallback (limit of the order of the candidate count)."""
    from oracle import lopq_oracle as O
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    z, X, Q = load_golden("c2")
    m = hip_model(z)
    coarse, fine = z["coarse"], z["fine"]
    n_dup = 3000
    dc, df = np.repeat(coarse[:1], n_dup, 0), np.repeat(fine[:1], n_dup, 0)
    sc, sf = np.repeat(coarse[1:2], 7, 0), np.repeat(fine[1:2], 7, 0)
    C = np.concatenate([coarse[:20000], dc, coarse[20000:], sc])
    F = np.concatenate([fine[:20000], df, fine[20000:], sf])
    ids = np.arange(len(C), dtype=np.int64) * 3 + 1
    s = LOPQSearcherHIP(m)
    s.add_codes_array(C, F, ids, dedup=False)
    oi = O.OracleCSRIndex(O.OracleModel.from_npz(z), C, F, ids=ids)
    Qx = np.concatenate([Q[:5], X[:2]])  # the last two queries sit on the duplicated vectors
    nq = len(Qx)
    cases = [(10000, 441), (10000, 1000), (10000, 3072), (40000, 2500), (200, 3000), (60000, 3073), (100000, 4000),
             (5000, None)]
    for quota, limit in cases:
        r = s.search_batch(Qx, quota=quota, limit=limit)
        L = quota if limit is None else limit
        assert r["ids"].shape == (nq, L)
        for qi in range(nq):
            oid, od, visited = oi.search(Qx[qi], quota=quota, limit=limit)
            n = len(oid)
            assert r["n_found"][qi] == n and r["visited"][qi] == visited, (quota, limit, qi)
            np.testing.assert_array_equal(r["ids"][qi, :n], oid, err_msg=str((quota, limit, qi)))
            np.testing.assert_allclose(r["dists"][qi, :n], od, rtol=1e-9)
            assert (r["ids"][qi, n:] == -1).all()
    # same bits as the exact scan kernel's LDS top-k (the path these limits took before)
    a = s.search_batch(Qx, quota=10000, limit=700)
    s.set_scan_mode(exact_only=True)
    b = s.search_batch(Qx, quota=10000, limit=700)
    s.set_scan_mode(exact_only=False)
    np.testing.assert_array_equal(a["ids"], b["ids"])
    np.testing.assert_array_equal(a["dists"].view(np.uint64), b["dists"].view(np.uint64))
    np.testing.assert_array_equal(a["visited"], b["visited"])


def _random_model(V, M, K, D, seed, dtype=np.float32):
    """Untrained model of a given shape (production configs use V = 256 ... 4096, conf/*.json): centroids drawn from the
    data distribution, random orthogonal local rotations."""
    from oracle import lopq_oracle as O
    from columbiaimagesearch_amd.lopq import LOPQModel
    rs = np.random.RandomState(seed)
    h, w, nf = D // 2, D // M, M // 2
    Cs = [rs.randn(V, h).astype(dtype) for _ in range(2)]
    Rs = [np.stack([np.linalg.qr(rs.randn(h, h))[0] for _ in range(V)]) for _ in range(2)]
    mus = [rs.randn(V, h) * 0.05 for _ in range(2)]
    subs = [[rs.randn(K, w) * 0.6 for _ in range(nf)] for _ in range(2)]
    return LOPQModel(parameters=(tuple(Cs), tuple(Rs), tuple(mus), tuple(subs))), O.OracleModel(Cs, Rs, mus, subs)


@pytest.mark.gpu
@pytest.mark.parametrize("V,M,K,D", [(300, 8, 64, 32), (1024, 4, 256, 16), (4096, 4, 16, 16), (300, 8, 256, 128), (300, 4, 256, 128), (300, 16, 64, 128)])
def test_wide_coarse_vocabulary_matches_oracle(V, M, K, D):
    """V > 256 (uint16 coarse codes, V*V up to a million cells, mostly empty): encode and search parity.  The 128-d shapes put
    w = 16 / 32 / 8 (sub-quantizer phases of 4 / 2 / 8 in k_adc_direct) and h = 64 (one full stage of the coarse prefilter) under test."""
    from oracle import lopq_oracle as O
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    m, om = _random_model(V, M, K, D, seed=V)
    rs = np.random.RandomState(1)
    X = rs.randn(6000, D).astype(np.float32)
    Q = rs.randn(10, D).astype(np.float32)
    coarse, fine = m.predict_batch(X)
    oc, of = O.predict_batch(om, X) if hasattr(O, "predict_batch") else (None, None)
    if oc is None:
        codes = [O.predict(om, x) for x in X[:500]]
        oc = np.array([c[0] for c in codes]); of = np.array([c[1] for c in codes])
        np.testing.assert_array_equal(coarse[:500], oc); np.testing.assert_array_equal(fine[:500], of)
    else:
        np.testing.assert_array_equal(coarse, oc); np.testing.assert_array_equal(fine, of)
    assert coarse.dtype == np.uint16 and int(coarse.max()) >= 256
    s = LOPQSearcherHIP(m)
    s.add_codes_array(coarse, fine)
    oi = O.OracleCSRIndex(om, coarse, fine)
    # with V = 4096 nearly all of the 16.7 M cells are empty: a quota of 20 already walks ~10^4 cells in the oracle's heap
    settings = [(5, 5), (200, 50)] if V <= 1024 else [(5, 5), (20, 20)]
    Q = Q if V <= 1024 else Q[:4]
    for quota, limit in settings:
        r = s.search_batch(Q, quota=quota, limit=limit)
        for qi in range(len(Q)):
            ids, dists, visited = oi.search(Q[qi], quota=quota, limit=limit)
            n = len(ids)
            assert r["n_found"][qi] == n and r["visited"][qi] == visited
            np.testing.assert_array_equal(r["ids"][qi, :n], ids)
            np.testing.assert_allclose(r["dists"][qi, :n], dists, rtol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("V,M,K,D", [(300, 8, 256, 128), (1024, 4, 256, 32), (1500, 4, 16, 32), (300, 4, 64, 24)])
def test_alternative_kernels_agree_bit_for_bit(monkeypatch, V, M, K, D):
    """Round 4 moved several kernels to other forms that must not change a bit: the tables' projection (float64 matrix cores /
    vector unit grouped / one table per block -- all ONE chain of fused multiply-adds per output), the encode's tile projection
    (matrix cores / vector unit), the coarse rank sort (registers / LDS), the band sizes of the parallel plan (with / without the
    hint of the previous launch).  Shapes: h = 64 (16 MFMA steps in flight), h = 16 (4), h = 12 (no matrix-core form)."""
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    m, _ = _random_model(V, M, K, D, seed=V + D)
    rs = np.random.RandomState(2)
    X = rs.randn(20000, D).astype(np.float32)
    Q = rs.randn(300, D).astype(np.float32)
    coarse, fine = m.predict_batch(X)
    monkeypatch.setenv("CIS_PROJECT_VALU", "1")
    c2, f2 = m.predict_batch(X)
    monkeypatch.delenv("CIS_PROJECT_VALU")
    np.testing.assert_array_equal(coarse, c2)
    np.testing.assert_array_equal(fine, f2)
    s = LOPQSearcherHIP(m)
    s.add_codes_array(coarse, fine)
    try:
        for quota, limit in ((40, 10), (600, 100)):
            want = s.search_batch(Q, quota=quota, limit=limit)
            want = s.search_batch(Q, quota=quota, limit=limit)  # (the second call runs with the plan's hint of the first)
            for env in ("CIS_TABLES_VALU", "CIS_TABLES_UNGROUPED", "CIS_RANK_SORT_LDS", "CIS_NO_PLAN_HINT", "CIS_NO_PAR_PLAN"):
                monkeypatch.setenv(env, "1")
                got = s.search_batch(Q, quota=quota, limit=limit)
                monkeypatch.delenv(env)
                for k in ("ids", "n_found", "visited"):
                    np.testing.assert_array_equal(got[k], want[k], err_msg="%s %s" % (env, k))
                np.testing.assert_array_equal(np.asarray(got["dists"]).view(np.uint64), np.asarray(want["dists"]).view(np.uint64), err_msg=env)
    finally:
        s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("V,h,dtype", [(300, 64, np.float32), (1000, 30, np.float64), (2048, 64, np.float32), (512, 128, np.float64),
                                       (640, 128, np.float32), (257, 7, np.float32)])
def test_coarse_prefilter_matches_numpy_argmin(V, h, dtype):
    """Coarse ids with many clusters: the matrix-core prefilter + exact re-check (k_coarse_mfma) against numpy's
    ((x - C)**2).sum(axis=1).argmin() (lopq/lopq/utils.py:33-53) and against the all-pairs exact kernels, on inputs built to
    tie: duplicated centroids (the first index must win), vectors ON centroids, midpoints of centroid pairs, zero vectors,
    vectors far outside the centroids' range, a ragged row count."""
    from oracle import lopq_oracle as O
    from columbiaimagesearch_amd.lopq import LOPQModel
    rs = np.random.RandomState(V + h)
    D = 2 * h
    Cs = [rs.randn(V, h).astype(dtype) * (0.3 + s) for s in range(2)]
    for C in Cs:
        C[V // 2] = C[3]            # exact duplicates: numpy returns the first
        C[V - 1] = C[V // 3]
    M = 2
    Rs = tuple(np.broadcast_to(np.eye(h), (V, h, h)).copy() for _ in range(2))
    mus = tuple(np.zeros((V, h)) for _ in range(2))
    subs = tuple([rs.randn(4, h)] for _ in range(2))
    m = LOPQModel(parameters=(tuple(Cs), Rs, mus, subs))
    om = O.OracleModel(Cs, list(Rs), list(mus), [list(subs[0]), list(subs[1])])
    n = 6001
    X = np.concatenate([rs.randn(n, h) * (0.3 + s) for s in range(2)], axis=1)
    pick = rs.randint(0, V, size=(n, 2))
    for s in range(2):
        sl = slice(s * h, (s + 1) * h)
        X[0:1500, sl] = Cs[s][pick[0:1500, s]]                                      # on a centroid (some on the duplicated ones)
        X[1500:3000, sl] = (Cs[s][pick[1500:3000, s]].astype(np.float64) + Cs[s][pick[1500:3000, 1 - s]]) / 2  # midpoints
        X[3000:3050, sl] = 0.0
        X[3050:3100, sl] *= 1e3
        X[3100:3150, sl] *= 1e-6
    X[:20, :h] = Cs[0][3]
    X = X.astype(dtype)
    want = O.predict_coarse(om, X)
    got = {}
    for mode in ("1", "0"):
        os.environ["CIS_COARSE"] = mode
        try:
            got[mode] = m.predict_coarse(X)
        finally:
            del os.environ["CIS_COARSE"]
    np.testing.assert_array_equal(got["0"], want)
    np.testing.assert_array_equal(got["1"], want)
    assert (want[:20, 0] == 3).all()


@pytest.mark.gpu
def test_coarse_prefilter_overflow_falls_back_to_exact_kernels():
    """Hundreds of identical centroids put every one of them on a row's candidate list: the list overflows, the kernel raises
    its flag and the predicated exact kernels redo the pass -- the answer is still numpy's (first index of the tie)."""
    from oracle import lopq_oracle as O
    from columbiaimagesearch_amd.lopq import LOPQModel
    rs = np.random.RandomState(5)
    V, h = 768, 16
    Cs = [rs.randn(V, h).astype(np.float32) for _ in range(2)]
    Cs[0][100:700] = Cs[0][100]
    Rs = tuple(np.broadcast_to(np.eye(h), (V, h, h)).copy() for _ in range(2))
    mus = tuple(np.zeros((V, h)) for _ in range(2))
    subs = tuple([rs.randn(4, h)] for _ in range(2))
    m = LOPQModel(parameters=(tuple(Cs), Rs, mus, subs))
    om = O.OracleModel(Cs, list(Rs), list(mus), [list(subs[0]), list(subs[1])])
    X = rs.randn(3000, 2 * h).astype(np.float32)
    X[:1000, :h] = Cs[0][100] + rs.randn(1000, h).astype(np.float32) * 1e-3
    os.environ["CIS_COARSE"] = "1"
    try:
        got = m.predict_coarse(X)
    finally:
        del os.environ["CIS_COARSE"]
    want = O.predict_coarse(om, X)
    np.testing.assert_array_equal(got, want)
    assert (want[:1000, 0] == 100).all()


@pytest.mark.gpu
def test_randomised_parity_fuzz():
    """Random model shapes / duplicate-heavy data / quota and limit over all three ranking paths (tests/tools/fuzz_parity.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(__file__), "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.VS = [2, 4, 16, 40]
    assert mod.run(60, 11) == 0
    # regression: the two-pass fixed-point scan must collect up to M + 1 units past its histogram edge (a sum brackets the exact
    # distance only to within M units); these two cases lost the candidate of rank `limit` before the fix
    os.environ["FUZZ_ONLY"], os.environ["FUZZ_MODE"] = "2,58", "4"
    try:
        mod.VS = [2, 4, 16, 40, 40, 300, 1024]  # the stream that found it
        assert mod.run(60, 24) == 0
    finally:
        del os.environ["FUZZ_ONLY"], os.environ["FUZZ_MODE"]
    mod.VS = [300]  # uint16 coarse ids: matrix-core coarse prefilter, tiny cells (parallel plan, direct ADC); the tool's CLI adds 1024
    assert mod.run(6, 13) == 0
    mod.VS = [2, 4, 16, 40]
    # shapes outside the specialised kernels (M not in {4, 8, 16}, K not a multiple of 4 / not a power of two): the float64
    # scan kernel and the global-table distance kernel of the all-candidates path
    mod.MS, mod.KS = [2, 6, 12, 32], [10, 100, 256]
    assert mod.run(30, 12) == 0


@pytest.mark.gpu
def test_kmeans_hip_distortion_close_to_sklearn():
    """GPU Lloyd iterations (training): k-means is not bit-reproducible, so the bar is the
    distortion -- within 3 % of scikit-learn's full k-means on the same data, and a model fitted with the GPU backend
    encodes with a quantisation error close to the scikit-learn-fitted one."""
    from sklearn.cluster import KMeans
    from columbiaimagesearch_amd.lopq import train as T
    import golden_inputs as gi
    X = gi.descriptor_like(30000, 32, 6, np.float64)
    for k in (16, 200):
        C, inertia = T.kmeans_hip(X, k, iters=25, n_init=2, random_state=3)
        ref = KMeans(n_clusters=k, n_init=2, max_iter=25, random_state=3).fit(X)
        d = ((X[:, None, :] - C[None]) ** 2).sum(-1).min(1).sum() if k <= 16 else None
        if d is not None:
            assert abs(d - inertia) <= 1e-3 * d  # the reported inertia is that of the returned centroids (float32 sums)
        assert inertia <= 1.03 * ref.inertia_, (k, inertia, ref.inertia_)
    from columbiaimagesearch_amd.lopq import LOPQModel
    errs = {}
    for backend in ("sklearn", "hip"):
        T.KMEANS_BACKEND = backend
        try:
            m = LOPQModel(V=8, M=4, subquantizer_clusters=64)
            m.fit(X, n_init=1, random_state=5)
        finally:
            T.KMEANS_BACKEND = "sklearn"
        coarse, fine = m.predict_batch(X[:5000])
        rec = np.stack([m.reconstruct((tuple(c), tuple(f))) for c, f in zip(coarse[:500], fine[:500])])
        errs[backend] = float(((X[:500] - rec) ** 2).sum(1).mean())
    assert errs["hip"] <= 1.1 * errs["sklearn"], errs


@pytest.mark.gpu
def test_training_accumulations_on_gpu_match_host():
    """SURVEY.md section 8f row 3 beyond k-means: the covariance accumulators (lopq/lopq/model.py:142-155, :263-267) and the
    per-cluster projection (:209-234) as float64 GPU products -- against numpy (summation order is the only difference), and a
    model whose rotations / PCA were fitted through them against the host-fitted one (same k-means seeds => same parameters up
    to that rounding, and the same codes on almost every vector)."""
    from columbiaimagesearch_amd.lopq import LOPQModelPCA
    from columbiaimagesearch_amd.lopq import train as T
    import golden_inputs as gi
    rs = np.random.RandomState(3)
    for n, d, groups in ((5000, 24, 1), (7000, 70, 9), (900, 130, 5)):
        X = rs.randn(n, d) * (1.0 + np.arange(d)) ** -0.3
        assign = rs.randint(0, groups, n) if groups > 1 else None
        if groups > 1:
            assign[assign == 2] = 3  # an empty group
        G, S = T.gram_hip(X, assign, groups)
        for g in range(groups):
            r = X if assign is None else X[assign == g]
            np.testing.assert_allclose(G[g], r.T.dot(r), rtol=1e-12, atol=1e-10)
            np.testing.assert_allclose(S[g], r.sum(axis=0), rtol=1e-12, atol=1e-10)
        if groups > 1:
            R = rs.randn(groups, d, d)
            mu = rs.randn(groups, d)
            T.ACCUM_BACKEND = "hip"
            try:
                got = T.project_to_local(X, assign, R, mu)
            finally:
                T.ACCUM_BACKEND = "host"
            np.testing.assert_allclose(got, T.project_to_local(X, assign, R, mu), rtol=1e-11, atol=1e-11)
    X = gi.descriptor_like(20000, 48, 8, np.float64)
    models = {}
    for backend in ("host", "hip"):
        T.ACCUM_BACKEND = backend
        try:
            m = LOPQModelPCA(V=4, M=4, subquantizer_clusters=32, renorm=True)
            m.fit(X, pca_dims=32, n_init=1, random_state=9)
        finally:
            T.ACCUM_BACKEND = "host"
        models[backend] = m
    a, b = models["host"], models["hip"]
    np.testing.assert_allclose(np.abs(a.pca_P), np.abs(b.pca_P), atol=1e-6)  # eigenvectors up to sign
    ca, fa = a.predict_batch(X[:4000])
    cb, fb = b.predict_batch(X[:4000])
    assert (ca != cb).any(axis=1).mean() < 0.01 and (fa != fb).any(axis=1).mean() < 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("cdtype", [np.float32, np.float64])
def test_multisequence_plan_with_thousands_of_cells_and_tied_sums(cdtype):
    """The plan of indexes with many coarse clusters (k_plan_par: banded selection + sort instead of one frontier step per
    cell) against the oracle's frontier walk, on a lattice model built to make the rank-pair sums d0[i] + d1[j] TIE by
    the hundreds (integer squares: 1 + 49 == 25 + 25): the heap's order is the (sum, i, j) order, ties included.
    Thousands of visited cells (several bands), every cell of the index (quota above its size), tiny cells throughout."""
    from oracle import lopq_oracle as O
    from columbiaimagesearch_amd.lopq import LOPQModel, LOPQSearcherHIP
    V, K = 128, 4
    c = np.arange(1, V + 1, dtype=cdtype).reshape(V, 1)   # (float64 coarse centroids: 64-bit sums and sort keys in k_plan_par)
    Cs = (c.copy(), c.copy())
    Rs = tuple(np.ones((V, 1, 1)) for _ in range(2))
    mus = tuple(np.zeros((V, 1)) for _ in range(2))
    subs = tuple([np.array([[-0.3], [-0.1], [0.1], [0.3]])] for _ in range(2))
    m = LOPQModel(parameters=(Cs, Rs, mus, subs))
    om = O.OracleModel(list(Cs), list(Rs), list(mus), [list(subs[0]), list(subs[1])])
    rs = np.random.RandomState(12)
    n = 30000
    cells = rs.randint(1, V + 1, size=(n, 2))
    X = (cells + rs.uniform(-0.4, 0.4, size=(n, 2))).astype(np.float32)
    coarse, fine = m.predict_batch(X)
    np.testing.assert_array_equal(coarse, cells - 1)
    s = LOPQSearcherHIP(m)
    s.add_codes_array(coarse, fine)
    oi = O.OracleCSRIndex(om, coarse, fine)
    Q = np.array([[0.0, 0.0], [-3.0, 200.0], [140.0, -1.0], [0.0, 131.0]], dtype=np.float32)  # distinct d0 and d1, tied sums
    sums = np.add.outer((Q[0, 0] - c[:, 0]) ** 2, (Q[0, 1] - c[:, 0]) ** 2)
    assert len(np.unique(sums)) < sums.size // 2  # the construction does tie
    for quota, limit in ((700, 30), (6000, 50), (40000, 60)):
        r = s.search_batch(Q, quota=quota, limit=limit)
        for qi in range(len(Q)):
            ids, dists, visited = oi.search(Q[qi], quota=quota, limit=limit)
            k = len(ids)
            assert int(r["visited"][qi]) == visited and int(r["n_found"][qi]) == k, (quota, qi, int(r["visited"][qi]), visited)
            np.testing.assert_array_equal(r["ids"][qi, :k], ids)
            np.testing.assert_allclose(r["dists"][qi, :k], dists, rtol=1e-9, atol=1e-12)
    # the visit ORDER itself (ranks are part of the ranking key): multisequence cells, tie groups included
    from columbiaimagesearch_amd.lopq.search import multisequence_batch
    got_cells, _ = multisequence_batch(Q, Cs, max_cells=3000)
    for qi in range(len(Q)):
        want = [cell for _, cell in zip(range(3000), (cc for _, cc in O.multisequence(om, Q[qi])))]
        np.testing.assert_array_equal(got_cells[qi], np.array(want))


@pytest.mark.gpu
@pytest.mark.parametrize("swap", [False, True])
def test_multisequence_plan_past_the_staged_ranks(swap, monkeypatch):
    """k_plan_par keeps the first 1024 ranks of either sorted list in LDS and reads deeper ranks in place; it lists only the visited cells
    that hold anything and gives half tables to the ranks that have one.  A lattice model with V = 1500 whose one half is 40 times coarser
    than the other (the region under a threshold is a thin ellipse: ~1400 ranks deep on one list after ~20 k cells), few points per cell,
    most cells empty: visited count, ids and distances against the oracle's frontier walk, and -- for a longer walk than the oracle
    finishes in seconds -- against the serial frontier walk of the library (CIS_NO_PAR_PLAN)."""
    from oracle import lopq_oracle as O
    from columbiaimagesearch_amd.lopq import LOPQModel, LOPQSearcherHIP
    V = 1500
    c = np.arange(1, V + 1, dtype=np.float32).reshape(V, 1)
    scale = np.array([40.0, 1.0] if swap else [1.0, 40.0], dtype=np.float32)
    Cs = ((c * scale[0]).copy(), (c * scale[1]).copy())
    Rs = tuple(np.ones((V, 1, 1)) for _ in range(2))
    mus = tuple(np.zeros((V, 1)) for _ in range(2))
    subs = tuple([np.array([[-0.3], [-0.1], [0.1], [0.3]])] for _ in range(2))
    m = LOPQModel(parameters=(Cs, Rs, mus, subs))
    om = O.OracleModel(list(Cs), list(Rs), list(mus), [list(subs[0]), list(subs[1])])
    rs = np.random.RandomState(5)
    n = 20000
    cells = rs.randint(1, V + 1, size=(n, 2))
    X = (cells * scale + rs.uniform(-0.4, 0.4, size=(n, 2))).astype(np.float32)
    coarse, fine = m.predict_batch(X)
    np.testing.assert_array_equal(coarse, cells - 1)
    s = LOPQSearcherHIP(m)
    s.add_codes_array(coarse, fine)
    oi = O.OracleCSRIndex(om, coarse, fine)
    Q = (np.array([[700.3, 30.6], [-200.0, 3.2], [1480.2, 1499.7]], dtype=np.float32) * scale).astype(np.float32)
    if swap:
        Q = Q[:, ::-1].copy() / scale[::-1] * scale
    r = s.search_batch(Q, quota=200, limit=40)
    deep = 0
    for qi in range(len(Q)):
        ids, dists, visited = oi.search(Q[qi], quota=200, limit=40)
        k = len(ids)
        assert int(r["visited"][qi]) == visited and int(r["n_found"][qi]) == k, (qi, int(r["visited"][qi]), visited)
        np.testing.assert_array_equal(r["ids"][qi, :k], ids)
        np.testing.assert_allclose(r["dists"][qi, :k], dists, rtol=1e-9, atol=1e-12)
        deep = max(deep, visited)
    assert deep > 15000
    # a walk of hundreds of thousands of cells: the banded plan against the frontier walk, cell by cell
    want = None
    for no_par in ("1", None):
        if no_par:
            monkeypatch.setenv("CIS_NO_PAR_PLAN", no_par)
        else:
            monkeypatch.delenv("CIS_NO_PAR_PLAN")
        got = s.search_batch(Q, quota=2500, limit=30)
        if want is None:
            want = got
    for k in ("ids", "visited", "n_found"):
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    np.testing.assert_array_equal(got["dists"].view(np.uint64), want["dists"].view(np.uint64))
    assert int(got["visited"].max()) > 200000
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("V,M,D,dup", [(96, 8, 128, 0), (96, 4, 128, 0), (96, 16, 128, 0), (64, 8, 64, 0), (64, 16, 64, 0), (64, 4, 64, 0), (96, 16, 256, 0), (96, 8, 256, 0), (96, 8, 128, 5000)])
def test_tiny_cells_prefilter_equals_exact_kernels(V, M, D, dup, monkeypatch):
    """k_tiny_select (byte-table prefilter + exact keys of the survivors, one workgroup per query) against the exact kernels it
    replaces (k_adc_direct + k_select_topl, CIS_NO_TINY=1) and against the oracle: ids, float64 distance bits, counts, visited --
    on indexes of tiny cells (thousands of cells of a few codes) for every sub-quantizer shape the kernel is built for, with a
    block of identical codes (`dup`: more exact ties at the cut than the survivor list holds, so those queries take the
    flagged route) and with quotas on both sides of the candidate layout's size."""
    import torch
    from oracle import lopq_oracle as O
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    K = 256
    m, om = _random_model(V, M, K, D, seed=V + M)
    rs = np.random.RandomState(M)
    n = 150_000
    X = rs.randn(n, D).astype(np.float32)
    if dup:
        X[1000:1000 + dup] = X[1000]
    coarse, fine = m.predict_batch(X)
    Q = np.concatenate([X[1000:1003] + 1e-3 * rs.randn(3, D).astype(np.float32), rs.randn(61, D).astype(np.float32)])
    s = LOPQSearcherHIP(m)
    s.add_codes_array(coarse, fine, ids=np.arange(n, dtype=np.int64), dedup=False)
    q = torch.as_tensor(Q).cuda()
    monkeypatch.setenv("CIS_TINY_MIN_QUOTA", "0")
    oi = O.OracleCSRIndex(om, coarse, fine)
    for quota, limit in [(4000, 100), (9000, 37), (2500, 300), (20000, 100)]:
        monkeypatch.setenv("CIS_NO_TINY", "1")
        want = {k: v.cpu().numpy() for k, v in s.search_batch_dev(q, quota=quota, limit=limit).items() if hasattr(v, "cpu")}
        monkeypatch.delenv("CIS_NO_TINY")
        got = {k: v.cpu().numpy() for k, v in s.search_batch_dev(q, quota=quota, limit=limit).items() if hasattr(v, "cpu")}
        for k in ("ids", "n_found", "visited"):
            np.testing.assert_array_equal(got[k], want[k], err_msg="quota %d limit %d %s" % (quota, limit, k))
        np.testing.assert_array_equal(got["dists"].view(np.uint64), want["dists"].view(np.uint64))
        for qi in (0, 5):
            ids, dists, visited = oi.search(Q[qi], quota, limit)
            assert visited == got["visited"][qi]
            np.testing.assert_array_equal(np.asarray(ids, np.int64), got["ids"][qi][:len(ids)])
            np.testing.assert_allclose(got["dists"][qi][:len(ids)], np.asarray(dists), rtol=1e-9)
    # two cell shards on this device: each rank's partial search takes the prefilter on ITS candidates (fewer than the quota,
    # some queries with fewer than `limit`), the merged lists equal the single index's
    from columbiaimagesearch_amd.lopq.search import merge_hits_dev
    quota, limit = 9000, 100
    want = s.search_batch_dev(q, quota=quota, limit=limit)
    parts = []
    for r in range(2):
        sh = LOPQSearcherHIP(m, shard=(r, 2))
        sh.add_codes_array(coarse, fine, ids=np.arange(n, dtype=np.int64), dedup=False)
        h, v = sh.search_partial_dev(q, quota=quota, limit=limit)
        np.testing.assert_array_equal(v.cpu().numpy(), want["visited"].cpu().numpy())
        parts.append(h)
        sh.close()
    out = merge_hits_dev(torch.stack(parts).contiguous())
    np.testing.assert_array_equal(out["ids"].cpu().numpy(), want["ids"].cpu().numpy())
    np.testing.assert_array_equal(out["dists"].cpu().numpy().view(np.uint64), want["dists"].cpu().numpy().view(np.uint64))
    s.close()


def test_search_views_share_the_index_and_pipeline_on_two_streams():
    """cis_index_create_view: a view answers exactly like its base (same storage, own workspaces), batches alternated between
    base and view on two streams give the serial results, a view is read-only, and an insert into the base is seen by the view."""
    import torch
    z, X, Q = load_golden("c2")
    m = hip_model(z)
    s = _build_searcher("c2", z, X, m)
    v = s.view()
    q = torch.as_tensor(Q).cuda().contiguous()
    want = {k: t.cpu().numpy() for k, t in s.search_batch_dev(q, quota=2000, limit=50).items()}
    got = {k: t.cpu().numpy() for k, t in v.search_batch_dev(q, quota=2000, limit=50).items()}
    for k in ("ids", "n_found", "visited"):
        np.testing.assert_array_equal(got[k], want[k])
    np.testing.assert_array_equal(got["dists"].view(np.uint64), want["dists"].view(np.uint64))
    # eight batches of different queries, alternating handles and streams, nothing synchronised in between
    batches = [q[i::4].contiguous() for i in range(4)] * 2
    serial = [{k: t.cpu().numpy() for k, t in s.search_batch_dev(b, quota=1000, limit=30).items()} for b in batches]
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for i, b in enumerate(batches):
        if i % 2:
            with torch.cuda.stream(side):
                outs.append(v.search_batch_dev(b, quota=1000, limit=30))
        else:
            outs.append(s.search_batch_dev(b, quota=1000, limit=30))
    torch.cuda.synchronize()
    for o, w in zip(outs, serial):
        for k in ("ids", "n_found", "visited"):
            np.testing.assert_array_equal(o[k].cpu().numpy(), w[k])
        np.testing.assert_array_equal(o["dists"].cpu().numpy().view(np.uint64), w["dists"].view(np.uint64))
    # read-only
    with pytest.raises(ValueError, match="view"):
        v.add_codes_array(z["coarse"][:4], z["fine"][:4], ids=np.arange(10 ** 9, 10 ** 9 + 4))
    # the view follows the base: a new item (a copy of the first query's best hit: same distance, later position) shows up
    best = int(want["ids"][0, 0])
    row = best if best < len(z["coarse"]) else 0  # _build_searcher's ids are the row numbers
    s.add_codes_array(z["coarse"][row:row + 1], z["fine"][row:row + 1], ids=np.array([10 ** 12], dtype=np.int64))
    torch.cuda.synchronize()
    a = s.search_batch_dev(q[:1].contiguous(), quota=2000, limit=50)
    b = v.search_batch_dev(q[:1].contiguous(), quota=2000, limit=50)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(a["ids"].cpu().numpy(), b["ids"].cpu().numpy())
    assert (a["ids"].cpu().numpy() == 10 ** 12).any()
    v.close()


@pytest.mark.gpu
def test_model_twin_encodes_alike_on_its_own_stream():
    """model.twin(): a second device handle on the same parameter arrays; passes through the model and its twin on two streams give
    the codes of the serial passes."""
    import torch
    z, X, Q = load_golden("c2")
    m = hip_model(z)
    t = m.twin()
    assert t is not m and t.Cs is m.Cs and t._handle() != m._handle()
    xa = torch.as_tensor(X[:3000]).cuda().contiguous()
    xb = torch.as_tensor(X[3000:5000]).cuda().contiguous()
    wa, wb = m.predict_batch_dev(xa), m.predict_batch_dev(xb)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(s1):
            ga = m.predict_batch_dev(xa)
        with torch.cuda.stream(s2):
            gb = t.predict_batch_dev(xb)
    torch.cuda.synchronize()
    assert torch.equal(ga[0], wa[0]) and torch.equal(ga[1], wa[1]) and torch.equal(gb[0], wb[0]) and torch.equal(gb[1], wb[1])
    assert (ga[0].cpu().numpy().view(np.uint16) == z["coarse"][:3000]).all() and (ga[1].cpu().numpy() == z["fine"][:3000]).all()
