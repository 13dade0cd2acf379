"""GPU parity of the HBM-resident index and its device-side insert (csrc/lopq_index.hip) against the oracle's dict index
(oracle/lopq_oracle.py:OracleIndex, a restatement of lopq/lopq/search.py:310-382): first (cell, id) wins, insertion order
inside a cell, counts -- whatever mix of host batches (cis_index_add) and device batches (cis_index_add_dev) built it.
"""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _model(name):
    from test_lopq_hip_parity import hip_model
    z, X, Q = load_golden(name)
    return hip_model(z), z, X, Q


def _oracle_index(z):
    from oracle import lopq_oracle as O
    return O.OracleIndex(O.OracleModel.from_npz(z))


def _assert_same_cells(s, oi, V, cells=None):
    if cells is None:
        cells = [(a, b) for a in range(V) for b in range(V)]
    for cell in cells:
        want = oi.get_cell(cell)
        got = s.get_cell(cell)
        assert [i for i, _ in got] == [i for i, _ in want], cell
        assert [tuple(int(v) for v in c.fine) for _, c in got] == [tuple(int(v) for v in c[1]) for _, c in want], cell


def _random_batches(rs, V, M, K, n_batches, id_range, max_n):
    out = []
    for _ in range(n_batches):
        n = int(rs.randint(0, max_n + 1))
        coarse = rs.randint(0, V, size=(n, 2)).astype(np.uint16)
        # few distinct cells and a small id range: duplicates inside a batch, across batches, same id in different cells
        coarse[:, 0] = coarse[:, 0] % 3
        fine = rs.randint(0, K, size=(n, M)).astype(np.uint8)
        ids = rs.randint(0, id_range, size=n).astype(np.int64)
        out.append((coarse, fine, ids))
    return out


@pytest.mark.parametrize("name,seed", [("tiny", 0), ("c1", 1), ("c2", 2)])
def test_mixed_host_and_device_inserts_match_the_dict_index(name, seed):
    import torch
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    m, z, X, Q = _model(name)
    V, M, K = m.V, m.M, m.subquantizer_clusters
    rs = np.random.RandomState(seed)
    s = LOPQSearcherHIP(m)
    oi = _oracle_index(z)
    for b, (coarse, fine, ids) in enumerate(_random_batches(rs, V, M, K, 9, 400, 700)):
        before = oi.nb_indexed
        oi.add_codes_arrays(coarse, fine, ids.tolist())
        if b % 3 == 0:
            added = s.add_codes_array(coarse, fine, ids)
        else:
            added, bad = s.add_codes_dev(torch.as_tensor(coarse.view(np.int16)).cuda(), torch.as_tensor(fine).cuda(), torch.as_tensor(ids).cuda())
            assert bad == 0
        assert added == oi.nb_indexed - before
        assert s.get_nb_indexed() == oi.nb_indexed
    _assert_same_cells(s, oi, V)
    # the index answers like the oracle's (ties between equal codes resolve by insertion order)
    for qi in range(4):
        want, visited = oi.search(Q[qi], quota=60, limit=25, with_dists=True)
        got, vis = s.search(Q[qi], quota=60, limit=25, with_dists=True)
        assert vis == visited and [r.id for r in got] == [r[0] for r in want]


def test_device_insert_counts_and_skips_bad_items():
    import torch
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    m, z, X, Q = _model("tiny")
    V, M = m.V, m.M
    s = LOPQSearcherHIP(m)
    coarse = np.array([[0, 0], [V, 0], [0, 1], [1, 1], [0, 0]], dtype=np.uint16)   # item 1: coarse code out of range
    fine = np.zeros((5, M), dtype=np.uint8)
    ids = np.array([5, 6, -3, 8, 5], dtype=np.int64)                               # item 2: negative id; item 4: duplicate of 0
    delta = torch.zeros(V * V, dtype=torch.int64, device="cuda")
    added, bad = s.add_codes_dev(torch.as_tensor(coarse.view(np.int16)).cuda(), torch.as_tensor(fine).cuda(), torch.as_tensor(ids).cuda(),
                                 cell_delta=delta)
    assert (added, bad) == (2, 2) and s.get_nb_indexed() == 2
    d = delta.cpu().numpy()
    assert d.sum() == 2 and d[0] == 1 and d[1 * V + 1] == 1
    assert [i for i, _ in s.get_cell((0, 0))] == [5] and [i for i, _ in s.get_cell((1, 1))] == [8]
    # the host entry point rejects the whole call and changes nothing
    with pytest.raises(ValueError):
        s.add_codes_array(coarse, fine, np.abs(ids))
    assert s.get_nb_indexed() == 2
    # an empty device batch is fine
    e = torch.zeros((0, 2), dtype=torch.int16, device="cuda")
    assert s.add_codes_dev(e, torch.zeros((0, M), dtype=torch.uint8, device="cuda"), torch.zeros(0, dtype=torch.int64, device="cuda")) == (0, 0)


def test_dedup_after_plain_adds_and_non_monotone_ids():
    """Ids below the cell's maximum take the lookup of the stored ids; plain (dedup=False) adds keep every item."""
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    m, z, X, Q = _model("tiny")
    M = m.M
    s = LOPQSearcherHIP(m)
    c = np.zeros((6, 2), dtype=np.uint16)
    f = np.arange(6 * M, dtype=np.uint8).reshape(6, M) % 8
    assert s.add_codes_array(c, f, np.array([9, 3, 9, 7, 3, 1]), dedup=False) == 6      # duplicates kept
    assert s.add_codes_array(c[:4], f[:4], np.array([7, 2, 10, 9]), dedup=True) == 2     # 7 and 9 are there already
    assert [i for i, _ in s.get_cell((0, 0))] == [9, 3, 9, 7, 3, 1, 2, 10]
    assert s.get_nb_indexed() == 8


def test_a_cell_run_that_ends_with_a_rejected_item_still_raises_the_cells_maximum():
    """The duplicate lookup skips the walk of a cell for ids at or above the cell's recorded maximum.  An in-place batch whose items of
    one cell END with a rejected one (a duplicate) must still record the larger id it accepted before it -- found in round 5: the fused
    small-batch kernel published a run's maximum from the run's last lane, which a rejected item does not do."""
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    m, z, X, Q = _model("tiny")
    M = m.M
    s = LOPQSearcherHIP(m)
    c = np.zeros((4, 2), dtype=np.uint16)
    f = np.arange(4 * M, dtype=np.uint8).reshape(4, M) % 8
    assert s.add_codes_array(c, f, np.array([50, 10, 20, 30]), dedup=True) == 4           # the bulk load: maximum 50
    assert s.add_codes_array(c[:2], f[:2], np.array([100, 50]), dedup=True) == 1           # in place; the run ends with the duplicate 50
    assert s.add_codes_array(c[:3], f[:3], np.array([100, 70, 100]), dedup=True) == 1      # 100 is stored: only 70 is new
    assert [i for i, _ in s.get_cell((0, 0))] == [50, 10, 20, 30, 100, 70]
    assert s.insert_counters()[0] >= 2                                                     # both small batches went in place


@pytest.mark.parametrize("V", [300, 1024])
def test_device_insert_with_thousands_of_cells(V):
    """V*V = 90 000 / 1 M cells: several radix passes over the cell keys, the thread-per-cell move of tiny cells."""
    import torch
    from oracle import lopq_oracle as O
    from test_lopq_hip_parity import _random_model
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    M, K, D = 4, 64, 16
    m, om = _random_model(V, M, K, D, seed=V)
    rs = np.random.RandomState(V)
    s = LOPQSearcherHIP(m)
    oi = O.OracleIndex(om)
    touched = set()
    for b in range(4):
        n = 5000
        coarse = rs.randint(0, V, size=(n, 2)).astype(np.uint16)
        coarse[: n // 2] = coarse[rs.randint(0, 40, size=n // 2)]  # some crowded cells
        fine = rs.randint(0, K, size=(n, M)).astype(np.uint8)
        ids = rs.randint(0, 3000, size=n).astype(np.int64)
        oi.add_codes_arrays(coarse, fine, ids.tolist())
        touched.update((int(a), int(b_)) for a, b_ in coarse)
        if b % 2:
            s.add_codes_array(coarse, fine, ids)
        else:
            s.add_codes_dev(torch.as_tensor(coarse.view(np.int16)).cuda(), torch.as_tensor(fine).cuda(), torch.as_tensor(ids).cuda())
        assert s.get_nb_indexed() == oi.nb_indexed
    cells = sorted(touched)
    _assert_same_cells(s, oi, V, cells[:300] + cells[-300:])


def test_bulk_insert_order_at_a_million_items():
    """One plain insert of 1M items, then a dedup insert of 300k ids of which half exist: positions inside the cells are the
    arrival order (checked through get_cell on sampled cells and a numpy restatement of the merge)."""
    import torch
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    m, z, X, Q = _model("c2")
    V, M = m.V, m.M
    rs = np.random.RandomState(5)
    n = 1_000_000
    coarse = rs.randint(0, V, size=(n, 2)).astype(np.uint16)
    fine = rs.randint(0, 256, size=(n, M)).astype(np.uint8)
    ids = np.arange(n, dtype=np.int64)
    s = LOPQSearcherHIP(m)
    assert s.add_codes_array(coarse, fine, ids, dedup=False) == n
    n2 = 300_000
    ids2 = np.concatenate([rs.randint(0, n, size=n2 // 2), n + np.arange(n2 - n2 // 2)]).astype(np.int64)
    rs.shuffle(ids2)
    c2 = np.where((ids2 < n)[:, None], coarse[np.minimum(ids2, n - 1)], rs.randint(0, V, size=(n2, 2))).astype(np.uint16)
    f2 = rs.randint(0, 256, size=(n2, M)).astype(np.uint8)
    added, bad = s.add_codes_dev(torch.as_tensor(c2.view(np.int16)).cuda(), torch.as_tensor(f2).cuda(), torch.as_tensor(ids2).cuda())
    # numpy restatement: an id < n sits in its old cell already; new ids are unique except for repeats inside the batch
    cell2 = c2[:, 0].astype(np.int64) * V + c2[:, 1]
    first = np.ones(n2, dtype=bool)
    seen = set()
    for i in range(n2):
        key = (int(cell2[i]), int(ids2[i]))
        if ids2[i] < n or key in seen:
            first[i] = False
        seen.add(key)
    assert added == int(first.sum()) and bad == 0 and s.get_nb_indexed() == n + added
    cell1 = coarse[:, 0].astype(np.int64) * V + coarse[:, 1]
    for c in (0, 17, V * V - 1):
        want_ids = np.concatenate([ids[cell1 == c], ids2[first & (cell2 == c)]])
        want_fine = np.concatenate([fine[cell1 == c], f2[first & (cell2 == c)]])
        got = s.get_cell((c // V, c % V))
        np.testing.assert_array_equal(np.array([i for i, _ in got], dtype=np.int64), want_ids)
        np.testing.assert_array_equal(np.array([code.fine for _, code in got], dtype=np.uint8), want_fine)


def test_l2_normalisation_kernel_matches_numpy():
    """cis_l2_normalize_dev against featsio's feat / np.linalg.norm(feat) (cufacesearch/featurizer/featsio.py:13-22)."""
    import torch
    from columbiaimagesearch_amd.ingest import l2_normalize_dev
    rs = np.random.RandomState(0)
    for dtype, d, tol in [(np.float32, 4096, 2 * 2.0 ** -23), (np.float64, 128, 4 * 2.0 ** -52), (np.float32, 37, 2 * 2.0 ** -23)]:
        x = (rs.rand(50, d) * (rs.rand(50, 1) * 10)).astype(dtype)
        x[7] = 0
        got = l2_normalize_dev(torch.as_tensor(x).cuda().clone()).cpu().numpy()
        want = np.stack([r / np.linalg.norm(r) if r.any() else r for r in x])
        assert got.dtype == dtype and (got[7] == 0).all()
        np.testing.assert_allclose(got, want, rtol=tol, atol=0)


def test_small_batches_are_inserted_in_place_and_equal_the_dict_index(monkeypatch):
    """Round 4: a batch whose accepted items fit the slack behind their cells is written IN PLACE (O(batch) bytes; round 5: up to 1024
    items in three launches -- k_ins_small_sort, k_ins_dedup, k_ins_small_place --:
    the reference appends to a per-cell list, lopq/lopq/search.py:349-364); a batch that does not fit rebuilds the layout and
    renews the slack.  Same cells, same order, same search results as the oracle's dict index on both routes -- ids below and above
    the cells' maxima, duplicates inside a batch and against stored items, a cell that fills up."""
    import torch
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    from oracle import lopq_oracle as O
    m, z, X, Q = _model("c2")
    V, M, K = m.V, m.M, m.subquantizer_clusters
    rs = np.random.RandomState(11)
    s = LOPQSearcherHIP(m)
    oi = _oracle_index(z)
    n0 = 20000
    coarse0, fine0 = z["coarse"][:n0], z["fine"][:n0]
    ids0 = rs.permutation(n0).astype(np.int64) * 3  # not monotone: later ids fall below the cells' maxima
    oi.add_codes_arrays(coarse0, fine0, ids0.tolist())
    assert s.add_codes_array(coarse0, fine0, ids0) == n0
    assert s.insert_counters() == (0, 1)  # the bulk load builds the layout (with slack behind every cell)
    for b in range(30):
        n = int(rs.randint(1, 300))
        pick = rs.randint(0, len(z["coarse"]), size=n)
        coarse, fine = z["coarse"][pick].copy(), z["fine"][pick].copy()
        ids = rs.randint(0, 3 * n0 + 500, size=n).astype(np.int64)  # ~1/3 of them already stored somewhere
        if b % 5 == 4:  # one cell takes a burst that no slack holds: the rebuild route
            coarse[:] = coarse[0]
            ids = np.arange(10 ** 6 + b * 1000, 10 ** 6 + b * 1000 + n, dtype=np.int64)
        before = oi.nb_indexed
        oi.add_codes_arrays(coarse, fine, ids.tolist())
        if b % 2:
            added = s.add_codes_array(coarse, fine, ids)
        else:
            added, bad = s.add_codes_dev(torch.as_tensor(coarse.view(np.int16)).cuda(), torch.as_tensor(fine).cuda(), torch.as_tensor(ids).cuda())
            assert bad == 0
        assert added == oi.nb_indexed - before, b
        assert s.get_nb_indexed() == oi.nb_indexed
    inplace, rebuilt = s.insert_counters()
    assert inplace >= 15 and rebuilt >= 2, (inplace, rebuilt)
    _assert_same_cells(s, oi, V)
    for qi in range(4):
        want, visited = oi.search(Q[qi], quota=2000, limit=50)
        r = s.search_batch(Q[qi:qi + 1], quota=2000, limit=50)
        k = int(r["n_found"][0])
        assert k == len(want) and int(r["visited"][0]) == visited
        np.testing.assert_array_equal(r["ids"][0, :k], np.array([w[0] for w in want], dtype=np.int64))
    # tight packing (CIS_INSERT_SLACK=-1 is read once per process: not switchable here) stays covered by the bulk tests above
