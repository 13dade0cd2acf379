"""The host-pointer search in two halves (include/cis_hip.h: cis_index_search_async / cis_index_search_wait, cis_host_alloc): what the
reference's callers use -- queries and results in host memory (searcher_lopqhbase.py:849-857) -- without a blocking copy per batch."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_async_search_on_pinned_buffers_equals_the_blocking_call():
    from test_lopq_hip_parity import hip_model
    from columbiaimagesearch_amd import _lib
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    z, X, Q = load_golden("c2")
    m = hip_model(z)
    s = LOPQSearcherHIP(m)
    s.add_codes_array(z["coarse"], z["fine"])
    nq = 20   # (the fixture holds 64 queries)
    want = [s.search_batch(Q[i * nq:(i + 1) * nq], quota=1000, limit=50) for i in range(3)]
    lanes = [s, s.view(), s.view()]
    qpin = []
    for i in range(3):
        q = _lib.pinned_empty((nq, Q.shape[1]), Q.dtype)
        q[...] = Q[i * nq:(i + 1) * nq]
        qpin.append(q)
    # three batches in flight, each on its own handle, results in library-made pinned arrays
    outs = [lanes[i].search_batch_async(qpin[i], quota=1000, limit=50) for i in range(3)]
    for i in range(3):
        r = lanes[i].search_wait()
        assert r is outs[i]
        for k in ("ids", "n_found", "visited"):
            assert (r[k] == want[i][k]).all(), (i, k)
        assert np.array_equal(np.nan_to_num(r["dists"]), np.nan_to_num(want[i]["dists"]))
    assert lanes[0].search_wait() is None  # nothing in flight any more
    # pageable arrays work as well (the runtime stages them), and a second call on the same handle waits for the first
    out = {"ids": np.empty((nq, 50), np.int64), "dists": np.empty((nq, 50)), "n_found": np.empty(nq, np.int32), "visited": np.empty(nq, np.int32)}
    s.search_batch_async(Q[:nq], quota=1000, limit=50, out=out)
    out2 = s.search_batch_async(Q[nq:2 * nq], quota=1000, limit=50)
    s.search_wait()
    assert (out["ids"] == want[0]["ids"]).all() and (out2["ids"] == want[1]["ids"]).all()
    with pytest.raises(ValueError):
        s.search_batch_async(Q[:nq], quota=1000, limit=50, out={"ids": np.empty((nq, 49), np.int64), "dists": out["dists"], "n_found": out["n_found"], "visited": out["visited"]})
    # a view whose base is closed first is orphaned, not dangling: searches through it fail cleanly
    v = s.view()
    import gc
    s2 = LOPQSearcherHIP(m)
    s2.add_codes_array(z["coarse"][:1000], z["fine"][:1000])
    v2 = s2.view()
    _lib.lib().cis_index_destroy(s2._ix)   # the library call itself, bypassing the Python close() that closes the views first
    s2._ix = None
    with pytest.raises(ValueError, match="base index was destroyed"):
        v2.search_batch(Q[:2], quota=10, limit=5)
    v2.close()
    for x in lanes[1:] + [v]:
        x.close()
    s.close()
    gc.collect()


def test_copy_outs_started_by_other_handles_calls_land_in_the_right_buffers():
    """cis_index_search_async of one handle puts the finished searches' results of the OTHER handles on the copy stream (round 6: the
    copy-out no longer waits for its owner's cis_index_search_wait).  Three handles in rotation, wait-then-launch as bench.py's host
    leg does, different queries and quotas per step, a pause that lets every search finish before the next launch (so that launch does
    start the others' copy-outs): every result equals the blocking call's; a view closed with a batch in flight and never waited for
    leaves the others working."""
    import time
    from test_lopq_hip_parity import hip_model
    from columbiaimagesearch_amd import _lib
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    z, X, Q = load_golden("c2")
    m = hip_model(z)
    s = LOPQSearcherHIP(m)
    s.add_codes_array(z["coarse"], z["fine"])
    nq = 16
    lanes = [s, s.view(), s.view()]
    steps = [(b, (b * 7) % (len(Q) - nq), (500, 1000, 3000)[b % 3]) for b in range(12)]
    want = {b: s.search_batch(Q[o:o + nq], quota=quota, limit=40) for b, o, quota in steps}
    qpin = [_lib.pinned_empty((nq, Q.shape[1]), Q.dtype) for _ in lanes]
    flying = [None] * len(lanes)

    def check(lane):
        r = lanes[lane].search_wait()
        b = flying[lane]
        if b is None:
            assert r is None
            return
        for k in ("ids", "n_found", "visited"):
            assert (r[k] == want[b][k]).all(), (b, k)
        assert np.array_equal(np.nan_to_num(r["dists"]), np.nan_to_num(want[b]["dists"]))
        flying[lane] = None

    for b, o, quota in steps:
        lane = b % len(lanes)
        check(lane)
        if b % 4 == 3:
            time.sleep(0.05)   # every search in flight has finished: the next launch starts their copy-outs
        qpin[lane][...] = Q[o:o + nq]
        lanes[lane].search_batch_async(qpin[lane], quota=quota, limit=40)
        flying[lane] = b
    check(1)
    lanes[2].close()       # a batch in flight, never waited for
    check(0)
    r = s.search_batch(Q[:nq], quota=1000, limit=40)
    assert (r["ids"] == s.search_batch(Q[:nq], quota=1000, limit=40)["ids"]).all()
    lanes[1].close()
    s.close()


def test_limit_equals_quota_ranks_through_the_own_segmented_sort():
    """limit = None => limit = quota (lopq/lopq/search.py:213-214): with quota = 10000 every one of ~10 k retrieved candidates is returned,
    ranked by the stable sort of :210.  Above 3072 results per query the ranking is the library's own segmented merge sort
    (csrc/lopq_sort.hip: LDS tile sort + merge-path passes; rocPRIM until round 4) -- against the oracle, for segment lengths around
    the 4096-pair tile and across several merge passes, incl. a block of equal distances (stable: retrieval order)."""
    from test_lopq_hip_parity import hip_model
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    from oracle import lopq_oracle as O
    z, X, Q = load_golden("c2")
    m = hip_model(z)
    coarse, fine = z["coarse"].copy(), z["fine"].copy()
    coarse[2000:2300] = coarse[2000]   # 300 identical codes: equal distances in the ranked list
    fine[2000:2300] = fine[2000]
    s = LOPQSearcherHIP(m)
    s.add_codes_array(coarse, fine)
    om = O.OracleModel.from_npz(z)
    oi = O.OracleCSRIndex(om, coarse, fine)
    for quota, limit in ((10000, None), (4000, None), (4096, 4096), (4097, None), (20000, 17000), (50000, None)):
        r = s.search_batch(Q[:6], quota=quota, limit=limit)
        for qi in range(6):
            ids, dists, visited = oi.search(Q[qi], quota=quota, limit=limit)
            k = int(r["n_found"][qi])
            assert k == len(ids) and visited == int(r["visited"][qi]), (quota, limit, qi)
            assert (r["ids"][qi, :k] == ids).all(), (quota, limit, qi)
            assert np.allclose(r["dists"][qi, :k], dists, rtol=1e-9, atol=1e-12)
    s.close()
