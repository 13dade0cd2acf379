#!/usr/bin/env python3
"""The streaming route of the LIBRARY on the 200 M index (c4x): per nq, the whole exhaustive call and the stream kernel's own launch time
(HIP events of the library's profiling marks), plus the agreement of the route with the batch kernels on the same index.

    python tools/r06_stream_lib.py [N=200000000] [nqs=1,2,3,4,8,16]
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
    nqs = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3,4,8,16").split(",")]
    from columbiaimagesearch_amd import _lib
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    _lib.check(_lib.lib().cis_set_device(0))
    model, z = bench.load_model("c4")
    P = bench.mixture_centers("descriptor", device)
    n_chunks = 80
    N -= N % n_chunks
    chunk_n = N // n_chunks
    searcher = LOPQSearcherHIP(model)
    sub = 1 << 20
    for c in range(n_chunks):
        x = bench.gen_chunk(P, c, chunk_n, device)
        co_l, fi_l = [], []
        for a in range(0, chunk_n, sub):
            co, fi = model.predict_batch_dev(x[a:a + sub])
            co_l.append(co)
            fi_l.append(fi)
        ids = torch.arange(c * chunk_n, (c + 1) * chunk_n, dtype=torch.int64, device=device)
        searcher.add_codes_dev(torch.cat(co_l), torch.cat(fi_l), ids, dedup=False)
        del x, co_l, fi_l, ids
    torch.cuda.synchronize()
    M = model.M
    x0 = bench.gen_chunk(P, 0, min(chunk_n, 1 << 20), device)
    q_all = bench.make_queries(x0, 0, 8192, device)
    del x0

    def timed(fn, reps):
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts)), min(ts), max(ts)

    for nq in nqs:
        q = q_all[:nq].contiguous()
        for _ in range(3):
            searcher.search_batch_dev(q, quota=N, limit=bench.LIMIT)
        torch.cuda.synchronize()
        searcher.set_profiling(True, scan_only=True)
        searcher.read_profile()
        med, lo, hi = timed(lambda: searcher.search_batch_dev(q, quota=N, limit=bench.LIMIT), 9)
        prof = searcher.read_profile()
        searcher.set_profiling(False)
        k_ms = prof["scan_kernel_ms"] / max(prof["scan_launches"], 1)
        ls = searcher.last_stats()
        print("nq %2d exhaustive: call %.3f ms (min %.3f max %.3f)  %s %.4f ms per launch (%d launches) = %.3f of 8 TB/s physical, accounting %.2f  %.0f q/s  served/handed back %s"
              % (nq, med, lo, hi, ls["scan_kernel"], k_ms, prof["scan_launches"], N * M / k_ms / 1e6 / 8000.0, nq * N * M / k_ms / 1e6 / 8000.0, nq / med * 1e3,
                 searcher.stream_counters()), flush=True)
    # agreement with the batch kernels
    for nq in (1, 2, 3, 5):
        q = q_all[:nq].contiguous()
        r_s = searcher.search_batch_dev(q, quota=N, limit=bench.LIMIT)
        searcher.set_scan_mode(mode=5)
        r_b = searcher.search_batch_dev(q, quota=N, limit=bench.LIMIT)
        other = searcher.last_stats()["scan_kernel"]
        searcher.set_scan_mode(mode=0)
        torch.cuda.synchronize()
        ok = bool(torch.equal(r_s["ids"], r_b["ids"]) and torch.equal(r_s["dists"].view(torch.int64), r_b["dists"].view(torch.int64)) and torch.equal(r_s["visited"], r_b["visited"]))
        print("nq %d: route == %s: %s" % (nq, other, ok), flush=True)
    q1 = q_all[:1].contiguous()
    for _ in range(3):
        searcher.search_batch_dev(q1, quota=bench.QUOTA, limit=bench.LIMIT)
    med, lo, hi = timed(lambda: searcher.search_batch_dev(q1, quota=bench.QUOTA, limit=bench.LIMIT), 9)
    print("single query, quota %d: %.3f ms (min %.3f) %s" % (bench.QUOTA, med, lo, searcher.last_stats()["scan_kernel"]), flush=True)
    searcher.close()


if __name__ == "__main__":
    main()
