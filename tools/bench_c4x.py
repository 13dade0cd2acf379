#!/usr/bin/env python3
"""Exploration of the HBM-resident regime: an index whose codes (N x M bytes) exceed the 256 MB Infinity Cache several times.

    python tools/bench_c4x.py [N=200000000] [modes=0,2,3] [nqs=1,2,4,8,16,64]

Builds the C4 model's index over N generated vectors (encode on the GPU, device-side insert), then times
  (a) 8192-query batches at quota 10000 (one ~N/256-candidate cell per query), and
  (b) exhaustive searches (quota = N) of small batches on every requested scan route,
and prints, per line, the time per batch, SURVEY 8(d)'s accounting (candidates x M bytes / time) and the PHYSICAL minimum
(every code byte once per batch: N x M / time) against 8 TB/s.
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
    modes = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,2,3").split(",")]
    nqs = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "1,2,4,8,16,64").split(",")]
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
    t0 = time.perf_counter()
    enc_s = 0.0
    sub = 1 << 20
    for c in range(n_chunks):
        x = bench.gen_chunk(P, c, chunk_n, device)
        co_l, fi_l = [], []
        torch.cuda.synchronize()
        te = time.perf_counter()
        for a in range(0, chunk_n, sub):
            co, fi = model.predict_batch_dev(x[a:a + sub])
            co_l.append(co)
            fi_l.append(fi)
        torch.cuda.synchronize()
        enc_s += time.perf_counter() - te
        ids = torch.arange(c * chunk_n, (c + 1) * chunk_n, dtype=torch.int64, device=device)
        searcher.add_codes_dev(torch.cat(co_l), torch.cat(fi_l), ids, dedup=False)
        del x, co_l, fi_l, ids
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    M = model.M
    print("built %d x %d-byte codes (%.2f GB of codes + %.2f GB of ids) in %.1f s (encode %.1f s = %.1f M vectors/s)"
          % (N, M, N * M / 1e9, N * 8 / 1e9, build_s, enc_s, N / enc_s / 1e6), flush=True)
    x0 = bench.gen_chunk(P, 0, min(chunk_n, 1 << 20), device)
    q_all = bench.make_queries(x0, 0, 8192, device)
    del x0

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts)), min(ts), max(ts)

    # (a) the bench batch on the big index
    for mode in modes:
        try:
            searcher.set_scan_mode(mode=mode)
            med, lo, hi = timed(lambda: searcher.search_batch_dev(q_all, quota=bench.QUOTA, limit=bench.LIMIT), 3)
            ls = searcher.last_stats()
            cand = ls["candidates"]
            print("mode %d  8192 queries quota %d: %.3f ms (min %.3f max %.3f)  %s  %.0f cand/query  accounting %.2f TB/s (%.2f of 8)  %.2f M q/s"
                  % (mode, bench.QUOTA, med, lo, hi, ls["scan_kernel"], cand / 8192.0, cand * M / med / 1e9, cand * M / med / 1e9 / 8.0, 8192 / med / 1e3), flush=True)
        except Exception as e:
            print("mode %d  8192 queries: %r" % (mode, e), flush=True)
    # (b) exhaustive, small batches
    for nq in nqs:
        q = q_all[:nq].contiguous()
        for mode in modes:
            try:
                searcher.set_scan_mode(mode=mode)
                med, lo, hi = timed(lambda: searcher.search_batch_dev(q, quota=N, limit=bench.LIMIT), 3)
                ls = searcher.last_stats()
                cand = ls["candidates"]
                print("mode %d  nq %3d exhaustive: %.3f ms (min %.3f max %.3f)  %s  items %d  accounting %.2f TB/s  physical-minimum %.2f TB/s = %.2f of 8 TB/s  %.0f q/s"
                      % (mode, nq, med, lo, hi, ls["scan_kernel"], ls["items"], cand * M / med / 1e9, N * M / med / 1e9, N * M / med / 1e9 / 8.0, nq / med * 1e3), flush=True)
            except Exception as e:
                print("mode %d  nq %d exhaustive: %r" % (mode, nq, e), flush=True)
    searcher.close()


if __name__ == "__main__":
    main()
