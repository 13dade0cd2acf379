#!/usr/bin/env python3
"""Latency of ONE query per call (what the reference's callers issue: searcher_lopqhbase.py:849-857) on the C4 index (10 M vectors) at the
API's quota 10000, and of small batches: median of the HIP-event time per call.
    python tools/r06_single_query.py [N=10000000] [nqs=1,2,4,8]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    nqs = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
    from columbiaimagesearch_amd import _lib
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    _lib.check(_lib.lib().cis_set_device(0))
    model, z = bench.load_model("c4")
    P = bench.mixture_centers("descriptor", device)
    n_chunks = 10
    chunk_n = N // n_chunks
    s = LOPQSearcherHIP(model)
    for c in range(n_chunks):
        x = bench.gen_chunk(P, c, chunk_n, device)
        co, fi = model.predict_batch_dev(x)
        s.add_codes_dev(co, fi, torch.arange(c * chunk_n, (c + 1) * chunk_n, dtype=torch.int64, device=device), dedup=False)
    x0 = bench.gen_chunk(P, 0, 1 << 20, device)
    q_all = bench.make_queries(x0, 0, 8192, device)
    for nq in nqs:
        qs = [q_all[i * nq:(i + 1) * nq].contiguous() for i in range(16)]
        for q in qs:
            s.search_batch_dev(q, quota=bench.QUOTA, limit=bench.LIMIT)
        torch.cuda.synchronize()
        ts = []
        for r in range(48):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            s.search_batch_dev(qs[r % 16], quota=bench.QUOTA, limit=bench.LIMIT)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("N %d nq %d quota %d: %.3f ms per call (min %.3f)  %s  candidates %d" % (N, nq, bench.QUOTA, float(np.median(ts)), min(ts), s.last_stats()["scan_kernel"],
                                                                                      s.last_stats()["candidates"]), flush=True)
    s.close()


if __name__ == "__main__":
    main()
