"""Insert of a 256-item batch into a resident index of N synthetic codes (uniform over the V*V cells), N = 10M and 50M: ms per
batch through cis_index_add_dev -- ids above every stored id (the duplicate lookup is answered by the per-cell maximum), and ids
below it (the lookup walks the cell) -- and how many batches were written in place.  usage: python tools/bench_insert.py [N,N,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
dev = torch.device("cuda", 0)
model, z = B.load_model("c4")
V, M = model.V, model.M
Ns = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "10000000,50000000").split(",")]
for N in Ns:
    g = torch.Generator(device=dev); g.manual_seed(1)
    s = LOPQSearcherHIP(model)
    for a in range(0, N, 10_000_000):  # bulk build in chunks of 10M
        n = min(10_000_000, N - a)
        co = torch.randint(0, V, (n, 2), generator=g, device=dev, dtype=torch.int16)
        fi = torch.randint(0, 256, (n, M), generator=g, device=dev, dtype=torch.uint8)
        s.add_codes_dev(co, fi, torch.arange(a, a + n, dtype=torch.int64, device=dev) * 2, dedup=False)  # even ids: odd ones are free below the maximum
    torch.cuda.synchronize()
    c0 = s.insert_counters()
    co = torch.randint(0, V, (256, 2), generator=g, device=dev, dtype=torch.int16)
    fi = torch.randint(0, 256, (256, M), generator=g, device=dev, dtype=torch.uint8)
    res = {}
    for tag, base in (("ids above the stored maximum", 4 * N), ("ids below it (cell walk)", 1)):
        for k in range(3):
            s.add_codes_dev(co, fi, torch.arange(256, dtype=torch.int64, device=dev) * 2 + base + 100000 * k, dedup=True)
        torch.cuda.synchronize(); t = time.perf_counter(); K = 20
        for k in range(K):
            added, bad = s.add_codes_dev(co, fi, torch.arange(256, dtype=torch.int64, device=dev) * 2 + base + 100000 * (k + 3), dedup=True)
            assert added == 256
        torch.cuda.synchronize()
        res[tag] = (time.perf_counter() - t) / K * 1e3
    c1 = s.insert_counters()
    print("N %d resident: 256-item insert %s; in place %d, rebuilds %d of %d batches (bulk build: %d rebuild(s))" % (
        N, ", ".join("%s %.3f ms" % (k, v) for k, v in res.items()), c1[0] - c0[0], c1[1] - c0[1], 46, c0[1]), flush=True)
    s.close(); del s; torch.cuda.empty_cache()
