"""Host time of one search call against the device time of the batch (C4 shape at 2M vectors): if they are close the step is bound by
the ~40 kernel launches of a batch, not by the kernels.  usage: python tools/profile_host.py [n_vectors]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
n_arg = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
sys.argv = [sys.argv[0]]
import bench
from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
dev = torch.device("cuda", 0)
model, z = bench.load_model("c4")
P = bench.mixture_centers("descriptor", dev)
N, chunks = n_arg, 16
s = LOPQSearcherHIP(model)
for c in range(chunks):
    x = bench.gen_chunk(P, c, N // chunks, dev)
    co, fi = model.predict_batch_dev(x)
    s.add_codes_dev(co, fi, torch.arange(c * (N // chunks), (c + 1) * (N // chunks), dtype=torch.int64, device=dev), dedup=False)
x0 = bench.gen_chunk(P, 0, N // chunks, dev)
q = bench.make_queries(x0, 0, 8192, dev)
out = s.search_batch_dev(q, quota=10000, limit=100)
torch.cuda.synchronize()
for reps in (1, 20):
    host = []
    t0 = time.perf_counter()
    for _ in range(reps):
        t = time.perf_counter()
        s.search_batch_dev(q, quota=10000, limit=100, out=out)
        host.append(time.perf_counter() - t)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("%2d back-to-back calls: host time per call %.3f ms (first %.3f, last %.3f), all enqueued after %.3f ms, finished after %.3f ms = %.3f ms per batch"
          % (reps, sum(host) / reps * 1e3, host[0] * 1e3, host[-1] * 1e3, t_enq * 1e3, t_all * 1e3, t_all / reps * 1e3))
