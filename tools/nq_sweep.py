"""Scan-kernel time and step time against the batch size on the C4 index (is the static slot schedule quantised into rounds?).
usage: python tools/nq_sweep.py [nq,nq,...]"""
import os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench as B
from columbiaimagesearch_amd.lopq import LOPQSearcherHIP

N = int(os.environ.get("CIS_BENCH_N", 10_000_000)); NCH = 10
dev = torch.device("cuda", 0)
model, z = B.load_model("c4")
P = B.mixture_centers("descriptor", dev)
co, fi = [], []
for c in range(NCH):
    a, b = model.predict_batch_dev(B.gen_chunk(P, c, N // NCH, dev)); co.append(a); fi.append(b)
s = LOPQSearcherHIP(model)
s.add_codes_dev(torch.cat(co), torch.cat(fi), torch.arange(N, dtype=torch.int64, device=dev), dedup=False)
x0 = B.gen_chunk(P, 0, N // NCH, dev)
nqs = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "4096,6144,7168,7680,8192,9216,10240,12288,16384").split(",")]
qall = B.make_queries(x0, 0, max(nqs), dev)
for nq in nqs:
    q = qall[:nq].contiguous()
    for _ in range(3):
        s.search_batch_dev(q, quota=10000, limit=100)
    torch.cuda.synchronize()
    s.set_profiling(True, scan_only=True); s.read_profile()
    t = time.perf_counter(); reps = 10
    for _ in range(reps):
        s.search_batch_dev(q, quota=10000, limit=100)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / reps * 1e3
    pr = s.read_profile(); s.set_profiling(False)
    st = s.last_stats()
    print("nq %6d  items %6d (slots ~%5d = %.2f rounds of 1024)  step %.3f ms  scan kernel %.3f ms  -> %.1f us per 1000 slots, %.2f M q/s" % (
        nq, st["items"], st["items"] // 4, st["items"] / 4 / 1024.0, ms, pr["scan_kernel_ms"] / max(pr["scan_launches"], 1),
        pr["scan_kernel_ms"] / max(pr["scan_launches"], 1) * 1e3 / (st["items"] / 4 / 1000.0), nq / ms / 1e3), flush=True)
