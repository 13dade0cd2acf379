#!/usr/bin/env python3
"""Driver for the PMC passes of the HBM-streaming kernel: builds the c4x index (N x 8-byte codes) and runs REPS exhaustive searches of
NQ queries -- nothing else touches the GPU afterwards, so every k_adc_stream<.., false> dispatch in the counter file is one full launch.
usage: stream_pmc_driver.py N NQ REPS"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench

N, nq, reps = int(float(sys.argv[1])), int(sys.argv[2]), int(sys.argv[3])
from columbiaimagesearch_amd import _lib
from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
device = torch.device("cuda", 0)
_lib.check(_lib.lib().cis_set_device(0))
model, z = bench.load_model("c4")
P = bench.mixture_centers("descriptor", device)
N -= N % 80
chunk_n = N // 80
s = LOPQSearcherHIP(model)
for c in range(80):
    x = bench.gen_chunk(P, c, chunk_n, device)
    co, fi = [], []
    for a in range(0, chunk_n, 1 << 20):
        c_, f_ = model.predict_batch_dev(x[a:a + (1 << 20)])
        co.append(c_); fi.append(f_)
    s.add_codes_dev(torch.cat(co), torch.cat(fi), torch.arange(c * chunk_n, (c + 1) * chunk_n, dtype=torch.int64, device=device), dedup=False)
    del x, co, fi
x0 = bench.gen_chunk(P, 0, min(chunk_n, 1 << 20), device)
q = bench.make_queries(x0, 0, 8192, device)[:nq].contiguous()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.search_batch_dev(q, quota=N, limit=100)
torch.cuda.synchronize()
e0.record()
for _ in range(reps):
    s.search_batch_dev(q, quota=N, limit=100)
e1.record()
torch.cuda.synchronize()
print("N %d nq %d: %.3f ms per exhaustive batch, kernel %s, stream counters %r" % (N, nq, e0.elapsed_time(e1) / reps, s.last_stats()["scan_kernel"], s.stream_counters()))
