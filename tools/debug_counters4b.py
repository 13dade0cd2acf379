"""Phase breakdown of k_adc_scan4 (a build with -DCIS_S3_COUNTERS): per slot, thread 0 of every workgroup, s_memtime (100 MHz) ticks.
usage: CIS_LIB_PATH=.../libcis_s3ctr.so python tools/debug_counters4b.py [config] [nq]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
cfgname = sys.argv[1] if len(sys.argv) > 1 else "c4"
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
sys.argv = [sys.argv[0]]
import bench
from columbiaimagesearch_amd import _lib
from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
cfg = bench.CONFIGS[cfgname]
model, z = bench.load_model(cfg["fixture"])
dev = torch.device("cuda", 0)
centers = bench.mixture_centers(cfg["gen"], dev)
N = cfg["n"]; chunk = N // 80
cs, fs = [], []
for c in range(80):
    co, fi = model.predict_batch_dev(bench.gen_chunk(centers, c, chunk, dev)); cs.append(co); fs.append(fi)
s = LOPQSearcherHIP(model); s.add_codes_dev(torch.cat(cs), torch.cat(fs), torch.arange(N, dtype=torch.int64, device=dev), dedup=False)
q = bench.make_queries(bench.gen_chunk(centers, 0, chunk, dev), 0, nq, dev)
s.search_batch_dev(q, quota=10000, limit=100); torch.cuda.synchronize()
fn = _lib.lib().cis_debug_counters3; fn.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 16)()
fn(buf, 1)
s.search_batch_dev(q, quota=10000, limit=100); torch.cuda.synchronize()
fn(buf, 1)
b = [int(x) for x in buf]
slots, wgs = b[0], b[10]
print("%s nq %d: slots %d  workgroups %d  candidates/query %.0f" % (cfgname, nq, slots, wgs, s.last_stats()["candidates"] / nq))
seq = [(11, "tables requested, scales written"), (2, "barrier 1 (tables in LDS)"), (3, "sample pass + barrier 2"), (4, "thresholds + barrier 3"),
       (13, "main pass"), (12, "second gather (per-query split)"), (5, "barrier 4"), (6, "verification, cut, write-out + barrier 5")]
prev = 0.0
for i, n in seq:
    t = b[i] / max(slots, 1) / 100.0
    print("  %-44s %7.2f us (+%.2f)" % (n, t, t - prev)); prev = t
print("  kernel per workgroup %.1f us, slots per workgroup %.2f" % (b[9] / max(wgs, 1) / 100.0, slots / max(wgs, 1)))
