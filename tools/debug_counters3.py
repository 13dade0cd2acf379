"""Cycle breakdown of k_adc_scan3 (a build with -DCIS_S3_COUNTERS): per slot, wave 0 of every workgroup, s_memtime (100 MHz) ticks.
usage: CIS_LIB_PATH=.../libcis_s3ctr.so python tools/debug_counters3.py [config]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
cfgname = sys.argv[1] if len(sys.argv) > 1 else "c4"
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
coarse = torch.cat(cs).cpu().numpy().view(np.uint16); fine = torch.cat(fs).cpu().numpy()
s = LOPQSearcherHIP(model); s.add_codes_array(coarse, fine, ids=np.arange(N, dtype=np.int64), dedup=False)
q = bench.make_queries(bench.gen_chunk(centers, 0, chunk, dev), 0, 8192, dev)
s.search_batch_dev(q, quota=10000, limit=100); torch.cuda.synchronize()
fn = _lib.lib().cis_debug_counters3; fn.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 16)()
fn(buf, 1)
s.search_batch_dev(q, quota=10000, limit=100); torch.cuda.synchronize()
fn(buf, 1)
b = [int(x) for x in buf]
slots, wgs = b[0], b[10]
names = ["T32 loads + max (to barrier 1)", "barrier 1 passed", "tables written + barrier 2", "loop end", "final compaction", "barrier 3 passed", "slot end"]
print("slots %d  workgroups %d  candidates/query %.0f" % (slots, wgs, s.last_stats()["candidates"] / 8192))
prev = 0.0
for i, n in enumerate(names):
    t = b[i + 1] / slots / 100.0
    print("  %-34s %7.2f us (+%.2f)" % (n, t, t - prev)); prev = t
if b[11]:
    print("  two-pass form: pass 1 end %.2f, barrier %.2f, thresholds %.2f, pass 2 end %.2f (same unit, from slot start)" % tuple(b[i] / slots / 100.0 for i in (11, 12, 13, 14)))
print("  queue fetch per attempt           %7.2f us; kernel per workgroup %.1f us, slots per workgroup %.2f" % (b[8] / (slots + 8 * wgs) / 100.0, b[9] / wgs / 100.0, slots / wgs))
