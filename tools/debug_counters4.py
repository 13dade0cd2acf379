"""Cycle breakdown of k_adc_scan4 (a build with -DCIS_S3_COUNTERS): per slot, thread 0 of every workgroup, s_memtime ticks scaled so that
the per-workgroup kernel time equals the measured launch.  usage: CIS_LIB_PATH=.../libcis_s3ctr.so python tools/debug_counters4.py [config]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
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
s.set_profiling(True, scan_only=True)
s.search_batch_dev(q, quota=10000, limit=100); torch.cuda.synchronize()
ms = s.read_profile()
fn(buf, 1)
b = [int(x) for x in buf]
slots, wgs = b[0], b[10]
print("slots %d  workgroups %d  profile %s" % (slots, wgs, ms))
tick = 1.0
print("  kernel per workgroup %.0f ticks; descriptor rounds %.0f ticks per workgroup" % (b[9] / wgs, b[1] / wgs))
names = {13: "B0 passed", 11: "tables staged + prefetch issued", 2: "B1 passed", 3: "sample pass + B2", 4: "thresholds + B3", 12: "main pass end", 5: "B4 passed", 6: "verify/cut/write + B5"}
prev = 0.0
for i in (13, 11, 2, 3, 4, 12, 5, 6):
    t = b[i] / max(slots, 1)
    print("  %-34s %8.1f ticks (+%.1f)" % (names[i], t, t - prev)); prev = t
print("  slots per workgroup %.2f; slot total x slots/wg = %.0f ticks" % (slots / wgs, prev * slots / wgs))
