import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.argv = [sys.argv[0]]
import bench
from columbiaimagesearch_amd import _lib
from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
model, z = bench.load_model("c4")
dev = torch.device("cuda", 0)
centers = bench.mixture_centers("descriptor", dev)
N = int(os.environ.get("CIS_DBG_N", 10_000_000)); chunk = N // 80
cs, fs = [], []
for c in range(80):
    x = bench.gen_chunk(centers, c, chunk, dev)
    co, fi = model.predict_batch_dev(x); cs.append(co); fs.append(fi)
coarse = torch.cat(cs).cpu().numpy().view(np.uint16); fine = torch.cat(fs).cpu().numpy()
s = LOPQSearcherHIP(model); s.add_codes_array(coarse, fine, ids=np.arange(N, dtype=np.int64), dedup=False)
x0 = bench.gen_chunk(centers, 0, chunk, dev)
q = bench.make_queries(x0, 0, 8192, dev)
s.search_batch_dev(q, quota=10000, limit=100); torch.cuda.synchronize()
L = _lib.lib(); fn = L.cis_debug_counters; fn.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 16)()
fn(buf, 1)
s.search_batch_dev(q, quota=10000, limit=100); torch.cuda.synchronize()
fn(buf, 1)
st = s.last_stats()
names = ["compactions", "exact_compactions", "clk_loop", "slow_iters", "appended", "clk_compact", "clk_slow", "clk_total", "clk_exact_fallback", "clk_final", "clk_epilogue_to_exact_end", "clk_prologue"]
d = {n: int(buf[i]) for i, n in enumerate(names)}
print(d, st)
items = st["items"]
print("per item: compactions %.1f appended %.0f slow iterations %.1f" % (d["compactions"] / items, d["appended"] / items, d["slow_iters"] / items))
print("exact fallback %.3f  final approx %.3f  loop end..exact end (incl. barrier) %.3f  prologue (tables) %.3f" % (
    d["clk_exact_fallback"] / d["clk_total"], d["clk_final"] / d["clk_total"], d["clk_epilogue_to_exact_end"] / d["clk_total"], d["clk_prologue"] / d["clk_total"]))
print("wave time shares: compaction in loop %.3f  slow path (excl. compaction) %.3f  loop %.3f  prologue+epilogue %.3f" % (
    d["clk_compact"] / d["clk_total"], d["clk_slow"] / d["clk_total"], d["clk_loop"] / d["clk_total"], 1 - d["clk_loop"] / d["clk_total"]))
