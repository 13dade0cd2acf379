"""Production operating point of the reference's release configurations (conf/conf_search_dlibface_release.json:10-16: pca 128, V=2048,
M=8; conf_search_sbpycaffe_release.json:9-16: V=4096): millions of tiny cells.  10M x 128-d descriptor-like vectors, quota 10000, limit 100.
The model has the right shapes and k-means-fitted codebooks (torch Lloyd iterations on the GPU) but identity local rotations -- timing only;
a spot check against the oracle guards the results.  Usage: python tools/bench_prodv.py [V] [N] [nq]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
V = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
NQ = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
sys.argv = [sys.argv[0]]
import bench
from columbiaimagesearch_amd.lopq import LOPQModel, LOPQSearcherHIP
dev = torch.device("cuda", 0)
P = bench.mixture_centers("descriptor", dev)
M, K, D = 8, 256, 128


def lloyd(x, k, iters, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    C = x[torch.randperm(x.shape[0], generator=g, device=dev)[:k]].clone()
    for _ in range(iters):
        a = torch.cat([torch.cdist(x[i:i + 65536], C).argmin(1) for i in range(0, x.shape[0], 65536)])
        S = torch.zeros_like(C).index_add_(0, a, x)
        n = torch.bincount(a, minlength=k).clamp(min=1).unsqueeze(1)
        C = S / n
    return C


xs = bench.gen_chunk(P, 999, 400000, dev).float()
h = D // 2
Cs, subs = [], []
for s in range(2):
    xh = xs[:, s * h:(s + 1) * h].contiguous()
    C = lloyd(xh, V, 8, s)
    Cs.append(C.cpu().numpy().astype(np.float32))
    res = xh - C[torch.cdist(xh, C).argmin(1)] if V <= 4096 else xh
    subs.append([lloyd(res[:, j * 16:(j + 1) * 16].contiguous(), K, 8, 10 + j).double().cpu().numpy() for j in range(M // 2)])
Rs = tuple(np.broadcast_to(np.eye(h), (V, h, h)).copy() for _ in range(2))
mus = tuple(np.zeros((V, h)) for _ in range(2))
model = LOPQModel(V=V, M=M, parameters=(tuple(Cs), Rs, mus, tuple(subs)))
chunk = N // 80
cs, fs, ev = [], [], []
for c in range(80):
    x = bench.gen_chunk(P, c, chunk, dev).float().contiguous()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); co, fi = model.predict_batch_dev(x); e1.record(); ev.append((e0, e1)); cs.append(co); fs.append(fi)
torch.cuda.synchronize()
enc = sum(a.elapsed_time(b) for a, b in ev) / 1e3
coarse = torch.cat(cs).cpu().numpy().view(np.uint16); fine = torch.cat(fs).cpu().numpy()
s = LOPQSearcherHIP(model); s.add_codes_array(coarse, fine, ids=np.arange(N, dtype=np.int64), dedup=False)
cell = coarse[:, 0].astype(np.int64) * V + coarse[:, 1]
occ = np.bincount(cell, minlength=V * V)
print("V=%d: %d x %d-d, encode %.1f M vectors/s; %d of %d cells occupied, mean %.1f / max %d codes per occupied cell" % (
    V, N, D, N / enc / 1e6, (occ > 0).sum(), V * V, occ[occ > 0].mean(), occ.max()))
q = bench.make_queries(bench.gen_chunk(P, 0, chunk, dev), 0, NQ, dev).float().contiguous()
FAST = os.environ.get("PRODV_FAST") == "1"  # profiling runs: the quota-10000 batches only, no route checks
for quota, limit in (((10000, 100),) if FAST else ((1000, 100), (10000, 100))):
    out = s.search_batch_dev(q, quota=quota, limit=limit); torch.cuda.synchronize()
    s.set_profiling(True); s.read_profile()
    t = time.perf_counter(); reps = 3
    for _ in range(reps):
        out = s.search_batch_dev(q, quota=quota, limit=limit)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    pr = s.read_profile(); st = s.last_stats()
    print("quota %5d limit %d: %.2f ms per %d queries = %.0f queries/s; visited cells/query %.0f, work items %d, candidates/query %.0f; stages %s" % (
        quota, limit, dt * 1e3, NQ, NQ / dt, float(out["visited"].float().mean()), st["items"], st["candidates"] / NQ,
        {k: round(v / reps, 2) for k, v in pr.items() if k.endswith("_ms")}))
    s.set_profiling(False)
    # three batches in flight through views of the index (what bench.py does at V = 16): the serial front end of one batch under
    # the ranking kernels of another
    from columbiaimagesearch_amd.streams import lane_streams
    NL = int(os.environ.get("PRODV_LANES", "3"))
    lanes = [(s, torch.cuda.current_stream())] + [(s.view(), ls) for ls in lane_streams(NL - 1)]
    for sv, stq in lanes:
        with torch.cuda.stream(stq):
            sv.search_batch_dev(q, quota=quota, limit=limit)
    torch.cuda.synchronize()
    t = time.perf_counter(); reps = 3 * NL
    for i in range(reps):
        sv, stq = lanes[i % NL]
        with torch.cuda.stream(stq):
            sv.search_batch_dev(q, quota=quota, limit=limit)
    torch.cuda.synchronize()
    dtp = (time.perf_counter() - t) / reps
    print("    %s batches in flight: %.2f ms per %d queries = %.0f queries/s" % ({2: "two", 3: "three", 4: "four"}.get(NL, str(NL)), dtp * 1e3, NQ, NQ / dtp))
    for sv, _ in lanes[1:]:
        sv.close()
if FAST:
    sys.exit(0)
# spot check against the oracle
from oracle import lopq_oracle as O
om = O.OracleModel(list(Cs), list(Rs), list(mus), [list(subs[0]), list(subs[1])])
oi = O.OracleCSRIndex(om, coarse, fine)
ids = out["ids"].cpu().numpy(); dd = out["dists"].cpu().numpy(); vis = out["visited"].cpu().numpy(); qh = q.cpu().numpy()
ok = True
for qi in range(4):
    eid, ed, ev_ = oi.search(qh[qi], quota=10000, limit=100)
    ok = ok and np.array_equal(ids[qi, :len(eid)], eid) and ev_ == vis[qi] and np.allclose(dd[qi, :len(ed)], ed, rtol=1e-9)
print("parity (4 queries vs oracle, quota 10000): %s" % ok)
# every query of the batch, bit for bit, across the library's own routes: direct ADC (default) / float64 tables + flat lookups /
# the float64 scan kernel per work item
ref = {k: out[k].cpu().numpy() for k in ("ids", "dists", "n_found", "visited")}
same = {}
for name, env, mode in (("tables", {"CIS_NO_DIRECT": "1"}, 0), ("serial_plan", {"CIS_NO_PAR_PLAN": "1"}, 0), ("exact_scan", {}, 1)):
    os.environ.update(env)
    s.set_scan_mode(mode=mode)
    nqc = NQ if name != "exact_scan" else min(NQ, 256)
    o2 = s.search_batch_dev(q[:nqc].contiguous(), quota=10000, limit=100)
    torch.cuda.synchronize()
    for k in env:
        del os.environ[k]
    s.set_scan_mode(mode=0)
    same[name] = all(np.array_equal(o2[k].cpu().numpy().view(np.uint64) if k == "dists" else o2[k].cpu().numpy(),
                                    ref[k][:nqc].view(np.uint64) if k == "dists" else ref[k][:nqc]) for k in ref)
print("routes agree on the whole batch (ids, float64 distance bits, counts): %s" % same)
