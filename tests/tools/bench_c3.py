"""BASELINE config C3 shape: 4096-d float32 non-negative (post-ReLU-like) unit features, LOPQModelPCA 4096 -> 256, V=16, M=16,
1M-vector index on one GPU.  Reports the encode rate and queries/s at quota=10000, limit=100 (8192 queries per step), with a
parity spot check against the oracle.  The model has the right shapes but synthetic (untrained) parameters: PCA = random
orthonormal basis, coarse centroids drawn from projected data, random local rotations, sub-centroids drawn from residuals."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from columbiaimagesearch_amd.lopq import LOPQModelPCA, LOPQSearcherHIP

D_IN, D, V, M, K = 4096, 256, 16, 16, 256
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
NQ, QUOTA, LIMIT, CH = 8192, 10000, 100, 32768
dev = torch.device("cuda", 0)
rs = np.random.RandomState(3)
gen = torch.Generator(device=dev); gen.manual_seed(5)
centers = torch.randn(64, D_IN, generator=gen, device=dev)


def chunk(n, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    comp = torch.randint(0, 64, (n,), generator=g, device=dev)
    x = torch.clamp(centers[comp] + 0.8 * torch.randn((n, D_IN), generator=g, device=dev), min=0.0)
    return (x / x.norm(dim=1, keepdim=True)).contiguous()


# synthetic parameters from a data sample (host)
P, _ = np.linalg.qr(rs.randn(D_IN, D))
xs = chunk(20000, 999).double().cpu().numpy()
mu = xs.mean(0)
y = (xs - mu).dot(P); y /= np.linalg.norm(y, axis=1, keepdims=True)
h, w, nf = D // 2, D // M, M // 2
Cs = tuple(y[rs.choice(len(y), V, replace=False)][:, s * h:(s + 1) * h].astype(np.float32) for s in range(2))
Rs = tuple(np.stack([np.linalg.qr(rs.randn(h, h))[0] for _ in range(V)]) for _ in range(2))
mus = tuple(np.zeros((V, h)) for _ in range(2))
subs = tuple([y[rs.choice(len(y), K, replace=False)][:, :w] * 0.3 for _ in range(nf)] for _ in range(2))
model = LOPQModelPCA(V=V, M=M, renorm=True, parameters=(Cs, Rs, mus, subs, P, mu))

s = LOPQSearcherHIP(model)
ev, t0 = [], time.time()
cs, fs = [], []
for a in range(0, N, CH):
    x = chunk(min(CH, N - a), 1000 + a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); co, fi = model.predict_batch_dev(x); e1.record(); ev.append((e0, e1))
    cs.append(co); fs.append(fi)
torch.cuda.synchronize()
enc = sum(a.elapsed_time(b) for a, b in ev) / 1e3
coarse = torch.cat(cs).cpu().numpy().view(np.uint16); fine = torch.cat(fs).cpu().numpy()
s.add_codes_array(coarse, fine, ids=np.arange(N, dtype=np.int64), dedup=False)
q = chunk(NQ, 77)
for _ in range(2):
    out = s.search_batch_dev(q, quota=QUOTA, limit=LIMIT)
torch.cuda.synchronize()
s.set_profiling(True)
K_STEPS = 8
t = time.perf_counter()
for _ in range(K_STEPS):
    out = s.search_batch_dev(q, quota=QUOTA, limit=LIMIT)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / K_STEPS
prof = s.read_profile(); st = s.last_stats()
print("C3: %d x %d-d -> PCA %d, V=%d, M=%d: encode %.2f M vectors/s; search %.3f ms per %d queries = %.2f M queries/s; %.0f candidates/query; stages %s" % (
    N, D_IN, D, V, M, N / enc / 1e6, dt * 1e3, NQ, NQ / dt / 1e6, st["candidates"] / NQ, {k: round(v / K_STEPS, 3) for k, v in prof.items() if k.endswith("_ms")}))
algo = st["candidates"] * M
print("scan kernel: %.3f ms, algorithmic %.0f GB/s = %.3f of 8 TB/s" % (prof["scan_kernel_ms"] / K_STEPS, algo / (prof["scan_kernel_ms"] / K_STEPS / 1e3) / 1e9, algo / (prof["scan_kernel_ms"] / K_STEPS / 1e3) / 8e12))
# parity spot check against the oracle (CPU) on a few queries
from oracle import lopq_oracle as O
om = O.OracleModel(list(Cs), list(Rs), list(mus), [list(subs[0]), list(subs[1])], pca_P=P, pca_mu=mu, renorm=True)
oi = O.OracleCSRIndex(om, coarse, fine)
ids = out["ids"].cpu().numpy(); dists = out["dists"].cpu().numpy()
qh = q.cpu().numpy()
ok = True; err = 0.0
for qi in range(8):
    eids, ed, _ = oi.search(qh[qi], quota=QUOTA, limit=LIMIT)
    ok = ok and np.array_equal(ids[qi, :len(eids)], eids)
    err = max(err, float(np.max(np.abs(dists[qi, :len(ed)] - ed) / np.maximum(np.abs(ed), 1e-300))))
print("parity (8 queries vs oracle): ids bit-exact %s, max relative distance error %.2e" % (ok, err))
