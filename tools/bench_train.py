"""Fit time of the production training shape on one MI355X: PCA on 200 k x 4096-d features (pca_subsample), LOPQ (V = 16, M = 16,
256 sub-quantizer clusters) on 2 M x 256-d projected vectors -- cufacesearch/cufacesearch/searcher/searcher_lopqhbase.py:272-420 feeds
lopq.LOPQModelPCA.fit exactly these; the reference accumulates covariances sample by sample in Python (lopq/lopq/model.py:142-155,
:263-267: hours).  Here the accumulators, the projections and the k-means steps run on the GPU (train.ACCUM_BACKEND / KMEANS_BACKEND =
"hip"); the eigendecompositions (one 4096 x 4096, 2 V of h x h) and the bucket allocation stay LAPACK / host.
usage: python tools/bench_train.py [n_pca n_train]   (default 200000 2000000)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from columbiaimagesearch_amd.lopq import train as T

n_pca = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
n_train = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
T.ACCUM_BACKEND = "hip"
T.KMEANS_BACKEND = "hip"
rs = np.random.RandomState(0)
t0 = time.perf_counter()
# post-ReLU-like 4096-d features with a decaying spectrum (so that the PCA has something to find)
basis = rs.randn(64, 4096).astype(np.float32)
Xp = np.maximum(rs.randn(n_pca, 64).astype(np.float32) @ basis * 0.2 + 0.3 * rs.randn(n_pca, 4096).astype(np.float32), 0)
Xp /= np.maximum(np.linalg.norm(Xp, axis=1, keepdims=True), 1e-12)
t_gen = time.perf_counter() - t0
t = time.perf_counter()
pca, dims = T.train_pca(Xp, pca_dims=256)
t_pca = time.perf_counter() - t
t = time.perf_counter()
# the 2 M training vectors arrive as 4096-d features and are projected in slices (searcher_lopqhbase.py:340 does this per batch)
from columbiaimagesearch_amd.lopq import LOPQModelPCA
pm = LOPQModelPCA(V=16, M=16, renorm=True, parameters=(None, None, None, None, pca["P"], pca["mu"]))
Y = np.empty((n_train, 256), dtype=np.float32)
t_proj = 0.0
for a in range(0, n_train, 100000):
    idx = rs.randint(0, n_pca, size=min(100000, n_train - a))
    xb = Xp[idx] + 0.02 * rs.randn(len(idx), 4096).astype(np.float32)
    tp = time.perf_counter()
    Y[a:a + len(idx)] = pm.apply_PCA(xb)   # cis_apply_pca: the float64 matrix cores (host arrays in and out)
    t_proj += time.perf_counter() - tp
t_prep = time.perf_counter() - t - t_proj
t = time.perf_counter()
Cs, Rs, mus, subs = T.train(Y, V=16, M=16, subquantizer_clusters=256, kmeans_coarse_iters=10, kmeans_local_iters=20, n_init=1, random_state=1)
t_fit = time.perf_counter() - t
# quality: mean squared reconstruction error of 20 k training vectors through the fitted model (lopq/lopq/eval.py:146-161 semantics)
from columbiaimagesearch_amd.lopq import LOPQModel
m = LOPQModel(V=16, M=16, parameters=(Cs, Rs, mus, subs))
sample = Y[:20000].astype(np.float64)
co, fi = m.predict_batch(sample)
rec = np.stack([m.reconstruct((tuple(c), tuple(f))) for c, f in zip(co[:2000], fi[:2000])])
err = float(((rec - sample[:2000]) ** 2).sum(axis=1).mean())
print("training at the production shape: data generation %.1f s; PCA fit on %d x 4096 %.1f s; apply_PCA of %d vectors on the GPU (host arrays in and out) %.1f s; "
      "LOPQ fit (V=16, M=16, K=256) on %d x 256 %.1f s; mean squared reconstruction error %.4f of unit-norm vectors (variance floor of "
      "the data ~ its within-component noise)" % (t_gen, n_pca, t_pca, n_train, t_proj, n_train, t_fit, err), flush=True)
