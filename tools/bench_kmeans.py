"""k-means (training) on the GPU vs scikit-learn at production-like sizes: 2M x 16-d, k = 256 (a fine codebook) and
2M x 64-d, k = 16 (a coarse codebook half).  Time includes the host <-> device copies of cis_kmeans."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from columbiaimagesearch_amd.lopq import train as T
for n, d, k, it in [(2_000_000, 16, 256, 20), (2_000_000, 64, 16, 10)]:
    x = np.random.RandomState(0).randn(n, d).astype(np.float32)
    T.kmeans_hip(x[:10000], k, 1)
    t = time.time(); C, inertia = T.kmeans_hip(x, k, it, n_init=1, random_state=1); dt = time.time() - t
    from sklearn.cluster import MiniBatchKMeans
    t = time.time(); km = MiniBatchKMeans(n_clusters=k, max_iter=it, n_init=1, batch_size=10000, random_state=1).fit(x[:200000]); ds = (time.time() - t) * n / 200000
    print("n=%d d=%d k=%d iters=%d: GPU %.2f s (inertia/n %.4f); scikit-learn MiniBatchKMeans (the reference's call) ~%.1f s extrapolated from 200k rows" % (n, d, k, it, dt, inertia / n, ds))
