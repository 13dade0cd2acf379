"""How many database vectors of the bench workload share their LOPQ code with another vector?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.argv = [sys.argv[0]]
import bench
model, z = bench.load_model("c4")
dev = torch.device("cuda", 0)
centers = bench.mixture_centers("descriptor", dev)
n = 2_000_000
x = bench.gen_chunk(centers, 0, n, dev)
co, fi = model.predict_batch_dev(x)
co = co.cpu().numpy().astype(np.int64); fi = fi.cpu().numpy().astype(np.uint8)
print("model D", getattr(model, "D", None), "V", model.V, "M", model.M, "fine shape", fi.shape, "coarse shape", co.shape)
key = np.zeros(n, dtype=[("c", np.int64), ("f", "V%d" % fi.shape[1])])
key["c"] = co[:, 0] * 65536 + co[:, 1]
key["f"] = fi.view("V%d" % fi.shape[1]).ravel()
u, cnts = np.unique(key, return_counts=True)
print("vectors", n, "distinct codes", len(u), "vectors in groups >1: %.4f" % (cnts[cnts > 1].sum() / n), "largest group", cnts.max())
cells, cc = np.unique(key["c"], return_counts=True)
print("cells used", len(cells), "largest cell share %.4f" % (cc.max() / n), "median cell", int(np.median(cc)))
for j in range(fi.shape[1]):
    h = np.bincount(fi[:, j], minlength=256) / n
    print("subq %d: effective centroids %.1f (1/sum p^2), max p %.3f" % (j, 1.0 / (h * h).sum(), h.max()))
