import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from conftest import load_golden
from test_lopq_hip_parity import hip_model, _build_searcher
name = sys.argv[1] if len(sys.argv) > 1 else "c1"
quota, limit = int(sys.argv[2]), int(sys.argv[3])
z, X, Q = load_golden(name)
m = hip_model(z)
s = _build_searcher(name, z, X, m)
a = s.search_batch(Q, quota=quota, limit=limit)
print("fast stats", s.last_stats())
s.set_scan_mode(exact_only=True)
b = s.search_batch(Q, quota=quota, limit=limit)
bad = np.nonzero((a["ids"] != b["ids"]).any(axis=1))[0]
print("queries differing:", len(bad), "of", len(Q))
for q in bad[:3]:
    d = np.nonzero(a["ids"][q] != b["ids"][q])[0]
    print("q", q, "first diff rank", d[0], "n diff", len(d))
    print(" fast ", a["ids"][q][d[0]:d[0]+5], a["dists"][q][d[0]:d[0]+5])
    print(" exact", b["ids"][q][d[0]:d[0]+5], b["dists"][q][d[0]:d[0]+5])
    missing = set(b["ids"][q]) - set(a["ids"][q])
    print(" missing from fast:", len(missing))
from oracle import lopq_oracle as O
om = O.OracleModel.from_npz(z)
if name == "c1":
    oc, of = O.compute_codes(om, X)
else:
    oc, of = z["coarse"], z["fine"]
oi = O.OracleCSRIndex(om, oc, of)
nb_f = nb_e = 0
for q in range(len(Q)):
    ids, dd, vis = oi.search(Q[q], quota=quota, limit=limit)
    nb_f += int(not np.array_equal(a["ids"][q][:len(ids)], ids))
    nb_e += int(not np.array_equal(b["ids"][q][:len(ids)], ids))
print("vs oracle: fast wrong", nb_f, "exact wrong", nb_e)
V = om.V
cellid = oc[:, 0].astype(np.int64) * V + oc[:, 1]
sizes = np.bincount(cellid, minlength=V * V)
for q in bad[:6]:
    missing = sorted(set(a["ids"][q].tolist()) - set(b["ids"][q].tolist()))
    xq = O.apply_pca(om, Q[q]) if om.has_pca else Q[q]
    order = [tuple(int(c) for c in cell) for _, cell in O.multisequence(om, xq)]
    for mid in missing:
        c = (int(oc[mid, 0]), int(oc[mid, 1]))
        pos = int(np.nonzero(np.nonzero(cellid == cellid[mid])[0] == mid)[0][0])
        print("q", q, "missing id", mid, "cell", c, "size", sizes[cellid[mid]], "visit rank", order.index(c), "pos in cell", pos)
