"""End-to-end ingest rate (BASELINE config C5 shape): batch-256 DeepSentibank forward -> L2 normalise -> LOPQ encode
(PCA 4096 -> 256, V=16, M=16, synthetic model parameters of the right shapes) -> index insert.  descriptors/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from columbiaimagesearch_amd.featurizer.synthetic import sentibank_weights
from columbiaimagesearch_amd.featurizer import SentiBankNet
from columbiaimagesearch_amd.ingest import BatchIngest
from columbiaimagesearch_amd.lopq import LOPQModelPCA, LOPQSearcherHIP


def synthetic_c3_model(seed=0, D_in=4096, D=256, V=16, M=16, K=256):
    rs = np.random.RandomState(seed)
    h, w = D // 2, D // M
    P, _ = np.linalg.qr(rs.randn(D_in, D))
    Cs = tuple((rs.randn(V, h) * 0.1).astype(np.float32) for _ in range(2))
    Rs = tuple(np.stack([np.linalg.qr(rs.randn(h, h))[0] for _ in range(V)]) for _ in range(2))
    mus = tuple(rs.randn(V, h) * 0.01 for _ in range(2))
    subs = tuple([rs.randn(K, w) * 0.05 for _ in range(M // 2)] for _ in range(2))
    return LOPQModelPCA(V=V, M=M, renorm=True, parameters=(Cs, Rs, mus, subs, P, rs.randn(D_in) * 0.01))


B = 256
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
net = SentiBankNet(sentibank_weights(0))
model = synthetic_c3_model()
s = LOPQSearcherHIP(model)
ing = BatchIngest(net, model, s)
x = (torch.randn(B, 3, 227, 227, device="cuda") * 50.0).contiguous()
for _ in range(2):
    ing.ingest_batch(x)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(steps):
    ing.ingest_batch(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / steps
t = time.perf_counter()
for _ in range(steps):
    ing.encode_batch_dev(x)
torch.cuda.synchronize()
de = (time.perf_counter() - t) / steps
print("batch %d: CNN + normalise + encode %.3f ms (%.0f descriptors/s); with device-side insert %.3f ms (%.0f descriptors/s); indexed %d" % (
    B, de * 1e3, B / de, dt * 1e3, B / dt, s.get_nb_indexed()))
