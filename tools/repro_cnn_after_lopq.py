"""Repro: tests/test_cnn_hip_parity.py::test_featurizer_surface fails when tests/test_lopq_hip_parity.py ran first in the process."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
os.chdir(REPO)
import numpy as np, torch, pytest
which = sys.argv[1:] or ["tests/test_lopq_hip_parity.py"]
rc = pytest.main(which + ["-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"])
print("pytest rc", rc)
from columbiaimagesearch_amd.featurizer.synthetic import sentibank_weights
from columbiaimagesearch_amd.featurizer import SentiBankNet
from oracle import cnn_oracle as C
w = sentibank_weights(0)
rs = np.random.RandomState(0)
x = (rs.randn(4, 3, 227, 227) * 50).astype(np.float32)
ref = C.forward_torch(x, w)
for trial in range(2):
    net = SentiBankNet(w)
    for n in (1, 2, 1, 2, 4):
        h = net.forward(x[:n])
        d = net.forward_dev(torch.from_numpy(x[:n]).cuda()).cpu().numpy()
        err_h = np.abs(h - ref[:n]).max(axis=1) / np.abs(ref[:n]).max()
        err_d = np.abs(d - ref[:n]).max(axis=1) / np.abs(ref[:n]).max()
        print("trial", trial, "n", n, "host-entry err per row", np.round(err_h, 6), "dev-entry err per row", np.round(err_d, 6))
    net.close()
