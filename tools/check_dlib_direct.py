"""dlib face ResNet: the direct 3x3 kernel against the implicit-GEMM route (CIS_CNN_NO_DIRECT) -- agreement and time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from columbiaimagesearch_amd.featurizer.synthetic import dlib_weights
from columbiaimagesearch_amd.featurizer import DLibFaceNet
net = DLibFaceNet(dlib_weights(0))
def run(x, direct):
    if direct: os.environ.pop("CIS_CNN_NO_DIRECT", None)
    else: os.environ["CIS_CNN_NO_DIRECT"] = "1"
    out = torch.empty(x.shape[0], 128, device="cuda")
    net.forward_dev(x, out); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): net.forward_dev(x, out)
    torch.cuda.synchronize()
    return out.cpu().numpy(), (time.perf_counter() - t) / 5
for B in (1, 3, 37, 256):
    torch.manual_seed(B)
    x = (torch.rand(B, 150, 150, 3, device="cuda") * 255).contiguous()
    a, ta = run(x, True)
    b, tb = run(x, False)
    print("batch %d: direct %.3f ms  igemm %.3f ms  max|diff| %.3e  max|out| %.3e  finite %s" % (B, ta * 1e3, tb * 1e3, np.abs(a - b).max(), np.abs(b).max(), np.isfinite(a).all()))
os.environ.pop("CIS_CNN_NO_DIRECT", None)
x = (torch.rand(256, 150, 150, 3, device="cuda") * 255).contiguous()
a, _ = run(x, True)
b, _ = run(x[100:101].contiguous(), True)
print("batch invariance (row 100 of 256 vs alone): bit-identical %s" % np.array_equal(a[100], b[0]))
