"""Time the DeepSentibank forward at batch 256 on the GPU (descriptors/s, MFMA utilisation vs the f32 peak)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from columbiaimagesearch_amd.featurizer.synthetic import sentibank_weights
from columbiaimagesearch_amd.featurizer import SentiBankNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
net = SentiBankNet(sentibank_weights(0))
x = (torch.randn(B, 3, 227, 227, device="cuda") * 50).contiguous()
out = torch.empty(B, 4096, device="cuda")
for _ in range(2):
    net.forward_dev(x, out)
torch.cuda.synchronize()
K = 5
t = time.perf_counter()
for _ in range(K):
    net.forward_dev(x, out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / K
flops = 2.0 * 720310816 * B
print("batch %d: %.3f ms  %.0f descriptors/s  %.1f TFLOP/s  MFMA(f32) util %.3f" % (B, dt * 1e3, B / dt, flops / dt / 1e12, flops / dt / 157.3e12))
