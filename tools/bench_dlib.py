"""Time the dlib face ResNet forward at batch 256 (descriptors/s, MFMA utilisation vs the f32 peak)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from columbiaimagesearch_amd.featurizer.synthetic import dlib_weights
from columbiaimagesearch_amd.featurizer import DLibFaceNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
net = DLibFaceNet(dlib_weights(0))
x = (torch.rand(B, 150, 150, 3, device="cuda") * 255).contiguous()
out = torch.empty(B, 128, device="cuda")
for _ in range(2):
    net.forward_dev(x, out)
torch.cuda.synchronize()
K = 5
t = time.perf_counter()
for _ in range(K):
    net.forward_dev(x, out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / K
flops = 2.0 * 270854144 * B  # multiply-accumulates per face (oracle/dlib_oracle.py:mac_per_face)
print("batch %d: %.3f ms  %.0f descriptors/s  %.1f TFLOP/s  MFMA(f32) util %.3f" % (B, dt * 1e3, B / dt, flops / dt / 1e12, flops / dt / 157.3e12))
