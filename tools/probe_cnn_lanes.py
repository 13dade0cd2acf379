"""Two (three) batches of 256 in flight through a CNN handle and its views (shared weights) on their own streams: do consecutive forwards fill each other's
workgroup rounds?  usage: probe_cnn_lanes.py dlib|cnn [batch] [lanes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from columbiaimagesearch_amd.featurizer.synthetic import dlib_weights, sentibank_weights
from columbiaimagesearch_amd.featurizer import DLibFaceNet, SentiBankNet
which = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
NL = int(sys.argv[3]) if len(sys.argv) > 3 else 2
if which == "dlib":
    mk = lambda: DLibFaceNet(dlib_weights(0)); xs = lambda: (torch.rand(B, 150, 150, 3, device="cuda") * 255).contiguous(); od = 128; mac = 270854144
else:
    mk = lambda: SentiBankNet(sentibank_weights(0)); xs = lambda: (torch.randn(B, 3, 227, 227, device="cuda") * 50).contiguous(); od = 4096; mac = 720310816
nets = [mk()]   # the views are created after the one-at-a-time measurement: from the first view on every handle runs its batch as ONE chain
x = [xs() for _ in range(NL)]
out = [torch.empty(B, od, device="cuda") for _ in range(NL)]
streams = [torch.cuda.Stream() for _ in range(NL)]
def run(K, lanes):
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(K):
        l = i % lanes
        with torch.cuda.stream(streams[l]):
            nets[l].forward_dev(x[l], out[l])
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / K
run(2, 1)
dt = min(run(24, 1) for _ in range(3))
print("%s batch %d, the handle alone (no views%s): %.3f ms per forward  %.0f descriptors/s  MFMA(f32) util %.3f" % (
    which, B, "; two half-batch chains on its own streams" if which == "dlib" else "", dt * 1e3, B / dt, 2.0 * mac * B / dt / 157.3e12))
while len(nets) < NL:
    nets.append(nets[0].view())
for lanes in range(1, NL + 1):
    run(2 * lanes, lanes)
    dt = min(run(24, lanes) for _ in range(3))
    print("%s batch %d, %d forward(s) in flight on the handle and its views (one chain each): %.3f ms per forward  %.0f descriptors/s  MFMA(f32) util %.3f" % (which, B, lanes, dt * 1e3, B / dt, 2.0 * mac * B / dt / 157.3e12))
ref = out[0].clone()
with torch.cuda.stream(streams[0]):
    nets[0].forward_dev(x[0], out[0])
torch.cuda.synchronize()
print("same descriptors alone and in flight:", bool(torch.equal(ref, out[0])))
