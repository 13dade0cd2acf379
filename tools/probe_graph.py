"""Does replaying the forward as a captured HIP graph shorten the idle gaps between its launches?  usage: probe_graph.py dlib|cnn [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from columbiaimagesearch_amd.featurizer.synthetic import dlib_weights, sentibank_weights
from columbiaimagesearch_amd.featurizer import DLibFaceNet, SentiBankNet
which = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if which == "dlib":
    net = DLibFaceNet(dlib_weights(0)); x = (torch.rand(B, 150, 150, 3, device="cuda") * 255).contiguous(); out = torch.empty(B, 128, device="cuda"); mac = 270854144
else:
    net = SentiBankNet(sentibank_weights(0)); x = (torch.randn(B, 3, 227, 227, device="cuda") * 50).contiguous(); out = torch.empty(B, 4096, device="cuda"); mac = 720310816
def timed(f, K=10):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(K): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / K
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): net.forward_dev(x, out)
    torch.cuda.synchronize()
    ref = out.clone()
    t_plain = timed(lambda: net.forward_dev(x, out))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        net.forward_dev(x, out)
    out.zero_(); g.replay(); torch.cuda.synchronize()
    same = bool((out == ref).all())
    t_graph = timed(g.replay)
for name, dt in (("launches", t_plain), ("graph replay", t_graph)):
    print("%s batch %d %-13s %.3f ms  MFMA(f32) util %.3f" % (which, B, name, dt * 1e3, 2.0 * mac * B / dt / 157.3e12))
print("graph output identical:", same)
