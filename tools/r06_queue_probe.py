"""How many other streams a process has used before a dlib handle makes its part streams, and what that does to the forward of one batch of 256
(two half-batch chains on the handle's own streams): usage r06_queue_probe.py [K streams used before]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from columbiaimagesearch_amd.featurizer.synthetic import dlib_weights
from columbiaimagesearch_amd.featurizer import DLibFaceNet
K = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda", 0)
pre = [torch.cuda.Stream() for _ in range(K)]
a = torch.zeros(1024, device=dev)
for s in pre:
    with torch.cuda.stream(s):
        a.add_(1.0)
torch.cuda.synchronize()
net = DLibFaceNet(dlib_weights(0))
x = (torch.rand(256, 150, 150, 3, device=dev) * 255).contiguous(); out = torch.empty(256, 128, device=dev)
def run(k, st):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(k):
        with torch.cuda.stream(st):
            net.forward_dev(x, out)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / k
side = torch.cuda.Stream()
for name, st in (("null stream", torch.cuda.current_stream()), ("side stream", side)):
    run(3, st)
    print("GPU_MAX_HW_QUEUES=%s, %2d streams used before: dlib 256 on the %s %.3f ms" % (os.environ.get("GPU_MAX_HW_QUEUES"), K, name, min(run(16, st) for _ in range(3)) * 1e3))
