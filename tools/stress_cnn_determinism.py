"""Stress: the same forward many times, alone and while another stream keeps the chip busy; every result must equal the first bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from columbiaimagesearch_amd.featurizer.synthetic import sentibank_weights, dlib_weights
from columbiaimagesearch_amd.featurizer import SentiBankNet
from columbiaimagesearch_amd.featurizer.dlibhip_featurizer import DLibFaceNet
rs = np.random.RandomState(0)
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device="cuda")
def busy(k):
    with torch.cuda.stream(side):
        for _ in range(k):
            a.mm(a)
for name, net, shape in (("sentibank", SentiBankNet(sentibank_weights(0)), (3, 227, 227)), ("dlib", DLibFaceNet(dlib_weights(0)), (150, 150, 3))):
    for n in (1, 2, 7, 33):
        x = torch.from_numpy((rs.randn(n, *shape) * 50).astype(np.float32)).cuda()
        ref = net.forward_dev(x).clone()
        torch.cuda.synchronize()
        bad = 0
        iters = 300 if n <= 7 else 100
        for it in range(iters):
            if it % 2:
                busy(2)
            out = net.forward_dev(x)
            if not torch.equal(out, ref):
                bad += 1
        # the host-buffer entry point as well (its own staging buffers)
        xh = x.cpu().numpy()
        h0 = net.forward(xh)
        for it in range(50):
            if it % 2:
                busy(2)
            if not np.array_equal(net.forward(xh), h0):
                bad += 1
        torch.cuda.synchronize()
        print(name, "batch", n, "mismatching runs:", bad, "host == dev:", np.array_equal(h0, ref.cpu().numpy()))
