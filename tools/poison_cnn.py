"""Which workspace does a forward read before writing it?  CIS_CNN_POISON fills one workspace at a time with NaN bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from columbiaimagesearch_amd.featurizer.synthetic import sentibank_weights, dlib_weights
from columbiaimagesearch_amd.featurizer import SentiBankNet
from columbiaimagesearch_amd.featurizer.dlibhip_featurizer import DLibFaceNet
rs = np.random.RandomState(0)
for name, net, shape in (("sentibank", SentiBankNet(sentibank_weights(0)), (3, 227, 227)), ("dlib", DLibFaceNet(dlib_weights(0)), (150, 150, 3))):
    for n in (1, 2, 5, 33, 256):
        x = torch.from_numpy((rs.randn(n, *shape) * 50).astype(np.float32)).cuda()
        os.environ.pop("CIS_CNN_POISON", None)
        ref = net.forward_dev(x).clone()
        ref = net.forward_dev(x).clone()
        for bit in range(5):
            os.environ["CIS_CNN_POISON"] = str(1 << bit)
            out = net.forward_dev(x).clone()
            out = net.forward_dev(x).clone()
            ok = torch.equal(out, ref)
            if not ok:
                bad = ~(out == ref)
                print(name, "n", n, "poison bit", bit, "MISMATCH: rows", bad.any(dim=1).nonzero().flatten().tolist()[:8], "nan", int(torch.isnan(out).sum()), "of", out.numel())
        os.environ.pop("CIS_CNN_POISON", None)
    print(name, "done")
