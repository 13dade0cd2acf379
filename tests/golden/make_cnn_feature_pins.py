#!/usr/bin/env python3
"""sha1 of the descriptors of the seeded synthetic networks on seeded batches (batch sizes that exercise full and partial tiles) -- run on
an MI355X:  python tests/golden/make_cnn_feature_pins.py > tests/golden/cnn_feature_pins.json
The pins say WHICH float32 summation order the forward kernels implement, nothing about any reference (the CNN oracles are unpinned)."""
import hashlib
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def feature_sha1(arch):
    h = hashlib.sha1()
    if arch == "sentibank":
        from oracle import cnn_oracle as C
        from columbiaimagesearch_amd.featurizer import SentiBankNet
        net, mk = SentiBankNet(C.synthetic_weights(0)), lambda n: C.synthetic_images(n, seed=11)
    else:
        from oracle import dlib_oracle as D
        from columbiaimagesearch_amd.featurizer import DLibFaceNet
        net, mk = DLibFaceNet(D.synthetic_weights(1)), lambda n: D.synthetic_chips(n, seed=11)
    for n in (1, 5, 64):
        f = net.forward(np.ascontiguousarray(mk(n), dtype=np.float32))
        h.update(np.ascontiguousarray(f).tobytes())
    net.close()
    return h.hexdigest()


if __name__ == "__main__":
    from columbiaimagesearch_amd import featurizer as F
    print(json.dumps({"feature_kernel_version": F.FEATURE_KERNEL_VERSION, "sentibank": feature_sha1("sentibank"), "dlib": feature_sha1("dlib")}, indent=1))
