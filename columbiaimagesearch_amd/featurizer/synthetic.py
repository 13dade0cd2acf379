"""Seeded synthetic weights and inputs of the two descriptor networks (the trained weights are not in the reference tree
and there is no network access): what bench.py's CNN leg, the tools and the tests feed to the HIP forwards.  Data
generators only -- the CPU restatements that CHECK the forwards live in oracle/ (test infrastructure)."""
import numpy as np

# DeepSentibank (cufacesearch/cufacesearch/featurizer/data/pycaffe_sentibank.prototxt:7-16,:47-58,:88-98,:105-116,:123-134,
# :153-160,:176-183): (name, out_channels, kernel, stride, pad, groups) and the two 4096-wide inner products
SENTIBANK_CONVS = [("conv1", 96, 11, 4, 0, 1), ("conv2", 256, 5, 1, 2, 2), ("conv3", 384, 3, 1, 1, 1),
                   ("conv4", 384, 3, 1, 1, 2), ("conv5", 256, 3, 1, 1, 2)]
SENTIBANK_FCS = [("fc6", 4096), ("fc7", 4096)]
SENTIBANK_POOL_AFTER = {"conv1", "conv2", "conv5"}
SENTIBANK_INPUT_HW = 227

# dlib face network (anet_type): (channels, number of blocks, first block is a down block)
DLIB_LEVELS = [(32, 3, False), (64, 4, True), (128, 3, True), (256, 3, True), (256, 1, True)]
DLIB_INPUT_HW = 150


def sentibank_layer_shapes():
    """[(name, weight shape (caffe OIHW / [out,in]), bias shape)] in forward order."""
    out = []
    c, hw = 3, SENTIBANK_INPUT_HW
    for name, oc, k, s, p, g in SENTIBANK_CONVS:
        out.append((name, (oc, c // g, k, k), (oc,)))
        hw = (hw + 2 * p - k) // s + 1
        c = oc
        if name in SENTIBANK_POOL_AFTER:
            hw = int(np.ceil((hw - 3) / 2.0)) + 1
    fin = c * hw * hw
    for name, oc in SENTIBANK_FCS:
        out.append((name, (oc, fin), (oc,)))
        fin = oc
    return out


def sentibank_weights(seed=0):
    """Seeded He-scaled weights in caffe layout (float32): {name_w, name_b}."""
    rs = np.random.RandomState(seed)
    w = {}
    for name, ws, bs in sentibank_layer_shapes():
        fan_in = int(np.prod(ws[1:]))
        w[name + "_w"] = (rs.randn(*ws) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        w[name + "_b"] = (rs.randn(*bs) * 0.05).astype(np.float32)
    return w


def sentibank_images(n, seed=1):
    """Mean-subtracted-pixel-like inputs, NCHW float32 (what preprocess_img hands to the net, sbpycaffe_img_featurizer.py:113-134)."""
    return (np.random.RandomState(seed).randn(n, 3, SENTIBANK_INPUT_HW, SENTIBANK_INPUT_HW) * 50.0).astype(np.float32)


def dlib_block_plan():
    """[(in_channels, out_channels, down)] for the 14 residual blocks in forward order."""
    plan, c = [], 32
    for n, count, down in DLIB_LEVELS:
        for b in range(count):
            plan.append((c, n, down and b == 0))
            c = n
    return plan


def dlib_weights(seed=0):
    """Seeded weights: conv OIHW, per-channel affine gamma/beta, fc [128][256] (float32)."""
    rs = np.random.RandomState(seed)
    w = {}

    def conv(name, oc, ic, k):
        w[name + "_w"] = (rs.randn(oc, ic, k, k) * np.sqrt(2.0 / (ic * k * k))).astype(np.float32)
        w[name + "_b"] = (rs.randn(oc) * 0.02).astype(np.float32)

    def affine(g, b, c):
        w[g] = (1.0 + 0.1 * rs.randn(c)).astype(np.float32)
        w[b] = (0.05 * rs.randn(c)).astype(np.float32)

    conv("conv0", 32, 3, 7)
    affine("aff0_g", "aff0_b", 32)
    for i, (cin, cout, down) in enumerate(dlib_block_plan()):
        conv("b%da" % i, cout, cin, 3)
        affine("b%da_g" % i, "b%da_beta" % i, cout)
        conv("b%db" % i, cout, cout, 3)
        affine("b%db_g" % i, "b%db_beta" % i, cout)
    w["fc_w"] = (rs.randn(128, 256) * np.sqrt(1.0 / 256)).astype(np.float32)
    return w


def dlib_chips(n, seed=1):
    """n aligned face chips, uint8 RGB [n,150,150,3] (what get_face_chip would hand to the network)."""
    return np.random.RandomState(seed).randint(0, 256, size=(n, DLIB_INPUT_HW, DLIB_INPUT_HW, 3)).astype(np.uint8)
