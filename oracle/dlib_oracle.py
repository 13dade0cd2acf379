"""CPU restatement of dlib's face-descriptor network (dlib_face_recognition_resnet_model_v1).  TEST INFRASTRUCTURE ONLY.

The reference calls third-party dlib: ``DLibFeaturizer.featurize`` cufacesearch/cufacesearch/featurizer/dlib_featurizer.py:86-105
-> ``face_recognition_model_v1.compute_face_descriptor(img, shape)`` (:105).  Neither the network definition nor the
weights are in the reference tree and dlib is not importable here => **parity unpinned**.  The architecture below is
dlib's published ``anet_type`` (examples/dnn_face_recognition_ex.cpp), restated from its layer semantics:

    input_rgb_image_sized<150>: (pixel - (122.782, 117.001, 104.298)) / 256, RGB
    con<32,7,7,2,2> -> affine -> relu -> max_pool<3,3,2,2>
    alevel4 = 3 x ares<32>;  alevel3 = ares_down<64> + 3 x ares<64>;  alevel2 = ares_down<128> + 2 x ares<128>
    alevel1 = ares_down<256> + 2 x ares<256>;  alevel0 = ares_down<256>
    avg_pool_everything -> fc_no_bias<128>
    block<N,stride> = affine(con<N,3,3,1,1>(relu(affine(con<N,3,3,stride,stride>(x)))))
    ares = relu(block(x) + x);  ares_down = relu(block_stride2(x) + avg_pool<2,2,2,2>(x))

dlib conventions restated: ``con`` / pooling layers pad nr/2 when the stride is 1 and 0 otherwise; output size
1 + (n + 2*pad - k) / stride (floor); ``add_prev`` adds tensors of different shape by zero-padding the smaller one
(spatially at the bottom/right and in channels); ``affine`` is a per-channel scale and shift (inference form of
batch-norm).  The reference keeps the descriptor as 128 float64 values (featsio.py:34-36).
"""
import numpy as np

MEAN_RGB = (122.782, 117.001, 104.298)
INPUT_HW = 150
FEAT_DIM = 128
# (channels, number of blocks, first block is a down block)
LEVELS = [(32, 3, False), (64, 4, True), (128, 3, True), (256, 3, True), (256, 1, True)]


def block_plan():
    """[(in_channels, out_channels, down)] for the 14 residual blocks in forward order."""
    plan, c = [], 32
    for n, count, down in LEVELS:
        for b in range(count):
            plan.append((c, n, down and b == 0))
            c = n
    return plan


def tensor_names():
    names = ["conv0_w", "conv0_b", "aff0_g", "aff0_b"]
    for i in range(len(block_plan())):
        for half in ("a", "b"):
            names += ["b%d%s_w" % (i, half), "b%d%s_b" % (i, half), "b%d%s_g" % (i, half), "b%d%s_beta" % (i, half)]
    return names + ["fc_w"]


def synthetic_weights(seed=0):
    """Seeded weights: conv OIHW, per-channel affine gamma/beta, fc [128][256] (float32); generator in the package."""
    from columbiaimagesearch_amd.featurizer.synthetic import dlib_weights
    return dlib_weights(seed)


def synthetic_chips(n, seed=1):
    """n aligned face chips, uint8 RGB [n,150,150,3] (what get_face_chip would hand to the network)."""
    from columbiaimagesearch_amd.featurizer.synthetic import dlib_chips
    return dlib_chips(n, seed)


def _pad_to(t, shape):
    import torch.nn.functional as F
    return F.pad(t, (0, shape[3] - t.shape[3], 0, shape[2] - t.shape[2], 0, shape[1] - t.shape[1]))


def forward_torch(chips, w):
    """chips uint8/float [n,150,150,3] RGB -> [n,128] float32."""
    import torch
    import torch.nn.functional as F
    T = lambda k: torch.from_numpy(w[k])
    x = torch.from_numpy(np.ascontiguousarray(chips)).float().permute(0, 3, 1, 2)
    x = (x - torch.tensor(MEAN_RGB).view(1, 3, 1, 1)) / 256.0
    aff = lambda t, g, b: t * T(g).view(1, -1, 1, 1) + T(b).view(1, -1, 1, 1)
    with torch.no_grad():
        x = F.relu(aff(F.conv2d(x, T("conv0_w"), T("conv0_b"), stride=2, padding=0), "aff0_g", "aff0_b"))
        x = F.max_pool2d(x, 3, 2, padding=0)
        for i, (cin, cout, down) in enumerate(block_plan()):
            s, p = (2, 0) if down else (1, 1)
            y = F.relu(aff(F.conv2d(x, T("b%da_w" % i), T("b%da_b" % i), stride=s, padding=p), "b%da_g" % i, "b%da_beta" % i))
            y = aff(F.conv2d(y, T("b%db_w" % i), T("b%db_b" % i), stride=1, padding=1), "b%db_g" % i, "b%db_beta" % i)
            skip = F.avg_pool2d(x, 2, 2) if down else x
            shape = [y.shape[0]] + [max(a, b) for a, b in zip(y.shape[1:], skip.shape[1:])]
            x = F.relu(_pad_to(y, shape) + _pad_to(skip, shape))
        x = x.mean(dim=(2, 3))
        return (x @ T("fc_w").t()).numpy()


def mac_per_face():
    """multiply-accumulates of the 29 convolutions + fc for one 150x150 chip."""
    total, hw = 0, (INPUT_HW - 7) // 2 + 1
    total += hw * hw * 32 * 7 * 7 * 3
    hw = (hw - 3) // 2 + 1
    for cin, cout, down in block_plan():
        if down:
            oh = (hw - 3) // 2 + 1
            total += oh * oh * cout * 9 * cin + oh * oh * cout * 9 * cout
            hw = max(oh, (hw - 2) // 2 + 1)
        else:
            total += 2 * hw * hw * cout * 9 * cin
    return total + 256 * 128
