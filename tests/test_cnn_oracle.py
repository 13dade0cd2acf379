"""The DeepSentibank oracle is pinned by two independent restatements of the prototxt (caffe itself and the
trained weights are not available: parity with the reference is unpinned)."""
import numpy as np

from oracle import cnn_oracle as C


def test_layer_shapes_follow_the_prototxt():
    shapes = dict((n, (w, b)) for n, w, b in C.layer_shapes())
    assert shapes["conv1"][0] == (96, 3, 11, 11) and shapes["conv2"][0] == (256, 48, 5, 5)  # group 2
    assert shapes["conv4"][0] == (384, 192, 3, 3) and shapes["conv5"][0] == (256, 192, 3, 3)
    assert shapes["fc6"][0] == (4096, 9216) and shapes["fc7"][0] == (4096, 4096)


def test_torch_and_numpy_restatements_agree():
    w = C.synthetic_weights(0)
    x = C.synthetic_images(1, seed=7)
    a = C.forward_torch(x, w)
    b = C.forward_numpy(x, w)
    assert a.shape == (1, 4096)
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-5 * np.abs(b).max())
    assert (a >= 0).all() and (a > 0).mean() > 0.2  # fc7 is read after the in-place ReLU


def test_pooling_is_ceil_mode_and_lrn_divides_alpha_by_n():
    x = np.arange(2 * 1 * 13 * 13, dtype=np.float64).reshape(2, 1, 13, 13)
    p = C._pool_numpy(x)
    assert p.shape == (2, 1, 6, 6) and p[0, 0, 5, 5] == x[0, 0, 12, 12]  # last window is clipped, not dropped
    y = np.ones((1, 7, 1, 1))
    out = C._lrn_numpy(y)
    assert np.isclose(out[0, 3, 0, 0], (1 + 1e-4 / 5 * 5) ** -0.75) and np.isclose(out[0, 0, 0, 0], (1 + 1e-4 / 5 * 3) ** -0.75)


def test_dlib_oracle_structure_and_shapes():
    """dlib face network restatement: 29 convolutions, 117 tensors, dlib's size rules give 150 -> 72 -> 35 -> ... -> 3."""
    from oracle import dlib_oracle as D
    plan = D.block_plan()
    assert len(plan) == 14 and sum(1 for p in plan if p[2]) == 4
    assert len(D.tensor_names()) == 117
    w = D.synthetic_weights(0)
    assert w["conv0_w"].shape == (32, 3, 7, 7) and w["fc_w"].shape == (128, 256)
    out = D.forward_torch(D.synthetic_chips(2, seed=3), w)
    assert out.shape == (2, 128) and np.isfinite(out).all() and np.abs(out).max() > 0
    assert D.mac_per_face() == 270854144
