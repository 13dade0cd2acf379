"""The image side of dlib's compute_face_descriptor(img, shape) (cufacesearch/cufacesearch/featurizer/dlib_featurizer.py:103-105):
get_face_chip_details + extract_image_chip restated (columbiaimagesearch_amd/featurizer/face_chip.py, csrc/face_chip.hip).  dlib is not
installed: parity with dlib itself is UNPINNED; these tests check the pieces against independent restatements and invariants."""
import numpy as np
import pytest

from conftest import has_gpu

gpu = pytest.mark.gpu


def _synthetic_landmarks(scale, angle, tx, ty, noise=0.0, seed=0):
    """68 landmarks = dlib's mean face under a known similarity (eyebrows / jaw filled with plausible points)."""
    from columbiaimagesearch_amd.featurizer import face_chip as F
    rs = np.random.RandomState(seed)
    lm = np.zeros((68, 2))
    lm[:17, 0] = np.linspace(0.0, 1.0, 17)
    lm[:17, 1] = 0.3 + 0.7 * np.sin(np.linspace(0, np.pi, 17))
    lm[17:, 0], lm[17:, 1] = F.MEAN_FACE_X, F.MEAN_FACE_Y
    c, s = np.cos(angle), np.sin(angle)
    R = np.array([[c, -s], [s, c]])
    out = (scale * (R @ lm.T)).T + np.array([tx, ty])
    return out + noise * rs.randn(68, 2)


def test_similarity_transform_recovers_a_known_similarity_and_is_the_least_squares_optimum():
    from columbiaimagesearch_amd.featurizer import face_chip as F
    rs = np.random.RandomState(1)
    f = rs.rand(30, 2) * 100
    c, s = np.cos(0.4), np.sin(0.4)
    M = 1.7 * np.array([[c, -s], [s, c]])
    t = f @ M.T + np.array([12.0, -7.0])
    m, b = F.find_similarity_transform(f, t)
    np.testing.assert_allclose(m, M, atol=1e-10)
    np.testing.assert_allclose(b, [12.0, -7.0], atol=1e-9)
    # noisy targets: no nearby similarity (scale, angle, shift perturbed) has a smaller squared error
    t2 = t + rs.randn(30, 2)
    m2, b2 = F.find_similarity_transform(f, t2)
    err = lambda mm, bb: ((f @ mm.T + bb - t2) ** 2).sum()
    e0 = err(m2, b2)
    sc, an = np.hypot(m2[0, 0], m2[1, 0]), np.arctan2(m2[1, 0], m2[0, 0])
    for ds, da, dx in [(1e-3, 0, 0), (-1e-3, 0, 0), (0, 1e-3, 0), (0, -1e-3, 0), (0, 0, 0.05), (0, 0, -0.05)]:
        cc, ss = np.cos(an + da), np.sin(an + da)
        assert err((sc + ds) * np.array([[cc, -ss], [ss, cc]]), b2 + dx) >= e0 - 1e-9


def test_chip_details_of_the_mean_face_under_a_similarity():
    """Landmarks = the mean face scaled by S, rotated by A, shifted: the chip rectangle is the padded mean-face square under the
    same similarity -- side 150 * (S * (2 * 0.25 + 1) / 150) pixels, angle A, centred on the image of the square's centre."""
    from columbiaimagesearch_amd.featurizer import face_chip as F
    for S, A, tx, ty in [(200.0, 0.0, 50.0, 80.0), (90.0, 0.3, 400.0, 120.0), (333.0, -0.7, 10.0, 10.0)]:
        d = F.chip_details_from_landmarks(_synthetic_landmarks(S, A, tx, ty))
        assert abs(d["angle"] - A) < 1e-9
        side = S * 1.5 / 150.0 * 150.0            # chip pixel = S * 1.5 / 150 image pixels
        l, t, r, b = d["rect"]
        assert abs((r - l + 1) - side) < 1e-6 and abs((b - t + 1) - side) < 1e-6
        c, s = np.cos(A), np.sin(A)
        centre = S * (np.array([[c, -s], [s, c]]) @ np.array([0.5, 0.5])) + np.array([tx, ty])  # mean-face square [-.25, 1.25]^2 has centre (.5, .5)
        np.testing.assert_allclose([(l + r) / 2, (t + b) / 2], centre, atol=1e-6)


def test_chip_maps_pick_the_pyramid_level_dlib_would():
    from columbiaimagesearch_amd.featurizer import face_chip as F
    small = F.chip_details_from_landmarks(_synthetic_landmarks(120.0, 0.0, 300, 300))    # 180-pixel region -> 150: no pyramid
    big = F.chip_details_from_landmarks(_synthetic_landmarks(520.0, 0.2, 900, 700))      # 780-pixel region: two levels down
    bb, n_levels, maps = F.chip_maps([small, big], 2000, 2400)
    assert maps[0][0] == -1 and maps[1][0] == 1 and n_levels == 2
    # the map sends the chip's corners onto the rotated rectangle's corners (level coordinates)
    lv, m6 = maps[0]
    tl = np.array([m6[0], m6[3]])
    tr = np.array([m6[0] + 149 * m6[1], m6[3] + 149 * m6[4]])
    assert abs(np.linalg.norm(tr - tl) - (small["rect"][2] - small["rect"][0])) < 1e-6


@gpu
def test_chips_on_the_gpu_equal_the_numpy_restatement_and_feed_the_network():
    import torch
    from columbiaimagesearch_amd.featurizer import face_chip as F
    from columbiaimagesearch_amd.featurizer import DLibFaceNet
    from columbiaimagesearch_amd.featurizer.synthetic import dlib_weights
    rs = np.random.RandomState(3)
    yy, xx = np.mgrid[0:900, 0:1200]
    img = np.stack([(xx * 0.2 + yy * 0.1) % 256, (xx * 0.05 + 40 * np.sin(yy / 30.0)) % 256, rs.randint(0, 256, size=xx.shape)], axis=-1).astype(np.uint8)
    shapes = [_synthetic_landmarks(110.0, 0.15, 200, 150, noise=1.0, seed=1), _synthetic_landmarks(480.0, -0.25, 500, 100, noise=2.0, seed=2),
              _synthetic_landmarks(150.0, 0.0, 1100, 700, noise=1.0, seed=3)]   # the last one leaves the image: black pixels
    chips = F.face_chips(img, shapes).cpu().numpy()
    details = [F.chip_details_from_landmarks(s) for s in shapes]
    bb, n_levels, maps = F.chip_maps(details, 900, 1200)
    sub = img[bb[1]:bb[3] + 1, bb[0]:bb[2] + 1]
    levels = {-1: sub}
    cur = sub
    for k in range(n_levels):
        cur = F.pyramid_down2_numpy(cur)
        levels[k] = cur
    assert n_levels >= 1 and any(lv >= 0 for lv, _ in maps)
    for i, (lv, m6) in enumerate(maps):
        want = F.extract_chips_numpy(levels[lv], m6)
        # float64 on both sides; a pixel whose value lands within rounding of an integer may differ by one grey level
        diff = np.abs(chips[i] - want)
        assert diff.max() <= 1.0 and (diff > 0).mean() < 1e-3, (i, diff.max(), (diff > 0).mean())
    assert (chips[2] == 0).any() and (chips[0] > 0).any()
    # a face whose landmarks sit exactly on the mean face at chip scale and no rotation: the chip is the image crop itself.  (dlib
    # centres the extraction rectangle on the image of chip point (75, 75) while the chip's pixel centre is (74.5, 74.5): its
    # chips sit half a pixel up-left of the landmarks' frame -- restated as it is, hence the -0.5 here.)
    S = 150.0 / 1.5
    lm = _synthetic_landmarks(S, 0.0, 300.0 - 0.5 + 0.25 * S, 200.0 - 0.5 + 0.25 * S)
    c0 = F.face_chips(img, [lm]).cpu().numpy()[0]
    # (assign_pixel truncates: where the float64 interpolation of an exactly aligned sample lands a hair under the integer, the chip is
    # one grey level below the crop -- never above, never more)
    diff = img[200:350, 300:450].astype(np.float32) - c0
    assert diff.min() >= 0.0 and diff.max() <= 1.0, (diff.min(), diff.max())
    # ... and the chips feed the network like any aligned chips do
    net = DLibFaceNet(dlib_weights(0))
    a = net.forward_dev(F.face_chips(img, shapes[:2])).cpu().numpy()
    b = net.forward(chips[:2])
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-5 * max(1.0, np.abs(b).max()))
