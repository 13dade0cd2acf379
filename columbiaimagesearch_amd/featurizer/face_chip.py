"""Aligned 150x150 face chips from 68 landmarks: what dlib's ``compute_face_descriptor(img, shape)`` does to the image before
its network runs (cufacesearch/cufacesearch/featurizer/dlib_featurizer.py:103-105 -- the shape comes from
``dlib.shape_predictor(pred_path)(img, rect)`` :103, which stays the caller's).

Restated from dlib's published sources (third-party, version unpinned in cufacesearch/cufacesearch/requirements.txt:3, not
installed here -- parity is UNPINNED; the tests check the pieces against independent numpy restatements):

* ``chip_details_from_landmarks`` = ``get_face_chip_details(shape, size=150, padding=0.25)`` (image_transforms/interpolation.h):
  the landmarks 17..67 without the eyebrows (17..26) and the lower lip (55..59, 65..67) against dlib's mean face shape, scaled by
  ``(padding + mean) / (2 padding + 1) * size``; ``find_similarity_transform`` (Umeyama's least-squares similarity, geometry/
  point_transforms.h) from chip to image coordinates; angle and scale read off the transform of (1, 0); the extraction rectangle
  centred on the transformed chip centre (``centered_drect``: width - 1, height - 1 between the outer pixels).
* ``chip_maps`` = the per-chip part of ``extract_image_chips``: the pyramid level whose rectangle is no more than the chip's size
  (``pyramid_down<2>``: ``point_down(p) = p / 2 - (1.25, 0.75)``), the bounding box of the rotated rectangles grown by 2^(depth+1),
  and the affine map chip pixel -> level pixel from three corner correspondences (``find_affine_transform``).
* the pixels: csrc/face_chip.hip (``cis_pyramid_down2_dev``, ``cis_extract_chips_dev``: dlib's interpolate_bilinear).
"""
import numpy as np

from .. import _lib

CHIP_SIZE = 150
CHIP_PADDING = 0.25

# dlib image_transforms/interpolation.h, get_face_chip_details: landmarks 17 .. 67 of the 68-point model
MEAN_FACE_X = np.array([
    0.000213256, 0.0752622, 0.18113, 0.29077, 0.393397, 0.586856, 0.689483, 0.799124,
    0.904991, 0.98004, 0.490127, 0.490127, 0.490127, 0.490127, 0.36688, 0.426036,
    0.490127, 0.554217, 0.613373, 0.121737, 0.187122, 0.265825, 0.334606, 0.260918,
    0.182743, 0.645647, 0.714428, 0.793132, 0.858516, 0.79751, 0.719335, 0.254149,
    0.340985, 0.428858, 0.490127, 0.551395, 0.639268, 0.726104, 0.642159, 0.556721,
    0.490127, 0.423532, 0.338094, 0.290379, 0.428096, 0.490127, 0.552157, 0.689874,
    0.553364, 0.490127, 0.42689], dtype=np.float64)
MEAN_FACE_Y = np.array([
    0.106454, 0.038915, 0.0187482, 0.0344891, 0.0773906, 0.0773906, 0.0344891,
    0.0187482, 0.038915, 0.106454, 0.203352, 0.307009, 0.409805, 0.515625, 0.587326,
    0.609345, 0.628106, 0.609345, 0.587326, 0.216423, 0.178758, 0.179852, 0.231733,
    0.245099, 0.244077, 0.231733, 0.179852, 0.178758, 0.216423, 0.244077, 0.245099,
    0.780233, 0.745405, 0.727388, 0.742578, 0.727388, 0.745405, 0.780233, 0.864805,
    0.902192, 0.909281, 0.902192, 0.864805, 0.784792, 0.778746, 0.785343, 0.778746,
    0.784792, 0.824182, 0.831803, 0.824182], dtype=np.float64)
assert MEAN_FACE_X.shape == (51,) and MEAN_FACE_Y.shape == (51,)

_USED = np.array([i for i in range(17, 68) if not ((55 <= i <= 59) or (65 <= i <= 67) or (17 <= i <= 26))])


def find_similarity_transform(from_pts, to_pts):
    """dlib's find_similarity_transform (Umeyama 1991): (m [2,2], b [2]) with to ~ m @ from + b, m = scale * rotation."""
    f = np.asarray(from_pts, dtype=np.float64)
    t = np.asarray(to_pts, dtype=np.float64)
    n = f.shape[0]
    mean_from, mean_to = f.sum(axis=0) / n, t.sum(axis=0) / n
    sigma_from = ((f - mean_from) ** 2).sum() / n
    cov = (t - mean_to).T @ (f - mean_from) / n
    u, d, vt = np.linalg.svd(cov)
    s = np.eye(2)
    det_cov = np.linalg.det(cov)
    if det_cov < 0 or (det_cov == 0 and np.linalg.det(u) * np.linalg.det(vt) < 0):
        if d[1] < d[0]:
            s[1, 1] = -1
        else:
            s[0, 0] = -1
    r = u @ s @ vt
    c = 1.0
    if sigma_from != 0:
        c = 1.0 / sigma_from * np.trace(np.diag(d) @ s)
    return c * r, mean_to - c * (r @ mean_from)


def chip_details_from_landmarks(landmarks, size=CHIP_SIZE, padding=CHIP_PADDING):
    """landmarks [68, 2] (x, y) image coordinates -> dict(rect=(left, top, right, bottom) float64, angle, rows, cols):
    dlib's get_face_chip_details + the chip_details constructor."""
    lm = np.asarray(landmarks, dtype=np.float64)
    if lm.shape != (68, 2):
        raise ValueError("68 landmarks (x, y) expected, got %r" % (lm.shape,))
    fx = (padding + MEAN_FACE_X[_USED - 17]) / (2 * padding + 1) * size
    fy = (padding + MEAN_FACE_Y[_USED - 17]) / (2 * padding + 1) * size
    m, b = find_similarity_transform(np.stack([fx, fy], axis=1), lm[_USED])
    p = m @ np.array([1.0, 0.0])
    angle = float(np.arctan2(p[1], p[0]))
    scale = float(np.sqrt(p[0] * p[0] + p[1] * p[1]))
    centre = m @ (np.array([size, size], dtype=np.float64) / 2.0) + b
    w = size * scale - 1.0  # centered_drect: width--, height--
    h = size * scale - 1.0
    return {"rect": (centre[0] - w / 2, centre[1] - h / 2, centre[0] + w / 2, centre[1] + h / 2), "angle": angle,
            "rows": int(size), "cols": int(size)}


def _rect_area(r):
    w, h = r[2] - r[0] + 1.0, r[3] - r[1] + 1.0   # drectangle::area(): width() * height(), width = right - left + 1
    return 0.0 if (w <= 0 or h <= 0) else w * h


def _rect_down(r):
    """pyramid_down<2>::rect_down: both corners through point_down(p) = p / 2 - (1.25, 0.75)"""
    return (r[0] / 2.0 - 1.25, r[1] / 2.0 - 0.75, r[2] / 2.0 - 1.25, r[3] / 2.0 - 0.75)


def _rotate(centre, p, angle):
    c, s = np.cos(angle), np.sin(angle)
    d = np.asarray(p, dtype=np.float64) - centre
    return np.array([c * d[0] - s * d[1], s * d[0] + c * d[1]]) + centre


def chip_maps(details, img_rows, img_cols):
    """The geometry of extract_image_chips for ALL chips of one image: (bounding_box (l, t, r, b) ints or None, n_levels,
    [(level, map6)]) -- level -1 = the sub-image `bounding_box` itself, level k = its k+1-th pyramid_down<2>; map6 = the affine map
    chip (c, r) -> pixel of that level, source = (m0 + m1 c + m2 r, m3 + m4 c + m5 r)."""
    if not details:
        return None, 0, []
    max_depth = 0
    bb = None
    for d in details:
        chip_size = float(d["rows"] * d["cols"])
        depth, grow = 0, 2.0
        rect = _rect_down(d["rect"])
        while _rect_area(rect) > chip_size:
            rect = _rect_down(rect)
            depth += 1
            grow *= 2
        r0 = d["rect"]
        centre = np.array([(r0[0] + r0[2]) / 2.0, (r0[1] + r0[3]) / 2.0])
        pts = [_rotate(centre, c, d["angle"]) for c in ((r0[0], r0[1]), (r0[2], r0[1]), (r0[0], r0[3]), (r0[2], r0[3]))]
        # rectangle += point: the smallest integer rectangle that holds the (rounded) points; then grown and cut to the image
        xs = [int(np.floor(p[0] + 0.5)) for p in pts]
        ys = [int(np.floor(p[1] + 0.5)) for p in pts]
        g = int(grow)
        l, t, r, b = min(xs) - g, min(ys) - g, max(xs) + g, max(ys) + g
        l, t, r, b = max(l, 0), max(t, 0), min(r, img_cols - 1), min(b, img_rows - 1)
        if l <= r and t <= b:
            bb = (l, t, r, b) if bb is None else (min(bb[0], l), min(bb[1], t), max(bb[2], r), max(bb[3], b))
        max_depth = max(max_depth, depth)
    if bb is None:
        bb = (0, 0, -1, -1)
    out = []
    for d in details:
        chip_size = float(d["rows"] * d["cols"])
        level = -1
        rect = tuple(v - (bb[0] if i % 2 == 0 else bb[1]) for i, v in enumerate(d["rect"]))  # translate_rect(rect, -bounding_box.tl_corner())
        while _rect_area(_rect_down(rect)) > chip_size:
            level += 1
            rect = _rect_down(rect)
        centre = np.array([(rect[0] + rect[2]) / 2.0, (rect[1] + rect[3]) / 2.0])
        tl = _rotate(centre, (rect[0], rect[1]), d["angle"])
        tr = _rotate(centre, (rect[2], rect[1]), d["angle"])
        bl = _rotate(centre, (rect[0], rect[3]), d["angle"])
        cols, rows = d["cols"], d["rows"]
        ex = (tr - tl) / float(cols - 1)   # find_affine_transform of (0,0)->tl, (cols-1,0)->tr, (0,rows-1)->bl
        ey = (bl - tl) / float(rows - 1)
        out.append((level, np.array([tl[0], ex[0], ey[0], tl[1], ex[1], ey[1]], dtype=np.float64)))
    return bb, max_depth, out


def extract_chips_numpy(level_img, map6, size=CHIP_SIZE):
    """numpy restatement of csrc/face_chip.hip:k_extract_chips (dlib's interpolate_bilinear + assign_pixel): test checker."""
    nr, nc = level_img.shape[:2]
    r, c = np.meshgrid(np.arange(size, dtype=np.float64), np.arange(size, dtype=np.float64), indexing="ij")
    x = map6[0] + map6[1] * c + map6[2] * r
    y = map6[3] + map6[4] * c + map6[5] * r
    left, top = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)
    ok = (left >= 0) & (top >= 0) & (left + 1 < nc) & (top + 1 < nr)
    l2, t2 = np.clip(left, 0, max(nc - 2, 0)), np.clip(top, 0, max(nr - 2, 0))
    lr, tb = (x - left)[..., None], (y - top)[..., None]
    im = level_img.astype(np.float64)
    v = (1 - tb) * ((1 - lr) * im[t2, l2] + lr * im[t2, l2 + 1]) + tb * ((1 - lr) * im[t2 + 1, l2] + lr * im[t2 + 1, l2 + 1])
    out = np.floor(v.clip(0, 255))  # assign_pixel(unsigned char&, double): clamp, then static_cast (truncation, no + 0.5)
    out[~ok] = 0
    return out.astype(np.float32)


def pyramid_down2_numpy(img):
    """numpy restatement of dlib's pyramid_down<2> on an RGB uint8 image (csrc/face_chip.hip:k_pyr_down2_*): test checker."""
    a = img.astype(np.int64)
    nr, nc = a.shape[:2]
    tc, orows = (nc - 3) // 2, (nr - 3) // 2
    cols = 2 * np.arange(tc)
    tmp = a[:, cols] + 4 * a[:, cols + 1] + 6 * a[:, cols + 2] + 4 * a[:, cols + 3] + a[:, cols + 4]
    rows = 2 * np.arange(orows) + 2
    out = tmp[rows - 2] + 4 * tmp[rows - 1] + 6 * tmp[rows] + 4 * tmp[rows + 1] + tmp[rows + 2]
    return (out // 256).astype(np.uint8)


def face_chips(img, landmarks_list, size=CHIP_SIZE, padding=CHIP_PADDING, device=None):
    """img [H, W, 3] uint8 RGB (numpy or CUDA tensor), landmarks_list: n arrays [68, 2] -> CUDA float32 tensor [n, size, size, 3]
    (0..255), the input of DLibFaceNet.forward_dev.  The chips of one image come out of at most a few kernel launches: the image
    (its bounding box of all faces) travels to HBM once, pyramid levels are built there, every chip is one thread per pixel."""
    import torch
    n = len(landmarks_list)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    out = torch.empty((n, size, size, 3), dtype=torch.float32, device=dev)
    if n == 0:
        return out
    if hasattr(img, "is_cuda"):
        timg = img if img.is_cuda else img.to(dev)
    else:
        timg = torch.as_tensor(np.ascontiguousarray(img, dtype=np.uint8)).to(dev)
    if timg.dim() != 3 or timg.shape[2] != 3 or timg.dtype != torch.uint8:
        raise ValueError("img must be [H, W, 3] uint8 RGB")
    H, W = int(timg.shape[0]), int(timg.shape[1])
    details = [chip_details_from_landmarks(lm, size, padding) for lm in landmarks_list]
    bb, n_levels, maps = chip_maps(details, H, W)
    L = _lib.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    if bb[2] < bb[0] or bb[3] < bb[1]:
        return out.zero_()
    sub = timg[bb[1]:bb[3] + 1, bb[0]:bb[2] + 1].contiguous()
    levels = {-1: sub}
    cur = sub
    for k in range(n_levels):
        nr, nc = int(cur.shape[0]), int(cur.shape[1])
        if nr <= 8 or nc <= 8:   # dlib: pyramid_down of a tiny image is empty; chips that want it come out black
            levels[k] = torch.zeros((1, 1, 3), dtype=torch.uint8, device=dev)
            cur = levels[k]
            continue
        nxt = torch.empty(((nr - 3) // 2, (nc - 3) // 2, 3), dtype=torch.uint8, device=dev)
        tmp = torch.empty((nr, (nc - 3) // 2, 3), dtype=torch.int32, device=dev)
        _lib.check(L.cis_pyramid_down2_dev(cur.data_ptr(), nr, nc, nxt.data_ptr(), tmp.data_ptr(), stream))
        levels[k] = nxt
        cur = nxt
    by_level = {}
    for i, (lv, m6) in enumerate(maps):
        by_level.setdefault(lv, []).append(i)
    for lv, idxs in by_level.items():
        src = levels[lv]
        m = torch.as_tensor(np.stack([maps[i][1] for i in idxs])).to(dev)
        part = out if len(idxs) == n else torch.empty((len(idxs), size, size, 3), dtype=torch.float32, device=dev)
        _lib.check(L.cis_extract_chips_dev(src.data_ptr(), int(src.shape[0]), int(src.shape[1]), m.data_ptr(), len(idxs), size,
                                           part.data_ptr(), stream))
        if part is not out:
            out[torch.as_tensor(idxs, device=dev)] = part
    return out
