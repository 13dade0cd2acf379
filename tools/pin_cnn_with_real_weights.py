#!/usr/bin/env python
"""Pin the CNN half of the path against the REAL third-party stacks, on a machine that has them.

The reference computes its descriptors in BVLC caffe (commit b963008a, setup/DockerBuild/setup_columbia_image_search.sh:36-38) and in
dlib (cufacesearch/cufacesearch/requirements.txt:3) with weights that are not in its tree; neither is installed in the build
container, so tests/test_cnn_hip_parity.py checks the HIP forwards against CPU restatements only ("parity unpinned").  This script
closes the gap wherever caffe / dlib and the weights DO exist: it runs the reference's own call sequence and writes a small golden file
(image bytes in, descriptors out) that tests/test_cnn_pinned.py picks up on any GPU box that also has the weights.

  python tools/pin_cnn_with_real_weights.py make-sentibank --prototxt pycaffe_sentibank.prototxt \\
         --caffemodel caffe_sentibank_train_iter_250000 --imgmean imagenet_mean.npy [--images a.jpg b.jpg ...]
  python tools/pin_cnn_with_real_weights.py make-dlib --pred shape_predictor_68_face_landmarks.dat \\
         --rec dlib_face_recognition_resnet_model_v1.dat --images faces1.jpg ...
  -> tests/golden/pin_sentibank.npz / tests/golden/pin_dlib.npz   (commit them: data, a few hundred KB)

  CIS_PIN_SENTIBANK_WEIGHTS=caffe_sentibank_train_iter_250000 CIS_PIN_IMGMEAN=imagenet_mean.npy \\
  CIS_PIN_DLIB_WEIGHTS=dlib_resnet.xml python -m pytest tests/test_cnn_pinned.py -m gpu      # (or the .dat itself: featurizer/dlib_dat.py; the .xml: INTEGRATION.md section 4c)

Python 2 or 3.  The make-* commands follow, line by line, what the reference runs:
  sentibank: sbpycaffe_img_featurizer.py:94 (caffe.Net(..., caffe.TEST)), :99-111 (Transformer: transpose, channel swap, mean),
             :113-134 (caffe.io.load_image, scipy.misc.imresize(.., (256,256,3), 'lanczos'), centre crop), :150-154 (forward, fc7)
  dlib:      dlib_featurizer.py:74,83 (shape_predictor, face_recognition_model_v1), :103-105 (sp(img, rect), compute_face_descriptor)
"""
from __future__ import print_function

import argparse
import hashlib
import io
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha1_of(path):
    h = hashlib.sha1()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def synthetic_jpegs(n, seed=7, size=(320, 400)):
    """Seeded test images (smooth structure + texture, different contrasts) as JPEG bytes, when no real images are given."""
    from PIL import Image
    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        yy, xx = np.mgrid[0:size[0], 0:size[1]].astype(np.float64)
        img = np.stack([127 + 100 * np.sin(xx / (9.0 + i)) * np.cos(yy / (13.0 + 2 * i)),
                        (xx * (0.3 + 0.1 * i) + yy * 0.2) % 256,
                        rs.randint(0, 256, size=size) * (0.2 + 0.15 * i) + 40], axis=-1)
        buf = io.BytesIO()
        Image.fromarray(img.clip(0, 255).astype(np.uint8)).save(buf, format="JPEG", quality=92)
        out.append(buf.getvalue())
    return out


def make_sentibank(args):
    import caffe  # noqa: the reference's dependency
    from scipy import misc
    from skimage import io as skio
    caffe.set_mode_cpu()
    net = caffe.Net(args.prototxt, args.caffemodel, caffe.TEST)                       # :94
    imgmean = np.load(args.imgmean)
    tsz, csz = (256, 256, 3), (227, 227)
    w_off, h_off = (tsz[0] - csz[0]) // 2, (tsz[1] - csz[1]) // 2                     # :69-80
    mu = imgmean[:, w_off:w_off + csz[0], h_off:h_off + csz[1]]
    tr = caffe.io.Transformer({"data": net.blobs["data"].data.shape})                 # :103-111
    tr.set_transpose("data", (2, 0, 1))
    tr.set_channel_swap("data", (2, 1, 0))
    tr.set_mean("data", mu)
    bufs = [open(p, "rb").read() for p in args.images] if args.images else synthetic_jpegs(6)
    feats = []
    for b in bufs:
        img = caffe.io.load_image(io.BytesIO(b))                                     # :122 (float RGB in [0, 1])
        img = misc.imresize(img, tsz, "lanczos")                                      # :127
        img = img[w_off:w_off + csz[0], h_off:h_off + csz[1], :]                      # :131
        net.blobs["data"].data[...] = tr.preprocess("data", img)                      # :134, :150
        net.forward()
        feats.append(net.blobs["fc7"].data[0].copy())                                 # :154
    out = args.out or os.path.join(REPO, "tests", "golden", "pin_sentibank.npz")
    np.savez_compressed(out, images=np.array(bufs, dtype=object), fc7=np.stack(feats).astype(np.float32),
                        weights_sha1=sha1_of(args.caffemodel), imgmean_sha1=sha1_of(args.imgmean),
                        made_with="caffe %s, scipy %s" % (getattr(caffe, "__version__", "?"), __import__("scipy").__version__))
    print("wrote", out, "(%d images)" % len(bufs))


def make_dlib(args):
    import dlib
    from skimage import io as skio
    sp = dlib.shape_predictor(args.pred)                                              # :74
    facerec = dlib.face_recognition_model_v1(args.rec)                                # :83
    det = dlib.get_frontal_face_detector()
    rec = {"images": [], "rects": [], "landmarks": [], "chips": [], "descriptors": []}
    for p in args.images:
        b = open(p, "rb").read()
        img = skio.imread(io.BytesIO(b))
        if img.ndim == 2:
            img = np.stack([img] * 3, axis=-1)                                        # :97-99
        for r in det(img, 1):
            shape = sp(img, r)                                                        # :103
            rec["images"].append(b)
            rec["rects"].append([r.left(), r.top(), r.right(), r.bottom()])
            rec["landmarks"].append([[shape.part(i).x, shape.part(i).y] for i in range(68)])
            rec["chips"].append(np.asarray(dlib.get_face_chip(img, shape, size=150, padding=0.25)))
            rec["descriptors"].append(np.array(facerec.compute_face_descriptor(img, shape)))  # :105
    if not rec["images"]:
        raise SystemExit("no face found in the given images")
    out = args.out or os.path.join(REPO, "tests", "golden", "pin_dlib.npz")
    np.savez_compressed(out, images=np.array(rec["images"], dtype=object), rects=np.array(rec["rects"]),
                        landmarks=np.array(rec["landmarks"], dtype=np.float64), chips=np.stack(rec["chips"]).astype(np.uint8),
                        descriptors=np.stack(rec["descriptors"]).astype(np.float64), rec_sha1=sha1_of(args.rec),
                        made_with="dlib %s" % dlib.__version__)
    print("wrote", out, "(%d faces)" % len(rec["images"]))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd")
    a = sub.add_parser("make-sentibank")
    a.add_argument("--prototxt", required=True); a.add_argument("--caffemodel", required=True); a.add_argument("--imgmean", required=True)
    a.add_argument("--images", nargs="*"); a.add_argument("--out")
    b = sub.add_parser("make-dlib")
    b.add_argument("--pred", required=True); b.add_argument("--rec", required=True); b.add_argument("--images", nargs="+", required=True)
    b.add_argument("--out")
    args = ap.parse_args()
    if args.cmd == "make-sentibank":
        make_sentibank(args)
    elif args.cmd == "make-dlib":
        make_dlib(args)
    else:
        ap.print_help()
        sys.exit(2)


if __name__ == "__main__":
    main()
