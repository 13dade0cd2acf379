"""dlib face-descriptor featurizer on the MI355X.

Mirror of DLibFeaturizer (cufacesearch/cufacesearch/featurizer/dlib_featurizer.py:50-105): constructor
``(global_conf_in, prefix)`` reading ``<prefix>pred_path`` (68-landmark shape predictor) and ``<prefix>rec_path``
(recognition network weights), ``featurize(img, bbox)`` -> 128 float64 values.  The landmark predictor is dlib host code (out
of scope, like image decoding: supply it through ``landmark_fn`` or pass ``landmarks=``); the chip alignment
(get_face_chip_details + extract_image_chip, featurizer/face_chip.py + csrc/face_chip.hip) and the 29-convolution ResNet run in
libcis_hip.so.  ``featurize_chips`` takes already aligned 150x150 RGB chips, ``featurize_landmarks`` an image and its shapes.

Weights: ``rec_path`` is dlib's ``.dat`` (featurizer/dlib_dat.py, unpinned), an ``.npz`` with the 117 arrays of ``tensor_names()``, or the XML that dlib's own
``net_to_xml`` writes from ``dlib_face_recognition_resnet_model_v1.dat`` (featurizer/dlib_weights.py; the ``.dat`` itself
is dlib's private C++ stream format -- INTEGRATION.md section 4 gives the six-line export program).
"""
import numpy as np

from .. import _lib
from .generic_featurizer import GenericFeaturizer

INPUT_HW = 150
FEAT_DIM = 128
N_BLOCKS = 14


def tensor_names():
    names = ["conv0_w", "conv0_b", "aff0_g", "aff0_b"]
    for i in range(N_BLOCKS):
        for half in ("a", "b"):
            names += ["b%d%s_w" % (i, half), "b%d%s_b" % (i, half), "b%d%s_g" % (i, half), "b%d%s_beta" % (i, half)]
    return names + ["fc_w"]


class DLibFaceNet(object):
    """The network alone: chips [n,150,150,3] (uint8 or float, RGB 0..255) -> [n,128] float32."""

    def __init__(self, weights):
        arrs = [np.ascontiguousarray(weights[n], dtype=np.float32) for n in tensor_names()]
        ptrs = (_lib.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        out = _lib.c_void_p()
        _lib.check(_lib.lib().cis_cnn_create(_lib.ctypes.byref(out), 2, ptrs, len(arrs)))
        self._h = out.value

    def view(self):
        """A second handle on the same weights with its own workspaces and streams (cis_cnn_create_view): run consecutive batches
        through the net and its views, each on its own torch stream -- several batches in flight fill the chip better than one.  The
        view keeps its base alive; close views before the base."""
        out = _lib.c_void_p()
        _lib.check(_lib.lib().cis_cnn_create_view(_lib.ctypes.byref(out), self._h))
        v = object.__new__(type(self))
        v._h = out.value
        v._base = self
        self._views = getattr(self, "_views", __import__("weakref").WeakSet())
        self._views.add(v)
        return v

    def close(self):
        for v in list(getattr(self, "_views", ())):
            v.close()
        if getattr(self, "_h", None):
            _lib.lib().cis_cnn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, chips):
        x = np.ascontiguousarray(chips, dtype=np.float32)
        if x.ndim != 4 or x.shape[1:] != (INPUT_HW, INPUT_HW, 3):
            raise ValueError("expected chips [n,150,150,3], got %r" % (x.shape,))
        out = np.empty((x.shape[0], FEAT_DIM), dtype=np.float32)
        _lib.check(_lib.lib().cis_cnn_forward(self._h, _lib.ptr(x), x.shape[0], _lib.ptr(out)))
        return out

    def forward_dev(self, x, out=None):
        """x: contiguous float32 CUDA tensor [n,150,150,3] -> CUDA tensor [n,128]; asynchronous"""
        import torch
        if not (x.is_cuda and x.is_contiguous() and x.dtype == torch.float32 and tuple(x.shape[1:]) == (INPUT_HW, INPUT_HW, 3)):
            raise ValueError("x must be a contiguous float32 [n,150,150,3] tensor on the GPU")
        if out is None:
            out = torch.empty((x.shape[0], FEAT_DIM), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().cis_cnn_forward_dev(self._h, x.data_ptr(), x.shape[0], out.data_ptr(),
                                                  torch.cuda.current_stream(x.device).cuda_stream))
        return out


class DLibHIPFeaturizer(GenericFeaturizer):
    def __init__(self, global_conf_in, prefix="DLIBFEAT_"):
        super(DLibHIPFeaturizer, self).__init__(global_conf_in, prefix)
        self.set_pp(pp="DLibHIPFeaturizer")
        self.pred_path = self.get_param("pred_path")
        self.rec_path = str(self.get_required_param("rec_path"))
        if self.rec_path.endswith(".npz"):
            z = np.load(self.rec_path)
            weights = {k: z[k] for k in z.files}
        elif self.rec_path.endswith(".xml"):
            from .dlib_weights import weights_from_net_xml
            weights = weights_from_net_xml(self.rec_path)
        elif self.rec_path.endswith(".dat"):
            # dlib's own stream (what the reference's config holds): walked by featurizer/dlib_dat.py -- a restatement of dlib's
            # serialisation that has never met a real file (no dlib here): it stops at the first byte that disagrees and names it;
            # the net_to_xml export is the fallback
            from .dlib_dat import weights_from_dat
            weights = weights_from_dat(self.rec_path)
        else:
            raise NotImplementedError("rec_path must be dlib's .dat, its net_to_xml export (.xml), or an .npz with the "
                                      "117 network tensors -- see INTEGRATION.md section 4")
        self.net = DLibFaceNet(weights)
        self._sp = None
        self.chip_fn = None      # (img, bbox) -> aligned 150x150x3 chip
        self.landmark_fn = None  # (img, bbox) -> [68, 2] landmarks (x, y): the chip is then cut on the GPU (featurizer/face_chip.py)

    def featurize_chips(self, chips):
        """aligned 150x150 RGB chips -> [n,128] float64 (dtype of the reference's descriptors, featsio.py:34-36)"""
        return self.net.forward(chips).astype(np.float64)

    def featurize_dets(self, img, dets):
        """All detections of one image in ONE forward pass -> [n,128] float64 (the reference calls featurize once per
        detection, generic_extractor.py:238)."""
        if len(img.shape) == 2:
            img = np.stack([img] * 3, axis=-1)  # reference :97-99 gray2rgb
        return self.featurize_chips(np.stack([self._chip(img, d) for d in dets]))

    def featurize_landmarks(self, img, shapes):
        """img [H, W, 3] uint8 RGB + n landmark sets [68, 2] (what dlib.shape_predictor returns, :103) -> [n, 128] float64:
        get_face_chip_details(shape, 150, 0.25) on the host (a similarity transform per face), chip extraction and the network on the
        GPU -- compute_face_descriptor(img, shape) (:105) without dlib.  The chips never visit the host."""
        from .face_chip import face_chips
        if len(img.shape) == 2:
            img = np.stack([img] * 3, axis=-1)
        chips = face_chips(img, [np.asarray(sh, dtype=np.float64) for sh in shapes])
        return self.net.forward_dev(chips).cpu().numpy().astype(np.float64)

    def featurize(self, img, bbox=None, img_type="scikit", landmarks=None):
        """reference :86-105: landmarks on the detected box, aligned chip, network.  The landmarks come from `landmarks` / the
        `landmark_fn` hook (then nothing of dlib is needed), else from dlib's shape predictor (or the whole chip from `chip_fn`)."""
        if len(img.shape) == 2:
            img = np.stack([img] * 3, axis=-1)
        if landmarks is None and self.landmark_fn is not None and self.chip_fn is None:
            landmarks = self.landmark_fn(img, bbox)
        if landmarks is not None:
            return self.featurize_landmarks(img, [landmarks])[0]
        return self.featurize_chips(np.asarray(self._chip(img, bbox))[None])[0]

    def _chip(self, img, bbox):
        if self.chip_fn is not None:
            return np.asarray(self.chip_fn(img, bbox))
        try:
            import dlib
        except ImportError:
            raise ImportError("dlib is needed for the landmark predictor / chip alignment of featurize(img, bbox); "
                              "use featurize_chips() with aligned chips")
        if self._sp is None:
            self._sp = dlib.shape_predictor(str(self.pred_path))
        rect = dlib.rectangle(int(bbox["left"]), int(bbox["top"]), int(bbox["right"]), int(bbox["bottom"]))
        shape = self._sp(img, rect)
        # the landmark predictor is dlib's; the chip is cut on the GPU like everything after it
        from .face_chip import face_chips
        lm = np.array([[shape.part(i).x, shape.part(i).y] for i in range(68)], dtype=np.float64)
        return face_chips(img, [lm])[0].cpu().numpy()
