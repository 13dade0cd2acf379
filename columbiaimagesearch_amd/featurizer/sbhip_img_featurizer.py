"""DeepSentibank featurizer on the MI355X.

Mirror of SentiBankPyCaffeImgFeaturizer (cufacesearch/cufacesearch/featurizer/sbpycaffe_img_featurizer.py:22-154):
same constructor ``(global_conf_in, prefix)``, same configuration keys (``<prefix>sbcaffe_path`` = weights,
``<prefix>imgmean_path`` = imagenet_mean.npy), same ``featurize(img, bbox=None, img_type="buffer")`` -> 4096
float32 (un-normalised: the caller normalises, generic_extractor.py:238-248).  Preprocessing stays on the host
(decode, 256x256 lanczos resize through uint8, centre crop 227, RGB->BGR, mean subtraction: :113-134); the
forward pass runs in libcis_hip.so.  ``featurize_batch`` is the addition that feeds the GPU whole batches.

Weights: ``sbcaffe_path`` is the reference's ``.caffemodel`` (:5,56-61), read by featurizer/caffemodel.py (a
hand-written NetParameter/BlobProto wire-format reader: no caffe, no protoc), or a ``.npz`` with the 14 arrays
``conv1_w, conv1_b, ..., fc7_w, fc7_b`` in caffe layout.
"""
import io

import numpy as np

from .. import _lib
from .generic_featurizer import GenericFeaturizer

TENSOR_NAMES = ["conv1", "conv2", "conv3", "conv4", "conv5", "fc6", "fc7"]
INPUT_HW = 227
FEAT_DIM = 4096


def bytescale(data, low=0, high=255, cmin=None, cmax=None):
    """scipy.misc.bytescale as imresize -> toimage applies it to the float image caffe.io.load_image returns:
    the image's own min..max is stretched to 0..255 before the resize (so a low-contrast image is NOT resized
    as its uint8 self).  float32 arithmetic like the reference's (float32 image, python-float scale)."""
    if cmin is None:
        cmin = data.min()
    if cmax is None:
        cmax = data.max()
    cscale = cmax - cmin
    if cscale == 0:
        cscale = 1
    scale = float(high - low) / cscale
    bytedata = (data - cmin) * data.dtype.type(scale) + data.dtype.type(low)
    return (bytedata.clip(low, high) + data.dtype.type(0.5)).astype(np.uint8)


_U8_AS_FLOAT = (np.arange(256, dtype=np.uint8) / 255.0).astype(np.float32)  # img_as_float of every possible pixel value


def load_and_bytescale(a8):
    """uint8 RGB image -> what bytescale((a8 / 255.0).astype(float32)) returns, through a 256-entry table: the chain is a
    function of the pixel value and of the image's min / max only (the float conversion is monotone, so the float min / max
    are the conversions of the uint8 min / max), and the table is computed by the very same numpy expressions on the 256
    possible values -- identical bytes at a fraction of the passes over the image."""
    lut = bytescale(_U8_AS_FLOAT, cmin=_U8_AS_FLOAT[a8.min()], cmax=_U8_AS_FLOAT[a8.max()])
    return lut[a8]


def sentibank_preprocess(img_buffer, mu, w_boff, w_eoff, h_boff, h_eoff, target_size=(256, 256, 3), out=None):
    """host-side restatement of reference :113-134 with PIL: caffe.io.load_image -> RGB float32 in [0,1]
    (skimage.img_as_float(...).astype(float32)); scipy.misc.imresize = toimage (bytescale: min..max -> 0..255,
    uint8) + PIL LANCZOS resize to 256x256; centre crop 227; HWC->CHW; RGB->BGR; subtract the cropped mean.
    A plain function of picklable arguments, so that worker processes run exactly what featurize() runs."""
    from PIL import Image
    if isinstance(img_buffer, (bytes, bytearray)):
        img_buffer = io.BytesIO(img_buffer)
    im = Image.open(img_buffer)
    if getattr(im, "n_frames", 1) > 1:
        im.seek(1)  # reference takes image[1] of a GIF (:123-125)
    rgb = im.convert("RGB")
    ext = rgb.getextrema()  # per band (min, max): the table needs the image's own range
    lut = bytescale(_U8_AS_FLOAT, cmin=_U8_AS_FLOAT[min(e[0] for e in ext)], cmax=_U8_AS_FLOAT[max(e[1] for e in ext)])
    im = rgb.point(lut.tolist() * 3).resize((target_size[1], target_size[0]), Image.LANCZOS)  # load_and_bytescale, in PIL's C loop
    a = np.asarray(im, dtype=np.uint8)[w_boff:w_eoff, h_boff:h_eoff, :]
    chw = a.transpose(2, 0, 1)[::-1].astype(np.float32)  # channel swap (2,1,0): RGB -> BGR
    if out is not None:
        np.subtract(chw, mu, out=out)
        return out
    return chw - mu


class SentiBankNet(object):
    """The network alone: float32 NCHW batch in, fc7 (post-ReLU) out.  Owns a cis_cnn handle."""

    def __init__(self, weights):
        arrs = []
        for n in TENSOR_NAMES:
            arrs.append(np.ascontiguousarray(weights[n + "_w"], dtype=np.float32))
            arrs.append(np.ascontiguousarray(weights[n + "_b"], dtype=np.float32))
        ptrs = (_lib.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        out = _lib.c_void_p()
        _lib.check(_lib.lib().cis_cnn_create(_lib.ctypes.byref(out), 1, ptrs, len(arrs)))
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

    def forward(self, x):
        """x: [n,3,227,227] float32 (host) -> [n,4096] float32"""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 4 or x.shape[1:] != (3, INPUT_HW, INPUT_HW):
            raise ValueError("expected [n,3,227,227] float32, got %r" % (x.shape,))
        out = np.empty((x.shape[0], FEAT_DIM), dtype=np.float32)
        _lib.check(_lib.lib().cis_cnn_forward(self._h, _lib.ptr(x), x.shape[0], _lib.ptr(out)))
        return out

    def forward_dev(self, x, out=None):
        """x: contiguous float32 CUDA tensor [n,3,227,227] -> CUDA tensor [n,4096]; asynchronous"""
        import torch
        if not (x.is_cuda and x.is_contiguous() and x.dtype == torch.float32 and tuple(x.shape[1:]) == (3, INPUT_HW, INPUT_HW)):
            raise ValueError("x must be a contiguous float32 [n,3,227,227] tensor on the GPU")
        if out is None:
            out = torch.empty((x.shape[0], FEAT_DIM), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().cis_cnn_forward_dev(self._h, x.data_ptr(), x.shape[0], out.data_ptr(),
                                                  torch.cuda.current_stream(x.device).cuda_stream))
        return out


class SentiBankHIPImgFeaturizer(GenericFeaturizer):
    def __init__(self, global_conf_in, prefix="SBPYCAFFEIMGFEAT_"):
        super(SentiBankHIPImgFeaturizer, self).__init__(global_conf_in, prefix)
        self.set_pp(pp="SentiBankHIPImgFeaturizer")
        self.output_blobs = ["fc7"]
        self.target_size = (256, 256, 3)
        self.crop_size = (INPUT_HW, INPUT_HW)
        self.resize_type = "lanczos"  # reference :47
        self.sbcaffe_path = str(self.get_required_param("sbcaffe_path"))
        self.imgnetmean_path = str(self.get_required_param("imgmean_path"))
        # mean image handling as reference :66-80: (3,256,256) BGR mean, centre-cropped to 227
        imgmean = np.load(self.imgnetmean_path)
        off = (imgmean.shape[1] - INPUT_HW) // 2
        self.w_boff = self.h_boff = off
        self.w_eoff = self.h_eoff = off + INPUT_HW
        self.mu = np.ascontiguousarray(imgmean[:, off:off + INPUT_HW, off:off + INPUT_HW], dtype=np.float32)
        self.net = SentiBankNet(self._load_weights(self.sbcaffe_path))

    @staticmethod
    def _load_weights(path):
        if path.endswith(".npz"):
            z = np.load(path)
            return {k: z[k] for k in z.files}
        from .caffemodel import sentibank_weights
        return sentibank_weights(path)  # the reference's file: a serialised caffe NetParameter

    bytescale = staticmethod(lambda data, low=0, high=255: bytescale(data, low, high))

    def preprocess_spec(self):
        """What a worker process needs to run this featurizer's preprocessing without the GPU object (picklable):
        (function, constant arguments) -- extractor/preprocess_pool.py."""
        return sentibank_preprocess, (self.mu, self.w_boff, self.w_eoff, self.h_boff, self.h_eoff, self.target_size)

    def preprocess_img(self, img_buffer):
        """host-side restatement of reference :113-134 (sentibank_preprocess below)"""
        return sentibank_preprocess(img_buffer, self.mu, self.w_boff, self.w_eoff, self.h_boff, self.h_eoff, self.target_size)

    def featurize(self, img, bbox=None, img_type="buffer"):
        """reference :137-154 -> np.ndarray (4096,) float32; `bbox` is ignored there too"""
        return self.net.forward(self.preprocess_img(img)[None])[0]

    def featurize_batch(self, imgs):
        """list of image buffers -> [n,4096] float32 in one GPU batch"""
        return self.net.forward(np.stack([self.preprocess_img(i) for i in imgs]))
