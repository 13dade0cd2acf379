"""GenericExtractor with the reference's surface plus a batch entry point.

reference: cufacesearch/cufacesearch/extractor/generic_extractor.py:17-255.  The reference runs one image at a
time in N worker processes (DaemonBatchExtractor.run :49-162, spawned by
cufacesearch/cufacesearch/updater/extraction_processor.py:688-696).  On the GPU the unit of work is a batch:
`process_batch` featurizes a list of image buffers in one forward pass and returns, per image, exactly the
dictionary `process_buffer` would have produced (same column names, same base64 of the L2-normalised feature in
the featurizer's dtype) so that the rows can be pushed to HBase unchanged (extraction_processor.py:822).

Both branches of `process_buffer` are built: `detector_type == "full"` (no detector: the whole image is featurized,
:249-253) and the detector branch (:236-247: one feature per detection, column ``<extr_str>_<l>_<t>_<r>_<b>_<score>``).
The detector itself (dlib's HOG face detector, detector/dlib_detector.py) is out of scope for this repo (SURVEY.md
section 2 row 10): `detector_type == "dlib"` builds it from the `dlib` module when that is importable, and any object
with the reference's ``detect_from_buffer_noinfos(img_buffer, up_sample=1) -> (img, [bbox dict])`` can be passed as
``detector=``.
"""
from ..featurizer.featsio import get_feat_dtype, normfeatB64encode
from ..featurizer.generic_featurizer import get_featurizer

EXTR_STR_PROCESSED = "processed"  # cufacesearch/cufacesearch/indexer/hbase_indexer_minimal.py:40-41
EXTR_STR_FAILED = "failed"


def build_extr_str(featurizer_type, detector_type, input_type):
    """reference :17-18"""
    return "_".join([featurizer_type, "feat", detector_type, input_type])


def build_extr_str_processed(featurizer_type, detector_type, input_type):
    """reference :20-21"""
    return build_extr_str(featurizer_type, detector_type, input_type) + "_" + EXTR_STR_PROCESSED


def build_extr_str_failed(featurizer_type, detector_type, input_type):
    """reference :23-24"""
    return build_extr_str(featurizer_type, detector_type, input_type) + "_" + EXTR_STR_FAILED


PY2_FLOAT_STR = True  # column names as the reference's Python 2 wrote them (HBase rows written by it stay addressable)


def _py2_str(v):
    """What "{}".format(v) gave under the reference's Python 2 for a float: str(float) = 12 significant digits
    ('%.12g', with '.0' appended to integral values) where Python 3 prints the 17-digit repr.  Other types unchanged."""
    if PY2_FLOAT_STR and isinstance(v, float):
        s = "%.12g" % v
        if "." not in s and "e" not in s and "n" not in s:  # integral value (not inf / nan)
            s += ".0"
        return s
    return "{}".format(v)


def get_bbox_str(bbox):
    """left_top_right_bottom_score (reference: detector/utils.py:114-123)"""
    return "_".join(_py2_str(bbox[k]) for k in ("left", "top", "right", "bottom", "score"))


class DLibHOGDetector(object):
    """The reference's DLibFaceDetector (detector/dlib_detector.py) surface over the `dlib` module: host code, only
    built when dlib is installed."""

    def __init__(self):
        import dlib  # ImportError here = no detector in this environment
        self._det = dlib.get_frontal_face_detector()

    def detect_from_buffer_noinfos(self, img_buffer, up_sample=1):
        import io
        import numpy as np
        from PIL import Image
        img = np.asarray(Image.open(io.BytesIO(img_buffer) if isinstance(img_buffer, (bytes, bytearray)) else img_buffer).convert("RGB"))
        dets, scores, _ = self._det.run(img, up_sample, 0.0)
        return img, [{"left": d.left(), "top": d.top(), "right": d.right(), "bottom": d.bottom(), "score": s}
                     for d, s in zip(dets, scores)]


def get_detector(detector_type):
    """reference: detector/utils.py:100-112"""
    if detector_type == "dlib":
        return DLibHOGDetector()
    if detector_type == "full":
        return None
    raise ValueError("[{}: error] unknown 'detector' {}.".format("get_detector", detector_type))


class GenericExtractor(object):
    def __init__(self, detector_type, featurizer_type, input_type, extr_column, extr_prefix, global_conf, detector=None):
        """reference :168-199 (same arguments; `detector` optionally injects the detector object)"""
        self.detector_type = detector_type
        self.featurizer_type = featurizer_type
        self.input_type = input_type
        self.extr_column = extr_column
        self.global_conf = global_conf
        self.detector = detector if detector is not None else get_detector(self.detector_type)
        self.featurizer = get_featurizer(self.featurizer_type, self.global_conf, prefix=extr_prefix)
        self.extr_str = str(self.extr_column + ":" + build_extr_str(featurizer_type, detector_type, input_type))
        self.extr_str_processed = str(self.extr_column + ":" + build_extr_str_processed(featurizer_type, detector_type, input_type))
        self.extr_str_failed = str(self.extr_column + ":" + build_extr_str_failed(featurizer_type, detector_type, input_type))

    def init_out_dict(self):
        """reference :201-210"""
        return {self.extr_str_processed: str(0)}

    def failed_out_dict(self):
        """reference :212-219"""
        return {self.extr_str_failed: str(1)}

    def _row(self, feat):
        feat = feat.astype(get_feat_dtype(self.featurizer_type))  # reference :251
        return {self.extr_str: normfeatB64encode(feat), self.extr_str_processed: str(1)}

    def process_buffer(self, img_buffer):
        """reference :221-255"""
        if self.detector is None:
            return self._row(self.featurizer.featurize(img_buffer))  # :249-253
        return self.process_batch([img_buffer], _raise=True)[0]

    def _process_batch_dets(self, img_buffers, _raise):
        """Detector branch (:236-247) for a list of images.  With a featurizer that separates alignment from the network
        (`_chip` + `featurize_chips`: the dlib face featurizer) the aligned chips of ALL images' detections are gathered and
        go through the network in GPU batches of up to `chip_batch` chips; otherwise one `featurize(img, det)` per detection
        like the reference.  An image without detections keeps ``processed = "0"`` like init_out_dict (:201-210)."""
        import numpy as np
        out = [None] * len(img_buffers)
        dtype = get_feat_dtype(self.featurizer_type)
        batched = hasattr(self.featurizer, "_chip") and hasattr(self.featurizer, "featurize_chips")
        chips, owner = [], []  # aligned chips of every image, (image index, detection) each
        for i, buf in enumerate(img_buffers):
            try:
                img, dets = self.detector.detect_from_buffer_noinfos(buf, up_sample=1)
                row = self.init_out_dict()
                if dets and batched:
                    if len(img.shape) == 2:
                        img = np.stack([img] * 3, axis=-1)  # dlib_featurizer.py:97-99 gray2rgb
                    mine = [np.asarray(self.featurizer._chip(img, d)) for d in dets]  # any failure fails the image, not the batch
                    chips.extend(mine)
                    owner.extend((i, d) for d in dets)
                elif dets:
                    for det in dets:
                        feat = self.featurizer.featurize(img, det)
                        row[self.extr_str_processed] = str(1)
                        row[self.extr_str + "_" + get_bbox_str(det)] = normfeatB64encode(feat.astype(dtype))  # :241-247
                out[i] = row
            except Exception:
                if _raise:
                    raise
                out[i] = self.failed_out_dict()
        for a in range(0, len(chips), self.chip_batch):
            feats = self.featurizer.featurize_chips(np.stack(chips[a:a + self.chip_batch]))
            for (i, det), feat in zip(owner[a:a + self.chip_batch], feats):
                out[i][self.extr_str_processed] = str(1)
                out[i][self.extr_str + "_" + get_bbox_str(det)] = normfeatB64encode(feat.astype(dtype))  # :241-247
        return out

    chip_batch = 256  # chips per forward pass of the detector branch (BASELINE batch)

    def process_batch(self, img_buffers, _raise=False, pool=None):
        """One GPU forward for the whole list.  Images that cannot be decoded get the reference's failure row
        (DaemonBatchExtractor.run reports failed_out_dict on any exception, :109-127) and do not poison the batch.
        pool: an extractor.preprocess_pool.PreprocessPool of worker processes that decode / resize in parallel into a
        page-locked ring (rows are identical to the serial path: the workers run the featurizer's own preprocessing)."""
        if self.detector is not None:
            return self._process_batch_dets(img_buffers, _raise)
        import numpy as np
        out = [None] * len(img_buffers)
        if pool is not None:
            # two halves of the ring alternate: the workers decode batch b + 1 while the GPU runs batch b and the rows of
            # batch b are encoded here
            import torch
            half = max(pool.slots // 2, 1)
            starts = list(range(0, len(img_buffers), half))
            nxt = pool.start(img_buffers[0:half], 0) if starts else None
            for bi, a in enumerate(starts):
                ring, ok = pool.finish(nxt)
                if bi + 1 < len(starts):
                    b0 = starts[bi + 1]
                    nxt = pool.start(img_buffers[b0:b0 + half], ((bi + 1) % 2) * half)
                if ok.any():
                    x = torch.from_numpy(ring).cuda(non_blocking=True)  # slots of failed images hold stale data: computed, ignored
                    feats = self.featurizer.net.forward_dev(x).cpu().numpy()
                for k in range(len(ok)):
                    out[a + k] = self._row(feats[k]) if ok[k] else self.failed_out_dict()
            return out
        good, tensors = [], []
        for i, buf in enumerate(img_buffers):
            try:
                tensors.append(self.featurizer.preprocess_img(buf))
                good.append(i)
            except Exception:
                out[i] = self.failed_out_dict()
        if good:
            feats = self.featurizer.net.forward(np.stack(tensors))
            for i, f in zip(good, feats):
                out[i] = self._row(f)
        return out
