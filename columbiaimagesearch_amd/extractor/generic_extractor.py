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


def get_bbox_str(bbox):
    """left_top_right_bottom_score (reference: detector/utils.py:114-123)"""
    return "_".join(["{}"] * 5).format(bbox["left"], bbox["top"], bbox["right"], bbox["bottom"], bbox["score"])


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
        """Detector branch (:236-247) for a list of images: detections of ALL images are featurized in one GPU batch
        (`featurizer.featurize_dets(img, dets)` when the featurizer has it, else one `featurize(img, det)` each).
        An image without detections keeps ``processed = "0"`` like the reference's init_out_dict (:201-210)."""
        out = [None] * len(img_buffers)
        dtype = get_feat_dtype(self.featurizer_type)
        for i, buf in enumerate(img_buffers):
            try:
                img, dets = self.detector.detect_from_buffer_noinfos(buf, up_sample=1)
                row = self.init_out_dict()
                if dets:
                    if hasattr(self.featurizer, "featurize_dets"):
                        feats = self.featurizer.featurize_dets(img, dets)
                    else:
                        feats = [self.featurizer.featurize(img, d) for d in dets]
                    for det, feat in zip(dets, feats):
                        row[self.extr_str_processed] = str(1)
                        row[self.extr_str + "_" + get_bbox_str(det)] = normfeatB64encode(feat.astype(dtype))  # :241-247
                out[i] = row
            except Exception:
                if _raise:
                    raise
                out[i] = self.failed_out_dict()
        return out

    def process_batch(self, img_buffers, _raise=False):
        """One GPU forward for the whole list.  Images that cannot be decoded get the reference's failure row
        (DaemonBatchExtractor.run reports failed_out_dict on any exception, :109-127) and do not poison the batch."""
        if self.detector is not None:
            return self._process_batch_dets(img_buffers, _raise)
        good, tensors = [], []
        out = [None] * len(img_buffers)
        for i, buf in enumerate(img_buffers):
            try:
                tensors.append(self.featurizer.preprocess_img(buf))
                good.append(i)
            except Exception:
                out[i] = self.failed_out_dict()
        if good:
            import numpy as np
            feats = self.featurizer.net.forward(np.stack(tensors))
            for i, f in zip(good, feats):
                out[i] = self._row(f)
        return out
