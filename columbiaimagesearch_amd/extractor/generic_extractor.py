"""GenericExtractor with the reference's surface plus a batch entry point.

reference: cufacesearch/cufacesearch/extractor/generic_extractor.py:17-255.  The reference runs one image at a
time in N worker processes (DaemonBatchExtractor.run :49-162, spawned by
cufacesearch/cufacesearch/updater/extraction_processor.py:688-696).  On the GPU the unit of work is a batch:
`process_batch` featurizes a list of image buffers in one forward pass and returns, per image, exactly the
dictionary `process_buffer` would have produced (same column names, same base64 of the L2-normalised feature in
the featurizer's dtype) so that the rows can be pushed to HBase unchanged (extraction_processor.py:822).

Only the `detector_type == "full"` path is built (no detector: the whole image is featurized, :249-253); the dlib
HOG face detector is out of scope for this repo.
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


class GenericExtractor(object):
    def __init__(self, detector_type, featurizer_type, input_type, extr_column, extr_prefix, global_conf):
        """reference :168-199 (same arguments)"""
        if detector_type != "full":
            raise NotImplementedError("only the 'full' (no detector) extraction is built; got %r" % (detector_type,))
        self.detector_type = detector_type
        self.featurizer_type = featurizer_type
        self.input_type = input_type
        self.extr_column = extr_column
        self.global_conf = global_conf
        self.detector = None
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
        """reference :221-255 (full-image branch :249-253)"""
        return self._row(self.featurizer.featurize(img_buffer))

    def process_batch(self, img_buffers):
        """One GPU forward for the whole list.  Images that cannot be decoded get the reference's failure row
        (DaemonBatchExtractor.run reports failed_out_dict on any exception, :109-127) and do not poison the batch."""
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
