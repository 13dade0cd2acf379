"""CPU: the preprocessing pool (extractor/preprocess_pool.py) writes into its shared ring exactly what the featurizer's own
preprocessing returns (host restatement of sbpycaffe_img_featurizer.py:113-134), failures are flagged per slot."""
import io

import numpy as np


class _Spec(object):
    """What PreprocessPool needs of a featurizer: mu + preprocess_spec()."""

    def __init__(self):
        rs = np.random.RandomState(0)
        self.mu = (rs.rand(3, 227, 227) * 120).astype(np.float32)

    def preprocess_spec(self):
        from columbiaimagesearch_amd.featurizer.sbhip_img_featurizer import sentibank_preprocess
        return sentibank_preprocess, (self.mu, 14, 241, 14, 241, (256, 256, 3))


def test_pool_ring_equals_serial_preprocessing():
    from PIL import Image
    from columbiaimagesearch_amd.extractor.preprocess_pool import PreprocessPool
    from columbiaimagesearch_amd.featurizer.sbhip_img_featurizer import sentibank_preprocess
    spec = _Spec()
    rs = np.random.RandomState(1)
    bufs = []
    for i in range(9):
        b = io.BytesIO()
        Image.fromarray(rs.randint(0, 255, (150 + 31 * i, 400 - 23 * i, 3), dtype=np.uint8)).save(b, format="JPEG" if i % 2 else "PNG")
        bufs.append(b.getvalue())
    low = io.BytesIO()  # low contrast: bytescale stretches min..max before the resize
    Image.fromarray(rs.randint(100, 120, (64, 64, 3), dtype=np.uint8)).save(low, format="PNG")
    bufs.append(low.getvalue())
    bufs.insert(4, b"not an image")
    pool = PreprocessPool(spec, workers=2, slots=16)
    try:
        ring, ok = pool.run(bufs)
        assert ok.tolist() == [k != 4 for k in range(len(bufs))]
        fn, consts = spec.preprocess_spec()
        for k, b in enumerate(bufs):
            if k != 4:
                np.testing.assert_array_equal(ring[k], sentibank_preprocess(b, *consts))
        ring2, ok2 = pool.run(bufs[:3])  # the ring is reused
        assert ok2.all()
        np.testing.assert_array_equal(ring2[2], sentibank_preprocess(bufs[2], *consts))
    finally:
        pool.close()


def test_table_driven_bytescale_equals_the_plain_formula():
    """sentibank_preprocess looks the float conversion + bytescale up in a 256-entry table (PIL point); the plain statement of
    sbpycaffe_img_featurizer.py:113-134 -- float image, scipy bytescale, LANCZOS, crop, BGR, mean -- gives the same bits."""
    from PIL import Image
    from columbiaimagesearch_amd.featurizer.sbhip_img_featurizer import bytescale, sentibank_preprocess
    rs = np.random.RandomState(3)
    mu = (rs.rand(3, 227, 227) * 100).astype(np.float32)

    def plain(buf):
        im = Image.open(io.BytesIO(buf))
        img = (np.asarray(im.convert("RGB"), dtype=np.uint8) / 255.0).astype(np.float32)
        im = Image.fromarray(bytescale(img), mode="RGB").resize((256, 256), Image.LANCZOS)
        a = np.asarray(im, dtype=np.uint8)[14:241, 14:241, :]
        return a.transpose(2, 0, 1)[::-1].astype(np.float32) - mu

    cases = [(0, 256, "JPEG", 3), (100, 120, "PNG", 3), (7, 8, "PNG", 3), (250, 256, "PNG", 3), (3, 200, "JPEG", 3), (0, 255, "PNG", 1)]
    for lo, hi, fmt, ch in cases:
        shape = (211, 317, 3) if ch == 3 else (120, 90)
        b = io.BytesIO()
        Image.fromarray(rs.randint(lo, hi, shape).astype(np.uint8)).save(b, format=fmt)
        np.testing.assert_array_equal(sentibank_preprocess(b.getvalue(), mu, 14, 241, 14, 241), plain(b.getvalue()))
