"""Weight files the reference's featurizers load: caffe's .caffemodel (sbcaffe_path,
cufacesearch/featurizer/sbpycaffe_img_featurizer.py:5,99) decoded by our hand-written reader, checked against
google.protobuf's own serialiser/parser with a dynamic descriptor of caffe.proto's NetParameter / LayerParameter /
V1LayerParameter / BlobProto / BlobShape subset; dlib's net_to_xml export; and the Sentibank host preprocessing
(bytescale of scipy.misc.imresize).  CPU only."""
import io

import numpy as np
import pytest


def _caffe_pb():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "caffe_subset_test.proto"
    fd.package = "caffe_test"
    fd.syntax = "proto2"
    T = descriptor_pb2.FieldDescriptorProto

    def add(msg, name, number, typ, label=T.LABEL_OPTIONAL, type_name=None, packed=None):
        f = msg.field.add()
        f.name, f.number, f.type, f.label = name, number, typ, label
        if type_name:
            f.type_name = ".caffe_test." + type_name
        if packed is not None:
            f.options.packed = packed
        return f

    shape = fd.message_type.add(); shape.name = "BlobShape"
    add(shape, "dim", 1, T.TYPE_INT64, T.LABEL_REPEATED, packed=True)
    blob = fd.message_type.add(); blob.name = "BlobProto"
    add(blob, "shape", 7, T.TYPE_MESSAGE, type_name="BlobShape")
    add(blob, "data", 5, T.TYPE_FLOAT, T.LABEL_REPEATED, packed=True)
    add(blob, "diff", 6, T.TYPE_FLOAT, T.LABEL_REPEATED, packed=True)
    add(blob, "double_data", 8, T.TYPE_DOUBLE, T.LABEL_REPEATED, packed=True)
    for i, n in enumerate(["num", "channels", "height", "width"]):
        add(blob, n, i + 1, T.TYPE_INT32)
    layer = fd.message_type.add(); layer.name = "LayerParameter"
    add(layer, "name", 1, T.TYPE_STRING)
    add(layer, "type", 2, T.TYPE_STRING)
    add(layer, "bottom", 3, T.TYPE_STRING, T.LABEL_REPEATED)
    add(layer, "top", 4, T.TYPE_STRING, T.LABEL_REPEATED)
    add(layer, "blobs", 7, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name="BlobProto")
    v1 = fd.message_type.add(); v1.name = "V1LayerParameter"
    add(v1, "bottom", 2, T.TYPE_STRING, T.LABEL_REPEATED)
    add(v1, "top", 3, T.TYPE_STRING, T.LABEL_REPEATED)
    add(v1, "name", 4, T.TYPE_STRING)
    add(v1, "type", 5, T.TYPE_INT32)  # an enum on the wire is a varint
    add(v1, "blobs", 6, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name="BlobProto")
    add(v1, "blobs_lr", 7, T.TYPE_FLOAT, T.LABEL_REPEATED)  # an unpacked float field the reader must skip
    net = fd.message_type.add(); net.name = "NetParameter"
    add(net, "name", 1, T.TYPE_STRING)
    add(net, "layers", 2, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name="V1LayerParameter")
    add(net, "input", 3, T.TYPE_STRING, T.LABEL_REPEATED)
    add(net, "input_dim", 4, T.TYPE_INT32, T.LABEL_REPEATED)
    add(net, "layer", 100, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name="LayerParameter")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    desc = pool.FindMessageTypeByName("caffe_test.NetParameter")
    try:
        return message_factory.GetMessageClass(desc)
    except AttributeError:
        return message_factory.MessageFactory(pool).GetPrototype(desc)


def _small_layers(seed=0):
    rs = np.random.RandomState(seed)
    return [("conv1", (rs.randn(6, 3, 5, 5)).astype(np.float32), rs.randn(6).astype(np.float32)),
            ("conv2", (rs.randn(8, 3, 3, 3)).astype(np.float32), rs.randn(8).astype(np.float32)),
            ("fc6", (rs.randn(10, 72)).astype(np.float32), rs.randn(10).astype(np.float32))]


@pytest.mark.parametrize("v1", [False, True])
def test_we_read_what_google_protobuf_writes(v1):
    """NetParameter serialised by the official library (new `layer` format with BlobShape, and the 2015 `layers`
    format with legacy num/channels/height/width) -> our reader returns the same arrays."""
    from columbiaimagesearch_amd.featurizer.caffemodel import read_caffemodel
    Net = _caffe_pb()
    net = Net()
    net.name = "t"
    net.input.append("data")
    net.input_dim.extend([1, 3, 9, 9])
    for name, w, b in _small_layers():
        if v1:
            l = net.layers.add()
            l.name, l.type = name, 4
            l.bottom.append("x"); l.top.append(name)
            l.blobs_lr.extend([1.0, 2.0])
            for a in (w, b):
                bp = l.blobs.add()
                dims = (1,) * (4 - a.ndim) + a.shape
                bp.num, bp.channels, bp.height, bp.width = [int(d) for d in dims]
                bp.data.extend(a.ravel().tolist())
        else:
            l = net.layer.add()
            l.name, l.type = name, "Convolution"
            l.bottom.append("x"); l.top.append(name)
            for a in (w, b):
                bp = l.blobs.add()
                bp.shape.dim.extend(a.shape)
                bp.data.extend(a.ravel().tolist())
    # a layer without blobs (ReLU) must not show up
    if not v1:
        l = net.layer.add(); l.name, l.type = "relu1", "ReLU"
    got = read_caffemodel(net.SerializeToString())
    assert list(got) == ["conv1", "conv2", "fc6"]
    for name, w, b in _small_layers():
        gw, gb = got[name]
        if v1:
            assert gw.shape == (1,) * (4 - w.ndim) + w.shape  # legacy 4-D shape kept by the generic reader
        np.testing.assert_array_equal(gw.reshape(w.shape), w)
        np.testing.assert_array_equal(gb.reshape(b.shape), b)


@pytest.mark.parametrize("v1", [False, True])
def test_google_protobuf_reads_what_we_write(v1):
    from columbiaimagesearch_amd.featurizer.caffemodel import encode_caffemodel
    layers = [(n, 4 if v1 else "Convolution", [w, b]) for n, w, b in _small_layers(1)]
    msg = _caffe_pb()()
    msg.ParseFromString(encode_caffemodel(layers, v1=v1))
    recs = msg.layers if v1 else msg.layer
    assert [r.name for r in recs] == ["conv1", "conv2", "fc6"]
    for r, (n, w, b) in zip(recs, _small_layers(1)):
        np.testing.assert_array_equal(np.array(r.blobs[0].data, dtype=np.float32), w.ravel())
        if v1:
            assert (r.blobs[0].num, r.blobs[0].channels, r.blobs[0].height, r.blobs[0].width) == (1,) * (4 - w.ndim) + w.shape
        else:
            assert tuple(r.blobs[0].shape.dim) == w.shape


def test_unpacked_and_double_blobs():
    """`repeated float data` written unpacked (wire type 5 per value) and double_data are both legal encodings."""
    import struct
    from columbiaimagesearch_amd.featurizer import caffemodel as C
    vals = np.arange(6, dtype=np.float32) * 0.5
    blob = b"".join(C._varint((5 << 3) | 5) + struct.pack("<f", v) for v in vals) + C._ld(7, C._ld(1, bytes([2, 3])))
    net = C._ld(100, C._ld(1, b"ip") + C._ld(7, blob))
    np.testing.assert_array_equal(C.read_caffemodel(net)["ip"][0], vals.reshape(2, 3))
    dvals = np.arange(4, dtype=np.float64) / 3
    blob = C._ld(8, dvals.tobytes()) + C._ld(7, C._ld(1, bytes([4])))
    net = C._ld(100, C._ld(1, b"d") + C._ld(7, blob))
    got = C.read_caffemodel(net)["d"][0]
    assert got.dtype == np.float64
    np.testing.assert_array_equal(got, dvals)


@pytest.mark.parametrize("v1", [False, True])
def test_sentibank_caffemodel_round_trip(tmp_path, v1):
    """The full-size DeepSentibank weight set (56.9 M parameters) written as a .caffemodel and read back through
    the path SentiBankHIPImgFeaturizer._load_weights takes for the reference's sbcaffe_path."""
    from columbiaimagesearch_amd.featurizer.caffemodel import write_sentibank_caffemodel
    from columbiaimagesearch_amd.featurizer.sbhip_img_featurizer import SentiBankHIPImgFeaturizer
    from columbiaimagesearch_amd.featurizer.synthetic import sentibank_weights
    w = sentibank_weights(5)
    p = str(tmp_path / "caffe_sentibank_train_iter_250000")
    write_sentibank_caffemodel(w, p, v1=v1)
    got = SentiBankHIPImgFeaturizer._load_weights(p)
    assert sorted(got) == sorted(w)
    for k in w:
        assert got[k].dtype == np.float32 and got[k].shape == w[k].shape
        np.testing.assert_array_equal(got[k], w[k])


def test_sentibank_caffemodel_errors(tmp_path):
    from columbiaimagesearch_amd.featurizer.caffemodel import encode_caffemodel, sentibank_weights
    with pytest.raises(ValueError):
        sentibank_weights(encode_caffemodel([("conv1", "Convolution", [np.zeros((96, 3, 11, 11)), np.zeros(96)])]))
    with pytest.raises(ValueError):
        sentibank_weights(encode_caffemodel([("conv1", "Convolution", [np.zeros((96, 3, 3, 3)), np.zeros(96)])]))


def test_dlib_net_xml_round_trip(tmp_path):
    from columbiaimagesearch_amd.featurizer.dlib_weights import weights_from_net_xml, write_net_xml
    from columbiaimagesearch_amd.featurizer.synthetic import dlib_weights
    w = dlib_weights(4)
    p = str(tmp_path / "net.xml")
    write_net_xml(w, p)
    got = weights_from_net_xml(p)
    assert len(got) == 117
    for k in w:
        assert got[k].shape == w[k].shape
        np.testing.assert_array_equal(got[k], w[k])


def test_dlib_dat_stream_round_trip(tmp_path):
    """featurizer/dlib_dat.py: dlib's own serialisation of anet_type, written by the module's restatement of the serializer (both
    add_layer versions) and read back -- the same 117 arrays as the net_to_xml route gives; primitives against hand-made bytes;
    a stream that disagrees stops with the byte offset."""
    from columbiaimagesearch_amd.featurizer import dlib_dat as DD
    from columbiaimagesearch_amd.featurizer.dlib_weights import weights_from_net_xml, write_net_xml
    from columbiaimagesearch_amd.featurizer.synthetic import dlib_weights
    # primitives: 300 = control 0x02 + 2C 01; -5 = 0x81 05; float_details of 0.5f = mantissa 1, exponent -1
    s = DD._In(bytes([0x02, 0x2C, 0x01, 0x81, 0x05, 0x01, 0x01, 0x81, 0x01]) + b"1" + bytes([0x01, 0x03]) + b"abc")
    assert s.int() == 300 and s.int() == -5 and s.real() == 0.5 and s.bool() is True and s.string() == "abc"
    o = DD._Out()
    for v in (0.0, 1.0, -2.5, 3.0e-7, 122.782, float("inf")):
        o.real(v)
    s = DD._In(b"".join(o.parts))
    assert [s.real() for _ in range(6)] == [0.0, 1.0, -2.5, 3.0e-7, 122.782, float("inf")]
    assert len(DD.anet_layers()) == 1 + 1 + 1 + 14 * 8 + 4 * 3 + 5  # loss, fc, avg pool, 14 blocks of 8 layers, 3 more in the 4 down blocks, stem + input
    w = dlib_weights(5)
    for lv, iv in ((2, 3), (1, 2)):
        data = DD.write_dat(w, str(tmp_path / "net.dat"), layer_version=lv, input_layer_version=iv)
        got = DD.weights_from_dat(str(tmp_path / "net.dat"))
        assert len(got) == 117
        for k in w:
            assert got[k].shape == w[k].shape and got[k].dtype == np.float32
            np.testing.assert_array_equal(got[k], w[k])
    write_net_xml(w, str(tmp_path / "net.xml"))
    via_xml = weights_from_net_xml(str(tmp_path / "net.xml"))
    for k in w:
        np.testing.assert_array_equal(via_xml[k], got[k])
    with pytest.raises(ValueError, match="byte"):
        DD.weights_from_dat(data[:-3])
    bad = bytearray(data)
    bad[1] = 9  # the loss layer's version
    with pytest.raises(ValueError, match="add_loss_layer: version 9"):
        DD.weights_from_dat(bytes(bad))
    with pytest.raises(ValueError, match="behind the network"):
        DD.weights_from_dat(data + b"\x00")


def test_bytescale_is_the_scipy_misc_formula():
    """scipy.misc.bytescale on a float image (what imresize -> toimage applies, sbpycaffe_img_featurizer.py:126):
    (x - min) * 255/(max - min), clipped, +0.5, truncated.  A low-contrast image is stretched to the full range,
    a constant image maps to 0."""
    from columbiaimagesearch_amd.featurizer.sbhip_img_featurizer import SentiBankHIPImgFeaturizer as F
    rs = np.random.RandomState(0)
    img = (rs.randint(90, 140, size=(31, 17, 3)).astype(np.uint8) / 255.0).astype(np.float32)
    got = F.bytescale(img)
    cmin, cmax = float(img.min()), float(img.max())
    want = np.floor(np.clip((img.astype(np.float64) - cmin) * (255.0 / (cmax - cmin)), 0, 255) + 0.5)
    assert got.dtype == np.uint8 and got.min() == 0 and got.max() == 255
    assert np.abs(got.astype(np.int64) - want.astype(np.int64)).max() <= 1  # float32 vs float64 rounding of .5 cases
    assert (got.astype(np.int64) != want.astype(np.int64)).mean() < 0.01
    full = (rs.randint(0, 256, size=(16, 16, 3)).astype(np.uint8))
    full[0, 0, 0], full[0, 0, 1] = 0, 255
    np.testing.assert_array_equal(F.bytescale((full / 255.0).astype(np.float32)), full)  # full range: identity
    assert (F.bytescale(np.full((4, 4, 3), 0.3, dtype=np.float32)) == 0).all()


def test_preprocess_low_contrast_image(tmp_path):
    """End of the ADVICE item: a low-contrast PNG is stretched before the LANCZOS resize, so the preprocessed tensor
    spans the whole pixel range (minus the mean) instead of the image's narrow band."""
    from PIL import Image
    from columbiaimagesearch_amd.featurizer.sbhip_img_featurizer import SentiBankHIPImgFeaturizer as F
    rs = np.random.RandomState(1)
    a = rs.randint(100, 121, size=(300, 280, 3)).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, format="PNG")
    f = F.__new__(F)  # preprocessing only: no weights, no device
    f.target_size, f.crop_size = (256, 256, 3), (227, 227)
    f.w_boff = f.h_boff = 14
    f.w_eoff = f.h_eoff = 241
    f.mu = np.zeros((3, 227, 227), dtype=np.float32)
    x = f.preprocess_img(buf.getvalue())
    assert x.shape == (3, 227, 227) and x.dtype == np.float32
    assert x.min() < 40 and x.max() > 215  # stretched (LANCZOS smooths the extremes a little); raw would be 100..120
    # reference order of operations on the same image, written out step by step
    img = (a / 255.0).astype(np.float32)
    u8 = F.bytescale(img)
    ref = np.asarray(Image.fromarray(u8).resize((256, 256), Image.LANCZOS))[14:241, 14:241, :]
    np.testing.assert_array_equal(x, ref.transpose(2, 0, 1)[::-1].astype(np.float32))
