"""Reader (and writer) of caffe's binary ``.caffemodel`` -- the file the reference's ``sbcaffe_path`` points at
(``caffe_sentibank_train_iter_250000``, cufacesearch/cufacesearch/featurizer/sbpycaffe_img_featurizer.py:5,56-61,99).

A ``.caffemodel`` is a serialised ``NetParameter`` protobuf (caffe.proto of BVLC caffe, the commit pinned in
setup/DockerBuild/setup_columbia_image_search.sh:36-38).  Only the part that carries weights is decoded, with a
hand-written wire-format walker (no generated code, no protoc, no caffe):

    NetParameter      name=1 (string), layers=2 (V1LayerParameter, the pre-2015 format), layer=100 (LayerParameter)
    LayerParameter    name=1, type=2 (string), bottom=3, top=4, blobs=7 (BlobProto)
    V1LayerParameter  bottom=2, top=3, name=4, type=5 (enum), blobs=6 (BlobProto)
    BlobProto         num=1, channels=2, height=3, width=4 (legacy 4-D shape), data=5 (repeated float, packed or not),
                      diff=6, shape=7 (BlobShape), double_data=8, double_diff=9
    BlobShape         dim=1 (repeated int64, packed or not)

Both layer generations are read (a 2015 snapshot such as the Sentibank one stores ``layers``; anything re-saved by a
newer caffe stores ``layer``).  Packed float payloads are viewed with ``np.frombuffer`` -- a 227 MB model parses in
well under a second.
"""
import struct

import numpy as np

SENTIBANK_LAYERS = ["conv1", "conv2", "conv3", "conv4", "conv5", "fc6", "fc7"]
# caffe blob shapes of the layers the forward pass needs (data/pycaffe_sentibank.prototxt:7-197)
SENTIBANK_SHAPES = {"conv1": (96, 3, 11, 11), "conv2": (256, 48, 5, 5), "conv3": (384, 256, 3, 3),
                    "conv4": (384, 192, 3, 3), "conv5": (256, 192, 3, 3), "fc6": (4096, 9216), "fc7": (4096, 4096)}


# -- wire format -----------------------------------------------------------------------------------
def _read_varint(buf, pos):
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _fields(buf):
    """Yield (field number, wire type, value) over one message; value is an int (wire 0), a memoryview slice
    (wire 2) or the raw 4/8 bytes (wire 5/1)."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _read_varint(buf, pos)
        field, wire = key >> 3, key & 7
        if wire == 0:
            val, pos = _read_varint(buf, pos)
        elif wire == 2:
            n, pos = _read_varint(buf, pos)
            if pos + n > end:
                raise ValueError("truncated length-delimited field %d" % field)
            val = buf[pos:pos + n]
            pos += n
        elif wire == 5:
            val = buf[pos:pos + 4]
            pos += 4
        elif wire == 1:
            val = buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError("unsupported wire type %d (field %d)" % (wire, field))
        yield field, wire, val


def _parse_shape(buf):
    dims = []
    for field, wire, val in _fields(buf):
        if field == 1:
            if wire == 2:  # packed
                sub = bytes(val)
                p = 0
                while p < len(sub):
                    d, p = _read_varint(sub, p)
                    dims.append(d)
            else:
                dims.append(val)
    return dims


def _parse_blob(buf):
    legacy = {}
    shape = None
    chunks, scalars = [], []
    dchunks, dscalars = [], []
    for field, wire, val in _fields(buf):
        if field in (1, 2, 3, 4) and wire == 0:
            legacy[field] = val
        elif field == 5:
            if wire == 2:
                chunks.append(np.frombuffer(val, dtype="<f4"))
            else:
                scalars.append(struct.unpack("<f", bytes(val))[0])
        elif field == 8:
            if wire == 2:
                dchunks.append(np.frombuffer(val, dtype="<f8"))
            else:
                dscalars.append(struct.unpack("<d", bytes(val))[0])
        elif field == 7 and wire == 2:
            shape = _parse_shape(val)
    if chunks or scalars:
        parts = chunks + ([np.asarray(scalars, dtype=np.float32)] if scalars else [])
        data = parts[0] if len(parts) == 1 else np.concatenate(parts)
    elif dchunks or dscalars:
        parts = dchunks + ([np.asarray(dscalars, dtype=np.float64)] if dscalars else [])
        data = parts[0] if len(parts) == 1 else np.concatenate(parts)
    else:
        data = np.zeros(0, dtype=np.float32)
    if shape is None:
        shape = [legacy.get(k, 1) for k in (1, 2, 3, 4)] if legacy else [data.size]
    if int(np.prod(shape)) != data.size:
        raise ValueError("blob shape %r does not match %d values" % (shape, data.size))
    return data.reshape(shape)


def _parse_layer(buf, v1):
    name_f, blobs_f = (4, 6) if v1 else (1, 7)
    name, blobs = "", []
    for field, wire, val in _fields(buf):
        if field == name_f and wire == 2:
            name = bytes(val).decode("utf-8")
        elif field == blobs_f and wire == 2:
            blobs.append(_parse_blob(val))
    return name, blobs


def read_caffemodel(path_or_bytes):
    """{layer name: [blob arrays]} of every layer that holds blobs, in file order."""
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        raw = memoryview(path_or_bytes)
    else:
        raw = memoryview(np.fromfile(path_or_bytes, dtype=np.uint8)).cast("B")
    layers = {}
    for field, wire, val in _fields(raw):
        if wire == 2 and field in (2, 100):
            name, blobs = _parse_layer(val, v1=(field == 2))
            if blobs:
                layers[name] = blobs
    return layers


def sentibank_weights(path_or_bytes):
    """The 14 arrays SentiBankNet takes (``conv1_w, conv1_b, ..., fc7_w, fc7_b`` in caffe layout, float32) from a
    Sentibank ``.caffemodel``.  Legacy 4-D blobs (1,1,out,in) / (1,1,1,n) of old snapshots are reshaped."""
    layers = read_caffemodel(path_or_bytes)
    out = {}
    for name in SENTIBANK_LAYERS:
        if name not in layers or len(layers[name]) < 2:
            raise ValueError("caffemodel has no weights for layer %r (layers with blobs: %s)" % (name, sorted(layers)))
        w, b = layers[name][0], layers[name][1]
        want = SENTIBANK_SHAPES[name]
        if w.size != int(np.prod(want)) or b.size != want[0]:
            raise ValueError("layer %r: blob shapes %r / %r do not fit the DeepSentibank net %r" % (name, w.shape, b.shape, want))
        out[name + "_w"] = np.ascontiguousarray(w.reshape(want), dtype=np.float32)
        out[name + "_b"] = np.ascontiguousarray(b.reshape(want[0]), dtype=np.float32)
    return out


# -- writer (tests, and re-exporting converted weights for caffe users) -----------------------------------------
def _varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _blob(a, legacy):
    a = np.asarray(a, dtype=np.float32)
    if legacy:
        dims = (1,) * (4 - a.ndim) + tuple(a.shape)
        body = b"".join(_varint((f << 3) | 0) + _varint(int(d)) for f, d in zip((1, 2, 3, 4), dims))
        return body + _ld(5, np.ascontiguousarray(a, dtype="<f4").tobytes())
    shape = _ld(1, b"".join(_varint(int(d)) for d in a.shape))
    return _ld(5, np.ascontiguousarray(a, dtype="<f4").tobytes()) + _ld(7, shape)


def encode_caffemodel(layers, v1=False, net_name="net"):
    """bytes of a NetParameter holding `layers` = [(name, type, [blobs])]; v1=True writes the pre-2015 ``layers``
    field with legacy 4-D blob shapes (type is then the V1 enum value, e.g. 4 = CONVOLUTION, 14 = INNER_PRODUCT)."""
    out = _ld(1, net_name.encode())
    for name, ltype, blobs in layers:
        if v1:
            body = _ld(4, name.encode()) + _varint((5 << 3) | 0) + _varint(int(ltype))
            body += b"".join(_ld(6, _blob(b, True)) for b in blobs)
            out += _ld(2, body)
        else:
            body = _ld(1, name.encode()) + _ld(2, str(ltype).encode())
            body += b"".join(_ld(7, _blob(b, False)) for b in blobs)
            out += _ld(100, body)
    return out


def write_sentibank_caffemodel(weights, path, v1=False):
    """Write the 14 DeepSentibank arrays as a ``.caffemodel`` (inverse of sentibank_weights)."""
    layers = []
    for name in SENTIBANK_LAYERS:
        conv = name.startswith("conv")
        ltype = (4 if conv else 14) if v1 else ("Convolution" if conv else "InnerProduct")
        layers.append((name, ltype, [weights[name + "_w"], weights[name + "_b"]]))
    with open(path, "wb") as f:
        f.write(encode_caffemodel(layers, v1=v1, net_name="sentibank"))
