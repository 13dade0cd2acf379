"""LOPQModelParams protobuf exchange format, wire-compatible with the reference's generated
lopq/lopq/lopq_model_pb2.py (schema: Vector{values: packed float=1}, Matrix{values: packed float=1,
shape: uint32=2}, LOPQModelParams{D=1, V=2, M=3, num_subquantizers=4, Cs=5 Matrix*, Rs=6 Matrix*, mus=7 Vector*,
subs=8 Matrix*}).  Hand-written varint/length-delimited codec: no generated code, no protoc.

reference: LOPQModel.export_proto / load_proto, lopq/lopq/model.py:748-820 (values are stored as float32).
"""
import struct

import numpy as np


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


def _key(field, wire):
    return _varint((field << 3) | wire)


def _ld(field, payload):
    return _key(field, 2) + _varint(len(payload)) + payload


def _packed_floats(a):
    return np.ascontiguousarray(a, dtype="<f4").tobytes()


def _matrix(a):
    a = np.asarray(a)
    body = _ld(1, _packed_floats(a.ravel(order="C")))
    for s in a.shape:
        body += _key(2, 0) + _varint(int(s))
    return body


def _vector(a):
    return _ld(1, _packed_floats(np.asarray(a).ravel()))


def encode_model_params(model):
    """bytes of a LOPQModelParams message for `model` (model.py:755-781: missing parts are skipped)."""
    out = b""
    if model.Cs is not None:
        out += _key(1, 0) + _varint(2 * model.Cs[0].shape[1])
    out += _key(2, 0) + _varint(int(model.V)) + _key(3, 0) + _varint(int(model.M))
    out += _key(4, 0) + _varint(int(model.subquantizer_clusters))
    if model.Cs is not None:
        for C in model.Cs:
            out += _ld(5, _matrix(C))
    if model.Rs is not None:
        for Rsplit in model.Rs:
            for R in Rsplit:
                out += _ld(6, _matrix(R))
    if model.mus is not None:
        for musplit in model.mus:
            for mu in musplit:
                out += _ld(7, _vector(mu))
    if model.subquantizers is not None:
        for half in model.subquantizers:
            for sub in half:
                out += _ld(8, _matrix(sub))
    return out


def write_model_params(model, f):
    data = encode_model_params(model)
    if isinstance(f, str):
        with open(f, "wb") as fh:
            fh.write(data)
    else:
        f.write(data)
        f.close()  # the reference closes the handle it is given (model.py:785-786)


def _read_varint(buf, i):
    shift = n = 0
    while True:
        b = buf[i]
        i += 1
        n |= (b & 0x7F) << shift
        if not b & 0x80:
            return n, i
        shift += 7


def _fields(buf):
    i = 0
    while i < len(buf):
        key, i = _read_varint(buf, i)
        field, wire = key >> 3, key & 7
        if wire == 0:
            v, i = _read_varint(buf, i)
        elif wire == 2:
            n, i = _read_varint(buf, i)
            v = buf[i:i + n]
            i += n
        elif wire == 5:
            v = struct.unpack_from("<f", buf, i)[0]
            i += 4
        elif wire == 1:
            v = struct.unpack_from("<d", buf, i)[0]
            i += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wire)
        yield field, wire, v


def _parse_floats_and_shape(buf):
    vals, shape = [], []
    for field, wire, v in _fields(buf):
        if field == 1:
            if wire == 2:
                vals.append(np.frombuffer(v, dtype="<f4"))
            else:
                vals.append(np.array([v], dtype=np.float32))
        elif field == 2:
            if wire == 2:  # packed encoding is also legal for repeated scalars
                j = 0
                while j < len(v):
                    x, j = _read_varint(v, j)
                    shape.append(x)
            else:
                shape.append(v)
    vals = np.concatenate(vals) if vals else np.zeros(0, dtype=np.float32)
    return vals.astype(np.float64), shape  # np.reshape(C.values, C.shape) of python floats -> float64 (model.py:808)


def decode_model_params(data):
    """-> dict(D, V, M, num_subquantizers, Cs, Rs, mus, subs) with lists of arrays in file order."""
    out = {"Cs": [], "Rs": [], "mus": [], "subs": []}
    names = {1: "D", 2: "V", 3: "M", 4: "num_subquantizers"}
    for field, wire, v in _fields(bytes(data)):
        if field in names:
            out[names[field]] = v
        elif field in (5, 6, 8):
            vals, shape = _parse_floats_and_shape(v)
            out[{5: "Cs", 6: "Rs", 8: "subs"}[field]].append(vals.reshape(shape))
        elif field == 7:
            vals, _ = _parse_floats_and_shape(v)
            out["mus"].append(vals)
    return out


def read_model_params(filename):
    """reference: load_proto, model.py:788-820 -> LOPQModel (None if the file cannot be opened)."""
    from .model import LOPQModel
    try:
        with open(filename, "rb") as f:
            d = decode_model_params(f.read())
    except IOError:
        print(filename + ": Could not open file.")
        return None
    halves = lambda arr: [arr[:len(arr) // 2], arr[len(arr) // 2:]]
    Cs = Rs = mus = subs = None
    if d["Cs"]:
        Cs = tuple(d["Cs"])
    if d["Rs"]:
        Rs = tuple(np.stack(h) for h in halves(d["Rs"]))
    if d["mus"]:
        mus = tuple(np.stack(h) for h in halves(d["mus"]))
    if d["subs"]:
        subs = tuple(halves(d["subs"]))
    return LOPQModel(parameters=(Cs, Rs, mus, subs))
