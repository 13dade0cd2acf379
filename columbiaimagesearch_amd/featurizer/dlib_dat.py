"""dlib's own stream serialisation of the face-recognition ResNet (``dlib_face_recognition_resnet_model_v1.dat``).

The reference hands ``rec_path`` to ``dlib.face_recognition_model_v1`` (cufacesearch/cufacesearch/featurizer/dlib_featurizer.py:83),
i.e. to ``deserialize(path) >> net`` of ``anet_type``.  This module walks that stream in Python and returns the 117 arrays of
``tensor_names()``.

**Unpinned, and restated from the published dlib sources as the author remembers them** (dlib/serialize.h, dlib/float_details.h,
dlib/cuda/tensor.h, dlib/dnn/core.h, dlib/dnn/layers.h, dlib/dnn/input.h, dlib/dnn/loss.h of the 19.x line): there is no dlib, no
``.dat`` and no network in the build image, so the reader is tested against the writer below only (tests/test_weight_formats.py).  It
therefore checks everything it reads -- version numbers, the layers' name strings, tensor shapes against the architecture -- and stops
with the byte offset and what it expected the moment the stream disagrees; it never guesses.  The ``net_to_xml`` route
(dlib_weights.py) remains the documented fallback.

Stream primitives (serialize.h):
  * integers: one control byte (low nibble = number of value bytes n, bit 7 = negative), then n bytes little-endian magnitude;
  * float / double: ``float_details`` = (int64 mantissa, int16 exponent), value = mantissa * 2**exponent, exponents 32000 / 32001 /
    32002 = +inf / -inf / nan; bool: the character '1' or '0'; std::string: length, then the bytes;
  * ``resizable_tensor``: int version (2), four long long dims (num_samples, k, nr, nc), then the floats as raw little-endian IEEE;
    ``alias_tensor``: int version (1), four dims.
Network structure (dnn/core.h): ``add_loss_layer``: version 1, the loss, the subnetwork; ``add_layer``: version (1 or 2), the SUBNETWORK
first, then the layer's details, three bools, ``x_grad``, ``cached_output`` and (version 2) ``params_grad``; the innermost ``add_layer``
(over the input layer): version (2 or 3), the input layer, details, three bools, ``x_grad``, ``cached_output``, ``grad_final`` and
(version 3) the sample expansion factor; ``add_tag_layer`` / ``add_skip_layer``: version 1, the subnetwork.  So the layers' details
appear from the input to the output, each preceded by nothing but the run of version numbers at the head of the file.
"""
import struct

import numpy as np

from .dlibhip_featurizer import tensor_names
from .synthetic import DLIB_LEVELS

_INF, _NINF, _NAN = 32000, 32001, 32002


# ---- the architecture, outermost layer first (dnn_face_recognition_ex.cpp / dlib's face_recognition.cpp) -----------------------------
def _block(n, stride):  # block<N, affine, stride, SUBNET> = affine<con<N,3,3,1,1, relu<affine<con<N,3,3,stride,stride, SUBNET>>>>>
    return [("affine", n), ("con", n, 3, 1), ("relu",), ("affine", n), ("con", n, 3, stride)]


def _ares(n):  # relu<add_prev1<block<N, affine, 1, tag1<SUBNET>>>>
    return [("relu",), ("add_prev",)] + _block(n, 1) + [("tag",)]


def _ares_down(n):  # relu<add_prev2<avg_pool<2,2,2,2, skip1<tag2<block<N, affine, 2, tag1<SUBNET>>>>>>>
    return [("relu",), ("add_prev",), ("avg_pool", 2, 2, 2, 2), ("skip",), ("tag",)] + _block(n, 2) + [("tag",)]


def anet_layers():
    """The layers of anet_type from the loss down to the input."""
    spec = [("loss_metric",), ("fc_no_bias", 128), ("avg_pool", 0, 0, 1, 1)]
    for n, count, down in DLIB_LEVELS[::-1]:  # level0 (256, one down block) is outermost, level4 (32) innermost
        for b in range(count):
            spec += _ares_down(n) if (down and b == count - 1) else _ares(n)
    spec += [("max_pool", 3, 3, 2, 2), ("relu",), ("affine", 32), ("con", 32, 7, 2), ("input",)]
    return spec


class _In(object):
    def __init__(self, buf):
        self.b, self.p = memoryview(buf), 0

    def fail(self, what):
        raise ValueError("dlib .dat stream, byte %d: %s" % (self.p, what))

    def take(self, n):
        if self.p + n > len(self.b):
            self.fail("the stream ends %d bytes early" % (self.p + n - len(self.b)))
        out = self.b[self.p:self.p + n]
        self.p += n
        return out

    def int(self):
        c = self.take(1)[0]
        n, neg = c & 0x0F, bool(c & 0x80)
        if n == 0 or n > 8 or (c & 0x70):
            self.p -= 1
            self.fail("integer control byte 0x%02x" % c)
        v = int.from_bytes(bytes(self.take(n)), "little")
        return -v if neg else v

    def real(self):
        m, e = self.int(), self.int()
        if e == _INF:
            return float("inf")
        if e == _NINF:
            return float("-inf")
        if e == _NAN:
            return float("nan")
        return float(np.ldexp(float(m), e))

    def bool(self):
        c = bytes(self.take(1))
        if c not in (b"0", b"1"):
            self.p -= 1
            self.fail("bool character %r" % c)
        return c == b"1"

    def string(self):
        n = self.int()
        if n < 0 or n > 256:
            self.fail("string of length %d" % n)
        return bytes(self.take(n)).decode("latin-1")

    def expect_version(self, what, allowed):
        v = self.int()
        if v not in allowed:
            self.fail("%s: version %d, known: %s" % (what, v, list(allowed)))
        return v

    def tensor(self):
        self.expect_version("resizable_tensor", (2,))
        dims = tuple(self.int() for _ in range(4))
        n = 1
        for d in dims:
            if d < 0:
                self.fail("tensor dims %r" % (dims,))
            n *= d
        return np.frombuffer(bytes(self.take(4 * n)), dtype="<f4").reshape(dims) if n else np.zeros(dims, np.float32)

    def alias(self):
        self.expect_version("alias_tensor", (1,))
        return tuple(self.int() for _ in range(4))


def _details(s, layer, out):
    kind = layer[0]
    name = s.string()

    def named(*ok):
        if name not in ok:
            s.fail("layer %r: name string %r, expected one of %r" % (kind, name, ok))

    if kind == "input":
        named("input_rgb_image_sized")
        r, g, b = s.real(), s.real(), s.real()
        nr, nc = s.int(), s.int()
        if (nr, nc) != (150, 150):
            s.fail("input size %dx%d, the face network takes 150x150 chips" % (nr, nc))
        out.append(("input", (r, g, b)))
    elif kind == "con":
        named("con_4", "con_5")
        params = s.tensor()
        nf, nr, nc, sy, sx = (s.int() for _ in range(5))
        s.int(); s.int()  # padding_y, padding_x
        filt, bias = s.alias(), s.alias()
        for _ in range(4):
            s.real()      # learning-rate / weight-decay multipliers
        if name == "con_5":
            s.bool()      # use_bias
        if (nf, nr, nc, sy, sx) != (layer[1], layer[2], layer[2], layer[3], layer[3]):
            s.fail("con layer %r in the stream where the architecture has con<%d,%d,%d,%d,%d>" % ((nf, nr, nc, sy, sx), layer[1], layer[2], layer[2], layer[3], layer[3]))
        out.append(("con", params.ravel(), filt, bias))
    elif kind == "affine":
        named("affine_", "affine_2")
        params = s.tensor()
        gamma, beta = s.alias(), s.alias()
        s.int()  # mode
        if name == "affine_2":
            s.bool()
        if gamma[1] != layer[1] or params.size != 2 * layer[1]:
            s.fail("affine layer over %d channels (%d parameters) where the architecture has %d" % (gamma[1], params.size, layer[1]))
        out.append(("affine", params.ravel()))
    elif kind == "relu":
        named("relu_", "relu_2")
        if name == "relu_2":
            s.bool()
    elif kind in ("max_pool", "avg_pool"):
        named(kind + "_2")
        got = tuple(s.int() for _ in range(4))
        s.int(); s.int()  # paddings
        if got != tuple(layer[1:5]):
            s.fail("%s%r in the stream where the architecture has %r" % (kind, got, tuple(layer[1:5])))
    elif kind == "add_prev":
        named("add_prev_")
    elif kind == "fc_no_bias":
        named("fc_2", "fc_3")
        n_out, n_in = s.int(), s.int()
        params = s.tensor()
        s.alias(); s.alias()
        s.int()  # bias mode
        for _ in range(4):
            s.real()
        if name == "fc_3":
            s.bool()
        if (n_out, n_in) != (128, 256) or params.size != 128 * 256:
            s.fail("fc layer %d -> %d with %d parameters, expected 256 -> 128 without bias" % (n_in, n_out, params.size))
        out.append(("fc", params.ravel()))
    elif kind == "loss_metric":
        named("loss_metric_", "loss_metric_2")
        if name == "loss_metric_2":
            s.real(); s.real()  # margin, distance threshold
    else:
        s.fail("no reader for layer kind %r" % (kind,))


def _net(s, spec, i, out):
    layer = spec[i]
    kind = layer[0]
    if kind == "loss_metric":
        s.expect_version("add_loss_layer", (1,))
        _details(s, layer, out)
        _net(s, spec, i + 1, out)
    elif kind in ("tag", "skip"):
        s.expect_version("add_%s_layer" % kind, (1,))
        _net(s, spec, i + 1, out)
    elif spec[i + 1][0] == "input":
        v = s.expect_version("add_layer over the input layer", (2, 3))
        _details(s, spec[i + 1], out)
        _details(s, layer, out)
        s.bool(); s.bool(); s.bool()
        s.tensor(); s.tensor(); s.tensor()  # x_grad, cached_output, grad_final
        if v >= 3:
            s.int()                         # sample expansion factor
    else:
        v = s.expect_version("add_layer", (1, 2))
        _net(s, spec, i + 1, out)
        _details(s, layer, out)
        s.bool(); s.bool(); s.bool()
        s.tensor(); s.tensor()              # x_grad, cached_output
        if v >= 2:
            s.tensor()                      # params_grad


def weights_from_dat(path_or_bytes):
    """{name: float32 array} for DLibFaceNet from dlib's serialised anet_type (see the module docstring: unpinned)."""
    buf = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray, memoryview)) else open(path_or_bytes, "rb").read()
    s = _In(buf)
    layers = []
    _net(s, anet_layers(), 0, layers)
    if s.p != len(s.b):
        s.fail("%d bytes behind the network" % (len(s.b) - s.p))
    w = {}
    it = iter([l for l in layers if l[0] != "input"])  # forward order: the input's details were read first

    def conv(name, oc, ic, k):
        _, p, filt, bias = next(it)
        if tuple(filt) != (oc, ic, k, k) or p.size != oc * ic * k * k + oc:
            raise ValueError("dlib .dat: %s has filters %r (%d parameters), expected %r + %d biases" % (name, tuple(filt), p.size, (oc, ic, k, k), oc))
        w[name + "_w"] = np.array(p[:oc * ic * k * k], dtype=np.float32).reshape(oc, ic, k, k)
        w[name + "_b"] = np.array(p[oc * ic * k * k:], dtype=np.float32)

    def affine(g, b, c):
        _, p = next(it)
        w[g], w[b] = np.array(p[:c], dtype=np.float32), np.array(p[c:], dtype=np.float32)

    from .synthetic import dlib_block_plan
    conv("conv0", 32, 3, 7)
    affine("aff0_g", "aff0_b", 32)
    for i, (cin, cout, down) in enumerate(dlib_block_plan()):
        conv("b%da" % i, cout, cin, 3)
        affine("b%da_g" % i, "b%da_beta" % i, cout)
        conv("b%db" % i, cout, cout, 3)
        affine("b%db_g" % i, "b%db_beta" % i, cout)
    _, p = next(it)
    w["fc_w"] = np.ascontiguousarray(np.array(p, dtype=np.float32).reshape(256, 128).T)  # dlib: out = x . W, W is [inputs][outputs]
    assert sorted(w) == sorted(tensor_names())
    return w


# ---- the writer: the same statement of the format run backwards (tests; documents the expected file) ---------------------------------
class _Out(object):
    def __init__(self):
        self.parts = []

    def int(self, v):
        v = int(v)
        neg = 0x80 if v < 0 else 0
        mag = -v if v < 0 else v
        raw = mag.to_bytes(max(1, (mag.bit_length() + 7) // 8), "little")
        self.parts.append(bytes([len(raw) | neg]) + raw)

    def real(self, x, digits=53):
        x = float(x)
        if x != x:
            self.int(0); self.int(_NAN); return
        if x in (float("inf"), float("-inf")):
            self.int(0); self.int(_INF if x > 0 else _NINF); return
        m, e = np.frexp(x)
        mant, exp = int(m * (1 << digits)), int(e) - digits
        for _ in range(8):
            if mant & 0xFF or mant == 0:
                break
            mant >>= 8
            exp += 8
        self.int(mant); self.int(exp)

    def bool(self, v):
        self.parts.append(b"1" if v else b"0")

    def string(self, t):
        self.int(len(t)); self.parts.append(t.encode("latin-1"))

    def tensor(self, a):
        a = np.asarray(a, dtype="<f4")
        dims = tuple(a.shape) + (1,) * (4 - a.ndim) if a.size else (0, 0, 0, 0)
        self.int(2)
        for d in dims:
            self.int(d)
        self.parts.append(a.tobytes())

    def alias(self, dims):
        self.int(1)
        for d in dims:
            self.int(d)


def write_dat(w, path=None, layer_version=2, input_layer_version=3):
    """The stream weights_from_dat reads, from the 117 arrays (empty gradient / cache tensors, as a network saved after `clean()`)."""
    from .synthetic import dlib_block_plan
    params = []  # forward order, one entry per con / affine / fc layer

    def conv(name):
        params.append(("con", np.concatenate([w[name + "_w"].ravel(), w[name + "_b"].ravel()]), w[name + "_w"].shape))

    def affine(g, b):
        params.append(("affine", np.concatenate([w[g].ravel(), w[b].ravel()]), None))

    conv("conv0"); affine("aff0_g", "aff0_b")
    for i, _ in enumerate(dlib_block_plan()):
        conv("b%da" % i); affine("b%da_g" % i, "b%da_beta" % i)
        conv("b%db" % i); affine("b%db_g" % i, "b%db_beta" % i)
    params.append(("fc", np.ascontiguousarray(w["fc_w"].T).ravel(), None))
    it = iter(params)
    o = _Out()
    spec = anet_layers()

    def details(layer):
        kind = layer[0]
        if kind == "input":
            o.string("input_rgb_image_sized")
            for v in (122.782, 117.001, 104.298):
                o.real(np.float32(v), 24)
            o.int(150); o.int(150)
        elif kind == "con":
            _, p, shp = next(it)
            o.string("con_4"); o.tensor(p)
            for v in (layer[1], layer[2], layer[2], layer[3], layer[3], 0 if layer[2] == 7 or layer[3] == 2 else 1, 0 if layer[2] == 7 or layer[3] == 2 else 1):
                o.int(v)
            o.alias(shp); o.alias((1, shp[0], 1, 1))
            for v in (1.0, 1.0, 1.0, 0.0):
                o.real(v)
        elif kind == "affine":
            _, p, _ = next(it)
            o.string("affine_"); o.tensor(p); o.alias((1, layer[1], 1, 1)); o.alias((1, layer[1], 1, 1)); o.int(0)
        elif kind == "relu":
            o.string("relu_")
        elif kind in ("max_pool", "avg_pool"):
            o.string(kind + "_2")
            for v in tuple(layer[1:5]) + (0, 0):
                o.int(v)
        elif kind == "add_prev":
            o.string("add_prev_")
        elif kind == "fc_no_bias":
            _, p, _ = next(it)
            o.string("fc_2"); o.int(128); o.int(256); o.tensor(p.reshape(256, 128)); o.alias((256, 128, 1, 1)); o.alias((0, 0, 0, 0)); o.int(1)
            for v in (1.0, 1.0, 1.0, 0.0):
                o.real(v)
        elif kind == "loss_metric":
            o.string("loss_metric_2"); o.real(np.float32(0.04), 24); o.real(np.float32(0.6), 24)

    def net(i):
        layer = spec[i]
        kind = layer[0]
        if kind == "loss_metric":
            o.int(1); details(layer); net(i + 1)
        elif kind in ("tag", "skip"):
            o.int(1); net(i + 1)
        elif spec[i + 1][0] == "input":
            o.int(input_layer_version); details(spec[i + 1]); details(layer)
            o.bool(True); o.bool(True); o.bool(False)
            o.tensor(np.zeros(0)); o.tensor(np.zeros(0)); o.tensor(np.zeros(0))
            if input_layer_version >= 3:
                o.int(1)
        else:
            o.int(layer_version); net(i + 1); details(layer)
            o.bool(True); o.bool(True); o.bool(False)
            o.tensor(np.zeros(0)); o.tensor(np.zeros(0))
            if layer_version >= 2:
                o.tensor(np.zeros(0))

    net(0)
    data = b"".join(o.parts)
    if path is not None:
        with open(path, "wb") as f:
            f.write(data)
    return data
