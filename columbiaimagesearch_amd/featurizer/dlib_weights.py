"""Weights of dlib's face-recognition ResNet for DLibFaceNet, from what a dlib installation can export.

The reference loads ``dlib_face_recognition_resnet_model_v1.dat`` through dlib itself
(cufacesearch/cufacesearch/featurizer/dlib_featurizer.py:1-4,83).  That file is dlib's private C++ stream
serialisation of ``anet_type`` (not in the reference tree, dlib version unpinned: requirements.txt:3), and dlib's
Python module exposes no weights.  The portable route dlib itself documents is ``net_to_xml``: a six-line C++
program on any machine with dlib turns the ``.dat`` into XML, which this module converts
to the 117 arrays of ``tensor_names()``.  **Unpinned**: no dlib and no ``.dat`` exist in the build image, so the XML
reader is tested only against its own writer (tests/test_weight_formats.py).

net_to_xml layout relied on (dlib/dnn/utilities.h, layers' ``to_xml``): ``<net>`` holds ``<layer idx= type=>`` elements
from the OUTPUT (idx 0 = loss) to the input; computational layers carry one child -- ``<con num_filters= nr= nc=
stride_y= stride_x= ...>``, ``<affine_con>``, ``<fc_no_bias num_outputs=>`` ... -- whose text is ``mat(params)``:
``con``: filters [num_filters][k][nr][nc] then the num_filters biases; ``affine``: gamma[k] then beta[k];
``fc_no_bias``: [num_inputs][num_outputs].
"""
import numpy as np

from .dlibhip_featurizer import tensor_names
from .synthetic import dlib_block_plan


def _param_layers(path):
    import xml.etree.ElementTree as ET
    out = []
    for _, el in ET.iterparse(path, events=("end",)):
        tag = el.tag
        if tag.startswith("con") or tag.startswith("affine") or tag.startswith("fc"):
            vals = np.array((el.text or "").split(), dtype=np.float32)
            out.append((tag, dict(el.attrib), vals))
            el.clear()
    return out[::-1]  # the file runs from the loss layer down to the input


def weights_from_net_xml(path):
    """{name: float32 array} for DLibFaceNet from the XML written by dlib's ``net_to_xml(net, path)``."""
    layers = _param_layers(path)
    plan = dlib_block_plan()
    want = 2 + 4 * len(plan) + 1
    if len(layers) != want:
        raise ValueError("expected %d parameter layers (29 con + 29 affine + fc), found %d" % (want, len(layers)))
    w = {}
    it = iter(layers)

    def take_conv(name, oc, ic, k):
        tag, attr, v = next(it)
        if not tag.startswith("con") or v.size != oc * ic * k * k + oc:
            raise ValueError("layer %s: expected con with %d values, got <%s> with %d" % (name, oc * ic * k * k + oc, tag, v.size))
        w[name + "_w"] = v[:oc * ic * k * k].reshape(oc, ic, k, k).copy()
        w[name + "_b"] = v[oc * ic * k * k:].copy()

    def take_affine(g, b, c):
        tag, attr, v = next(it)
        if not tag.startswith("affine") or v.size != 2 * c:
            raise ValueError("layer %s: expected affine with %d values, got <%s> with %d" % (g, 2 * c, tag, v.size))
        w[g], w[b] = v[:c].copy(), v[c:].copy()

    take_conv("conv0", 32, 3, 7)
    take_affine("aff0_g", "aff0_b", 32)
    for i, (cin, cout, down) in enumerate(plan):
        take_conv("b%da" % i, cout, cin, 3)
        take_affine("b%da_g" % i, "b%da_beta" % i, cout)
        take_conv("b%db" % i, cout, cout, 3)
        take_affine("b%db_g" % i, "b%db_beta" % i, cout)
    tag, attr, v = next(it)
    if not tag.startswith("fc") or v.size != 256 * 128:
        raise ValueError("expected fc_no_bias with %d values, got <%s> with %d" % (256 * 128, tag, v.size))
    w["fc_w"] = np.ascontiguousarray(v.reshape(256, 128).T)  # dlib: out = x . W, W is [inputs][outputs]
    assert sorted(w) == sorted(tensor_names())
    return w


def write_net_xml(w, path):
    """Inverse of weights_from_net_xml in net_to_xml's layout (tests; also documents the expected file)."""
    plan = dlib_block_plan()
    fwd = []

    def conv(name, stride):
        oc, ic, k, _ = w[name + "_w"].shape
        vals = np.concatenate([w[name + "_w"].ravel(), w[name + "_b"].ravel()])
        fwd.append(("con", "num_filters='%d' nr='%d' nc='%d' stride_y='%d' stride_x='%d'" % (oc, k, k, stride, stride), vals))

    def affine(g, b):
        fwd.append(("affine_con", "", np.concatenate([w[g].ravel(), w[b].ravel()])))

    conv("conv0", 2)
    affine("aff0_g", "aff0_b")
    fwd.append(("relu", "", None))
    fwd.append(("max_pool", "nr='3' nc='3' stride_y='2' stride_x='2'", None))
    for i, (cin, cout, down) in enumerate(plan):
        conv("b%da" % i, 2 if down else 1)
        affine("b%da_g" % i, "b%da_beta" % i)
        fwd.append(("relu", "", None))
        conv("b%db" % i, 1)
        affine("b%db_g" % i, "b%db_beta" % i)
        fwd.append(("add_prev1", "", None))
        fwd.append(("relu", "", None))
    fwd.append(("avg_pool", "nr='0' nc='0'", None))
    fwd.append(("fc_no_bias", "num_outputs='128'", np.ascontiguousarray(w["fc_w"].T).ravel()))
    with open(path, "wt") as f:
        f.write("<?xml version='1.0' encoding='ISO-8859-1'?>\n<net>\n<layer idx='0' type='loss'><loss_metric margin='0.04' distance_threshold='0.6'/></layer>\n")
        for idx, (tag, attrs, vals) in enumerate(fwd[::-1], start=1):
            f.write("<layer idx='%d' type='comp'><%s %s>" % (idx, tag, attrs))
            if vals is not None:
                f.write("\n" + " ".join(repr(float(x)) for x in vals.astype(np.float32)) + "\n")
            f.write("</%s></layer>\n" % tag)
        f.write("<layer idx='%d' type='input'><input_rgb_image_sized r='122.782' g='117.001' b='104.298' nr='150' nc='150'/></layer>\n</net>\n" % (len(fwd) + 1))
