"""export_proto / load_proto: our hand-written LOPQModelParams codec against google.protobuf's own parser
(dynamic descriptor with the schema of the reference's lopq_model_pb2.py) and a full round trip."""
import io

import numpy as np
import pytest

from conftest import load_golden


def _pb_classes():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "lopq_model_test.proto"
    fd.package = "com.flickr.vision.lopq.test"
    T = descriptor_pb2.FieldDescriptorProto
    vec = fd.message_type.add(); vec.name = "Vector"
    f = vec.field.add(); f.name = "values"; f.number = 1; f.type = T.TYPE_FLOAT; f.label = T.LABEL_REPEATED; f.options.packed = True
    mat = fd.message_type.add(); mat.name = "Matrix"
    f = mat.field.add(); f.name = "values"; f.number = 1; f.type = T.TYPE_FLOAT; f.label = T.LABEL_REPEATED; f.options.packed = True
    f = mat.field.add(); f.name = "shape"; f.number = 2; f.type = T.TYPE_UINT32; f.label = T.LABEL_REPEATED
    par = fd.message_type.add(); par.name = "LOPQModelParams"
    for i, n in enumerate(["D", "V", "M", "num_subquantizers"]):
        f = par.field.add(); f.name = n; f.number = i + 1; f.type = T.TYPE_UINT32; f.label = T.LABEL_OPTIONAL
    for n, num, tn in [("Cs", 5, "Matrix"), ("Rs", 6, "Matrix"), ("mus", 7, "Vector"), ("subs", 8, "Matrix")]:
        f = par.field.add(); f.name = n; f.number = num; f.type = T.TYPE_MESSAGE; f.label = T.LABEL_REPEATED
        f.type_name = ".com.flickr.vision.lopq.test." + tn
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    desc = pool.FindMessageTypeByName("com.flickr.vision.lopq.test.LOPQModelParams")
    try:
        return message_factory.GetMessageClass(desc)
    except AttributeError:
        return message_factory.MessageFactory(pool).GetPrototype(desc)


def _model():
    from columbiaimagesearch_amd.lopq import LOPQModel
    z, X, Q = load_golden("tiny")
    nf = int(z["num_fine_splits"])
    subs = tuple([z["subs"][s, j] for j in range(nf)] for s in range(2))
    return LOPQModel(parameters=((z["Cs"][0], z["Cs"][1]), (z["Rs"][0], z["Rs"][1]), (z["mus"][0], z["mus"][1]), subs))


def test_official_parser_reads_our_bytes():
    from columbiaimagesearch_amd.lopq.proto import encode_model_params
    m = _model()
    msg = _pb_classes()()
    msg.ParseFromString(encode_model_params(m))
    assert (msg.D, msg.V, msg.M, msg.num_subquantizers) == (8, 4, 4, 16)
    assert len(msg.Cs) == 2 and len(msg.Rs) == 8 and len(msg.mus) == 8 and len(msg.subs) == 4
    np.testing.assert_allclose(np.reshape(msg.Cs[1].values, msg.Cs[1].shape), m.Cs[1].astype(np.float32))
    np.testing.assert_allclose(np.reshape(msg.Rs[5].values, msg.Rs[5].shape), m.Rs[1][1].astype(np.float32))
    np.testing.assert_allclose(np.array(msg.mus[2].values), m.mus[0][2].astype(np.float32))


def test_we_read_the_official_serialisation_and_round_trip(tmp_path):
    from columbiaimagesearch_amd.lopq import LOPQModel
    from columbiaimagesearch_amd.lopq.proto import encode_model_params
    m = _model()
    msg = _pb_classes()()
    msg.ParseFromString(encode_model_params(m))
    path = str(tmp_path / "model.lopq")
    with open(path, "wb") as f:
        f.write(msg.SerializeToString())  # bytes produced by the official library
    back = LOPQModel.load_proto(path)
    assert back.V == m.V and back.M == m.M and back.subquantizer_clusters == m.subquantizer_clusters
    for s in range(2):
        np.testing.assert_allclose(back.Cs[s], m.Cs[s].astype(np.float32))
        np.testing.assert_allclose(back.Rs[s], m.Rs[s].astype(np.float32))
        np.testing.assert_allclose(back.mus[s], m.mus[s].astype(np.float32))
        for a, b in zip(back.subquantizers[s], m.subquantizers[s]):
            np.testing.assert_allclose(a, b.astype(np.float32))
    m.export_proto(str(tmp_path / "m2.lopq"))
    again = LOPQModel.load_proto(str(tmp_path / "m2.lopq"))
    np.testing.assert_array_equal(again.Rs[0], back.Rs[0])
    assert LOPQModel.load_proto(str(tmp_path / "missing.lopq")) is None


def test_pca_variant_refuses_like_the_reference():
    from columbiaimagesearch_amd.lopq import LOPQModelPCA
    with pytest.raises(NotImplementedError):
        LOPQModelPCA().export_proto(io.BytesIO())
    with pytest.raises(NotImplementedError):
        LOPQModelPCA.load_mat("x.mat")
