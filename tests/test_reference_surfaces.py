"""The surfaces cufacesearch actually calls, against objects produced by the real reference (tests/golden/pk, made
by tests/golden/make_golden.py:make_pk): the storer's pickled ``lopq.model`` objects (storer/local.py:58,75,
searcher_lopqhbase.py:113), ``compute_codes_notparallel(data, model)`` (searcher_lopqhbase.py:503), the per-update
codes dict ``{id: [coarse, fine]}`` (:506-512) fed to ``add_codes_from_dict`` (:757), ``add_data``
(lopq/lopq/search.py:94-108), ``add_codes_from_local`` (:245-263), the LMDB searcher with the production
``id_lambda=str`` (searcher_lopqhbase.py:204-206) and the detector branch of ``GenericExtractor.process_buffer``
(generic_extractor.py:236-247)."""
import base64
import os
import pickle
import sys

import numpy as np
import pytest

from conftest import GOLDEN, has_gpu, sha1

PK = os.path.join(GOLDEN, "pk")
TAGS = ["lopq", "lopq_pca"]


@pytest.fixture()
def as_lopq():
    """``import lopq`` resolves to the HIP package for the duration of a test (what INTEGRATION.md section 1 does)."""
    saved = {k: sys.modules.get(k) for k in ("lopq", "lopq.model", "lopq.search", "lopq.utils")}
    import columbiaimagesearch_amd.lopq as L
    L.install_as_lopq()
    yield L
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def _inputs(tag):
    import golden_inputs as gi
    X, Q = gi.pk_inputs()
    z = dict(np.load(os.path.join(PK, "expected.npz")))
    assert str(z["inputs_sha1"]) == sha1(X) + sha1(Q)
    if tag == "lopq":
        X, Q = np.ascontiguousarray(X[:, :16]), np.ascontiguousarray(Q[:, :16])
    return X, Q, z


def _det_ids(n):
    return ["%040x_%d" % (i * 2654435761 % (1 << 61), i % 3) for i in range(n)]


def _load(name):
    with open(os.path.join(PK, name), "rb") as f:
        return pickle.load(f, encoding="latin1")  # SURVEY.md section 8b gotcha iii


# ---- CPU: the pickles resolve to our classes with the reference's attribute set; the oracle agrees with them ------
@pytest.mark.parametrize("tag", TAGS)
def test_pickled_reference_model_loads_as_our_class(as_lopq, tag):
    m = _load("model_%s.pkl" % tag)
    X, Q, z = _inputs(tag)
    want_cls = as_lopq.LOPQModelPCA if tag == "lopq_pca" else as_lopq.LOPQModel
    assert type(m) is want_cls
    assert sorted(m.__dict__) == [str(a) for a in z["%s_attrs" % tag]]  # exactly the reference's attributes
    assert (m.V, m.M, m.subquantizer_clusters, m.num_coarse_splits, m.num_fine_splits) == (4, 4, 16, 2, 2)
    # and pickling our object again keeps that attribute set (no device handle leaks into the pickle)
    m2 = pickle.loads(pickle.dumps(m, protocol=2))
    assert sorted(m2.__dict__) == sorted(m.__dict__)


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_matches_pickled_reference_codes(as_lopq, tag):
    from oracle import lopq_oracle as O
    m = _load("model_%s.pkl" % tag)
    X, Q, z = _inputs(tag)
    om = O.OracleModel(m.Cs, m.Rs, m.mus, m.subquantizers, getattr(m, "pca_P", None), getattr(m, "pca_mu", None),
                       getattr(m, "renorm", False))
    coarse, fine = O.compute_codes(om, X)
    np.testing.assert_array_equal(coarse, z["%s_coarse" % tag])
    np.testing.assert_array_equal(fine, z["%s_fine" % tag])
    codes = _load("codes_%s.pkl" % tag)
    ids = _det_ids(len(X))
    assert len(codes) == len(X)
    for i in (0, 1, 17, len(X) - 1):
        c, f = codes[ids[i]]
        assert tuple(int(v) for v in c) == tuple(coarse[i]) and tuple(int(v) for v in f) == tuple(fine[i])


def test_mat_file_written_by_the_reference(as_lopq, tmp_path):
    """The .mat exchange format (lopq/lopq/model.py:712-746): load_mat reads the file the REFERENCE's export_mat wrote
    (tests/golden/pk/model_lopq.mat) into the parameters of the reference's pickled model, and export_mat writes the
    same arrays under the same names."""
    from scipy.io import loadmat
    ref_path = os.path.join(PK, "model_lopq.mat")
    m = as_lopq.LOPQModel.load_mat(ref_path)
    want = _load("model_lopq.pkl")
    assert (m.V, m.M, m.subquantizer_clusters) == (want.V, want.M, want.subquantizer_clusters)
    for s in range(2):
        np.testing.assert_array_equal(m.Cs[s], want.Cs[s])
        np.testing.assert_array_equal(m.Rs[s], want.Rs[s])
        np.testing.assert_array_equal(m.mus[s], want.mus[s])
        assert len(m.subquantizers[s]) == len(want.subquantizers[s])
        for a, b in zip(m.subquantizers[s], want.subquantizers[s]):
            np.testing.assert_array_equal(a, b)
    out = str(tmp_path / "ours.mat")
    want.export_mat(out)  # `want` is OUR class (install_as_lopq) holding the reference's parameters
    ours, ref = loadmat(out), loadmat(ref_path)
    assert sorted(k for k in ours if not k.startswith("__")) == sorted(k for k in ref if not k.startswith("__"))
    for k in ("Cs", "Rs", "mus", "subs", "V", "M"):
        assert ours[k].shape == ref[k].shape and ours[k].dtype == ref[k].dtype, k
        np.testing.assert_array_equal(ours[k], ref[k])


def test_lmdb_searcher_hands_str_to_id_lambda():
    """the production searcher is LOPQSearcherLMDB(model, path, id_lambda=str); ids are sha1
    strings.  get_cell is host-side, so this part runs without a GPU."""
    from columbiaimagesearch_amd.lopq.search import LOPQSearcherLMDB

    class M(object):
        V, M = 4, 4
    s = LOPQSearcherLMDB(M(), None, id_lambda=str)
    ids = ["da39a3ee5e6b4b0d3255bfef95601890afd80709", "0a4d55a8d778e5022fab701977c5d840bbc486d0_12_34_56_78"]
    s.add_codes([((1, 2), (0, 1, 2, 3)), ((1, 2), (3, 2, 1, 0))], ids)
    cell = s.get_cell((1, 2))
    assert [i for i, _ in cell] == sorted(ids)  # key (byte) order, and plain str -- not "b'...'"
    assert all(type(i) is str for i, _ in cell)
    assert cell[0][1].fine == (3, 2, 1, 0) and s.get_nb_indexed() == 2


# ---- GPU ---------------------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


def _check_searches(searcher, tag, Q, z, ids, by_position=True):
    for quota, limit in ((20, 10), (500, 50)):
        key = "%s_q%d_l%d" % (tag, quota, limit)
        for qi, q in enumerate(Q):
            res, visited = searcher.search(q, quota=quota, limit=limit, with_dists=True)
            want = z[key + "_ids"][qi]
            want = want[want >= 0]
            assert visited == int(z[key + "_visited"][qi])
            assert [r.id for r in res] == [ids[p] for p in want]
            np.testing.assert_allclose([r.dist for r in res], z[key + "_dists"][qi][:len(want)], rtol=1e-9, atol=1e-12)


@gpu
@pytest.mark.parametrize("tag", TAGS)
def test_compute_codes_notparallel_on_pickled_model(as_lopq, tag):
    """searcher_lopqhbase.py:497-512 line by line, with `lopq` = this package: the codes dict equals the one the
    reference pickled -- keys, tuple lengths, values and the numpy scalar type predict_cluster picks."""
    from lopq.utils import compute_codes_notparallel, compute_codes_parallel
    m = _load("model_%s.pkl" % tag)
    X, Q, z = _inputs(tag)
    det_ids = _det_ids(len(X))
    data = [x for x in X]  # the reference passes a LIST of feature arrays (:485-487)
    codes = compute_codes_notparallel(data, m)
    codes_dict = dict()
    for i, code in enumerate(codes):
        codes_dict[det_ids[i]] = [code.coarse, code.fine]
    want = _load("codes_%s.pkl" % tag)
    assert codes_dict.keys() == want.keys()
    for k in want:
        assert codes_dict[k][0] == tuple(want[k][0]) and codes_dict[k][1] == tuple(want[k][1])
    k = det_ids[5]
    assert all(type(v) is type(w) for v, w in zip(codes_dict[k][0], want[k][0]))  # np.uint8 coarse ids
    assert all(type(v) is type(w) for v, w in zip(codes_dict[k][1], want[k][1]))
    par = list(compute_codes_parallel(data[:100], m, 4))
    assert par == codes[:100]
    one = m.predict(X[7])  # the per-vector call compute_partition makes
    assert one == codes[7] and type(one).__name__ == "LOPQCode"


@gpu
@pytest.mark.parametrize("tag", TAGS)
def test_cold_start_from_pickled_codes_dict(as_lopq, tag):
    """searcher_lopqhbase.py:655-770 (load_codes): model from the storer, codes dicts from the storer, add_codes_from_dict,
    then search_from_feats' call `search(feat, quota, limit, with_dists=True)` -- results of the real reference."""
    from lopq import LOPQSearcher
    m = _load("model_%s.pkl" % tag)
    X, Q, z = _inputs(tag)
    s = LOPQSearcher(m)
    s.add_codes_from_dict(_load("codes_%s.pkl" % tag))
    assert s.get_nb_indexed() == int(z["%s_nb_indexed" % tag])
    _check_searches(s, tag, Q, z, _det_ids(len(X)))
    res, _ = s.search(Q[0], quota=20, limit=10, with_dists=False)
    assert res[0]._fields == ("id", "code")
    code = res[0].code
    want = _load("codes_%s.pkl" % tag)[res[0].id]
    assert tuple(code.coarse) == tuple(want[0]) and tuple(code.fine) == tuple(want[1])


@gpu
@pytest.mark.parametrize("tag", TAGS)
def test_add_data_and_add_codes_from_local(as_lopq, tag, tmp_path):
    from lopq import LOPQSearcher
    m = _load("model_%s.pkl" % tag)
    X, Q, z = _inputs(tag)
    ids = _det_ids(len(X))
    # insertion order of the expected results = iteration order of the codes dict = det_ids order (py3 dicts)
    s = LOPQSearcher(m)
    s.add_data(X, ids=ids, num_procs=4)
    assert s.get_nb_indexed() == len(X)
    _check_searches(s, tag, Q, z, ids)
    # default ids of add_data are positions
    s2 = LOPQSearcher(m)
    s2.add_data(X)
    _check_searches(s2, tag, Q, z, list(range(len(X))))
    # the text format of add_codes_from_local: "id<TAB>[[c0, c1], [f0, ...]]", one file or Spark part-* files
    codes = _load("codes_%s.pkl" % tag)
    d = tmp_path / "rdd"
    d.mkdir()
    half = len(ids) // 2
    for name, part in (("part-00000", ids[:half]), ("part-00001", ids[half:])):
        with open(str(d / name), "wt") as f:
            for k in part:
                f.write("%s\t%s\n" % (k, [[int(v) for v in codes[k][0]], [int(v) for v in codes[k][1]]]))
    s3 = LOPQSearcher(m)
    s3.add_codes_from_local(str(d))
    assert s3.get_nb_indexed() == len(ids)
    _check_searches(s3, tag, Q, z, ids)


@gpu
def test_lmdb_searcher_string_ids_end_to_end(as_lopq):
    """id_lambda=str with sha1-like ids through search(): ids come back as the caller's strings, ranking (ties in key
    order, last write wins) equals the source restatement of the LMDB searcher."""
    from lopq.search import LOPQSearcherLMDB
    from oracle import lopq_oracle as O
    tag = "lopq_pca"
    m = _load("model_%s.pkl" % tag)
    X, Q, z = _inputs(tag)
    codes = _load("codes_%s.pkl" % tag)
    ids = _det_ids(len(X))
    s = LOPQSearcherLMDB(m, None, id_lambda=str)
    s.add_codes_from_dict(codes)
    s.add_codes([codes[ids[3]]] * 5, ["dup_%d" % i for i in range(5)])  # equal distances: key order decides
    s.add_codes([codes[ids[10]]], [ids[4]])                              # same cell or not: a put() of an existing key
    om = O.OracleModel(m.Cs, m.Rs, m.mus, m.subquantizers, m.pca_P, m.pca_mu, m.renorm)
    oi = O.OracleKeyOrderIndex(om, id_lambda=str)
    oi.add_codes([codes[k] for k in ids], ids)
    oi.add_codes([codes[ids[3]]] * 5, ["dup_%d" % i for i in range(5)])
    oi.add_codes([codes[ids[10]]], [ids[4]])
    assert s.get_nb_indexed() == oi.nb_indexed
    for q in list(Q) + [X[3]]:
        res, visited = s.search(q, quota=200, limit=40, with_dists=True)
        wres, wvis = oi.search(q, quota=200, limit=40, with_dists=True)
        assert visited == wvis
        assert all(type(r.id) is str for r in res)
        assert [r.id for r in res] == [w[0] for w in wres]
        np.testing.assert_allclose([r.dist for r in res], [w[2] for w in wres], rtol=1e-9, atol=1e-12)


@gpu
def test_extractor_detector_rows(tmp_path):
    """generic_extractor.py:236-247: one column per detection, `<extr_str>_<l>_<t>_<r>_<b>_<score>`, value = base64 of the
    L2-normalised float64 descriptor; an image without detections keeps processed = "0"."""
    import io
    from PIL import Image
    from columbiaimagesearch_amd.extractor.generic_extractor import GenericExtractor
    from columbiaimagesearch_amd.featurizer.featsio import featB64decode
    from oracle import dlib_oracle as DO
    w = DO.synthetic_weights(2)
    wpath = str(tmp_path / "dlib_w.npz")
    np.savez(wpath, **w)

    class FakeDetector(object):
        def detect_from_buffer_noinfos(self, img_buffer, up_sample=1):
            img = np.asarray(Image.open(io.BytesIO(img_buffer)).convert("RGB"))
            if img.shape[0] < 200:
                return img, []
            return img, [{"left": 10, "top": 20, "right": 160, "bottom": 170, "score": 1.25},
                         {"left": 40, "top": 30, "right": 190, "bottom": 180, "score": 0.5}]

    def chip_fn(img, bbox):
        return img[bbox["top"]:bbox["bottom"], bbox["left"]:bbox["right"], :]

    conf = {"DLIBFEAT_pred_path": "unused", "DLIBFEAT_rec_path": wpath}
    ex = GenericExtractor("dlib", "dlib", "face", "ext", "DLIBFEAT_", conf, detector=FakeDetector())
    ex.featurizer.chip_fn = chip_fn
    rs = np.random.RandomState(3)
    bufs = []
    for hw in (240, 120):
        b = io.BytesIO()
        Image.fromarray(rs.randint(0, 256, size=(hw, 260, 3)).astype(np.uint8)).save(b, format="PNG")
        bufs.append(b.getvalue())
    rows = ex.process_batch(bufs + [b"not an image"])
    base = "ext:dlib_feat_dlib_face"
    assert sorted(rows[0]) == sorted([base + "_processed", base + "_10_20_160_170_1.25", base + "_40_30_190_180_0.5"])
    assert rows[0][base + "_processed"] == "1"
    assert rows[1] == {base + "_processed": "0"}
    assert rows[2] == {base + "_failed": "1"}
    img = np.asarray(Image.open(io.BytesIO(bufs[0])).convert("RGB"))
    chips = np.stack([img[20:170, 10:160], img[30:180, 40:190]])
    want = DO.forward_torch(chips, w).astype(np.float64)
    want /= np.linalg.norm(want, axis=1, keepdims=True)
    for k, ref in zip((base + "_10_20_160_170_1.25", base + "_40_30_190_180_0.5"), want):
        got = featB64decode(rows[0][k], "dlib")
        assert got.dtype == np.float64 and got.shape == (128,)
        np.testing.assert_allclose(got, ref, atol=2e-4)
    assert ex.process_buffer(bufs[0]) == rows[0]  # the reference's per-image entry point
    # the chips of many images go through the network together (3 images x 2 detections in batches of 4 chips): same rows
    ex.chip_batch = 4
    many = ex.process_batch([bufs[0], bufs[1], bufs[0], b"x", bufs[0]])
    assert many[0] == rows[0] and many[2] == rows[0] and many[4] == rows[0] and many[1] == rows[1] and many[3] == rows[2]
    # Python-2 column names: the reference's "{}".format(score) printed 12 significant digits
    from columbiaimagesearch_amd.extractor.generic_extractor import get_bbox_str
    assert get_bbox_str({"left": 1, "top": 2, "right": 3, "bottom": 4, "score": 0.1 + 0.2}) == "1_2_3_4_0.3"
    assert get_bbox_str({"left": 1, "top": 2, "right": 3, "bottom": 4, "score": 2.0}) == "1_2_3_4_2.0"


def _lmdb_fixture():
    from test_lopq_hip_parity import hip_model
    from conftest import load_golden
    z, X, Q = load_golden("c2")
    return hip_model(z), z, Q


def test_lmdb_order_index_survives_a_restart(tmp_path):
    """LOPQSearcherLMDB(lmdb_path=...) re-opens what an earlier process stored (lopq/lopq/search.py:416-417, :445-470): same
    keys (cell as 2 x uint16 + bytes(id)), same values, put = last write wins, cells read back in key order.  Host logic
    only (no GPU): get_cell / nb_indexed before == after a reopen; a torn tail record is dropped; compaction keeps the content."""
    from columbiaimagesearch_amd.lopq import LOPQSearcherLMDB
    from columbiaimagesearch_amd.lopq import kvlog
    m, z, Q = _lmdb_fixture()
    n = 1500
    coarse, fine = z["coarse"][:n], z["fine"][:n]
    codes = [((int(c[0]), int(c[1])), tuple(int(v) for v in f)) for c, f in zip(coarse, fine)]
    ids = ["sha1_%05d" % (i * 7919 % n) for i in range(n)]  # production ids are strings; byte order != insertion order
    path = str(tmp_path / "lmdb_index")
    s = LOPQSearcherLMDB(m, path, id_lambda=str)
    s.add_codes(codes[:1000], ids[:1000])
    s.add_codes(codes[1000:], ids[1000:])
    s.add_codes([codes[5]], [ids[3]])            # an existing key gets a new value (last write wins) -- in the same cell or not
    cells = sorted({c for c, _ in codes})
    before = {c: s.get_cell(c) for c in cells}
    nb = s.get_nb_indexed()
    assert s._log is not None and os.path.exists(os.path.join(path, kvlog.FILE_NAME))
    key0 = s.encode_cell(codes[0][0]) + ids[0].encode()
    raw = open(os.path.join(path, kvlog.FILE_NAME), "rb").read()
    assert key0 in raw and s.encode_fine_codes(codes[0][1]) in raw   # the reference's key / value bytes, verbatim
    s.close()
    s2 = LOPQSearcherLMDB(m, path, id_lambda=str)
    assert s2.get_nb_indexed() == nb
    assert {c: s2.get_cell(c) for c in cells} == before
    for c in cells[:5]:
        assert [i for i, _ in before[c]] == sorted(i for i, _ in before[c])  # key (byte) order inside a cell
    s2.close()
    # a crash in the middle of an add_codes call leaves a torn group: the WHOLE call is dropped (the reference's LMDB write
    # transaction, search.py:445-467, is all-or-nothing), everything before it is kept
    import struct
    import zlib
    rec = lambda k, v: struct.pack("<II", len(k), len(v)) + k + v
    payload = rec(s2.encode_cell((0, 0)) + b"torn_a", b"\x01" * m.M) + rec(s2.encode_cell((0, 0)) + b"torn_b", b"\x02" * m.M)
    head = kvlog.HEADER.pack(kvlog.GROUP_TAG, 2, len(payload), zlib.crc32(payload) & 0xFFFFFFFF, 0)
    group = head + struct.pack("<I", zlib.crc32(head) & 0xFFFFFFFF) + payload
    with open(os.path.join(path, kvlog.FILE_NAME), "ab") as f:
        f.write(group[:-5])  # the first record of the call is complete on disk, the second is not
    s3 = LOPQSearcherLMDB(m, path, id_lambda=str)
    assert s3._log.dropped_torn_bytes == len(group) - 5
    assert s3.get_nb_indexed() == nb and {c: s3.get_cell(c) for c in cells} == before
    # many overwrites -> close() compacts the log to one record per live key
    for _ in range(3):
        s3.add_codes(codes[:1000], ids[:1000])
    size_before = os.path.getsize(os.path.join(path, kvlog.FILE_NAME))
    s3.close()
    assert os.path.getsize(os.path.join(path, kvlog.FILE_NAME)) < size_before / 2
    s4 = LOPQSearcherLMDB(m, path, id_lambda=str)
    assert s4.get_nb_indexed() == nb and {c: s4.get_cell(c) for c in cells} == before
    s4.close()
    # damage in the MIDDLE of the log is not a torn tail: it raises instead of silently dropping what follows
    s4 = LOPQSearcherLMDB(m, path, id_lambda=str)
    s4.add_codes(codes[:3], ["later_%d" % i for i in range(3)])
    s4.close()
    raw = bytearray(open(os.path.join(path, kvlog.FILE_NAME), "rb").read())
    raw[len(kvlog.MAGIC) + kvlog.HEADER_BYTES + 12] ^= 0xFF   # a payload byte of the first group
    open(os.path.join(path, kvlog.FILE_NAME), "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="damaged"):
        LOPQSearcherLMDB(m, path, id_lambda=str)
    raw[len(kvlog.MAGIC) + kvlog.HEADER_BYTES + 12] ^= 0xFF
    raw[len(kvlog.MAGIC) + 9] ^= 0xFF                            # a size byte of the first group's HEADER: caught by the header's own checksum
    open(os.path.join(path, kvlog.FILE_NAME), "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="header at byte"):
        LOPQSearcherLMDB(m, path, id_lambda=str)


def test_kvlog_transactions_are_bounded_groups_and_all_or_nothing(tmp_path):
    """lopq/kvlog.py: sizes are 64-bit and a transaction is written as BOUNDED groups (here 200 bytes, so a 40-record
    call spans several): a crash after some groups of a call drops the whole call; a zero-filled tail (file extended, data never
    written) is a torn tail, not valid empty groups; compaction streams; a CISKV2 log is migrated; an add_codes call that raises
    half way leaves memory and disk as they were (the reference's `with env.begin(write=True)`, lopq/lopq/search.py:459-467)."""
    import struct
    import zlib
    from columbiaimagesearch_amd.lopq import kvlog
    d = str(tmp_path / "log")
    log = kvlog.KVLog(d, group_bytes=200)
    a = [(b"\x00\x00\x01\x00key_%03d" % i, bytes([i % 251] * 8)) for i in range(40)]
    b = [(b"\x00\x00\x02\x00key_%03d" % i, bytes([(i + 7) % 251] * 8)) for i in range(40)]
    log.append(iter(a))
    size_a = os.path.getsize(log.path)
    log.append(iter(b))
    raw = open(log.path, "rb").read()
    n_groups = raw.count(kvlog.GROUP_TAG)
    assert n_groups >= 8 and log.records == 80
    assert kvlog.KVLog(d, group_bytes=200).load() == a + b
    # cut inside the LAST group of transaction b: every group of b goes, a stays
    open(log.path, "wb").write(raw[:-5])
    l2 = kvlog.KVLog(d, group_bytes=200)
    assert l2.load() == a and l2.dropped_torn_bytes == len(raw) - 5 - size_a and os.path.getsize(log.path) == size_a
    # cut exactly after a complete group in the middle of b (the transaction's closing group never arrived)
    second = raw.index(kvlog.GROUP_TAG, size_a + 4)
    open(log.path, "wb").write(raw[:second])
    l3 = kvlog.KVLog(d, group_bytes=200)
    assert l3.load() == a and os.path.getsize(log.path) == size_a
    # a zero-filled tail is a torn tail
    open(log.path, "wb").write(raw[:size_a] + b"\x00" * 5000)
    l4 = kvlog.KVLog(d, group_bytes=200)
    assert l4.load() == a and l4.dropped_torn_bytes == 5000 and os.path.getsize(log.path) == size_a
    # compaction: bounded groups again, same content, last write wins is the caller's job
    l4.compact(iter(a[:10] + b[:10]))
    assert kvlog.KVLog(d, group_bytes=200).load() == a[:10] + b[:10]
    # an append whose items cannot be written leaves nothing behind
    size0 = os.path.getsize(log.path)
    with pytest.raises(TypeError):
        l4.append(iter(a[:30] + [(b"k", None)]))
    assert os.path.getsize(log.path) == size0 and kvlog.KVLog(d, group_bytes=200).load() == a[:10] + b[:10]
    # the previous format (32-bit groups) is read and rewritten
    d2 = str(tmp_path / "old")
    os.makedirs(d2)
    rec = lambda k, v: struct.pack("<II", len(k), len(v)) + k + v
    payload = b"".join(rec(k, v) for k, v in a[:5])
    open(os.path.join(d2, kvlog.FILE_NAME), "wb").write(kvlog.MAGIC_V2 + struct.pack("<III", 5, len(payload), zlib.crc32(payload) & 0xFFFFFFFF) + payload)
    l5 = kvlog.KVLog(d2)
    assert l5.load() == a[:5] and open(l5.path, "rb").read().startswith(kvlog.MAGIC)
    # LOPQSearcherLMDB.add_codes: a malformed code in the middle of a call -> nothing of the call is applied or stored
    from columbiaimagesearch_amd.lopq import LOPQSearcherLMDB
    m, z, Q = _lmdb_fixture()
    path = str(tmp_path / "lmdb_index")
    s = LOPQSearcherLMDB(m, path, id_lambda=str)
    good = [((1, 2), tuple(range(m.M))), ((3, 4), tuple(range(m.M)))]
    s.add_codes(good, ["x", "y"])
    size1 = os.path.getsize(os.path.join(path, kvlog.FILE_NAME))
    with pytest.raises((TypeError, ValueError)):
        s.add_codes(good + [((5, "not a cluster"), tuple(range(m.M)))], ["p", "q", "r"])
    assert s.get_nb_indexed() == 2 and os.path.getsize(os.path.join(path, kvlog.FILE_NAME)) == size1
    assert s.get_cell((1, 2))[0][0] == "x" and s.get_cell((5, 0)) == []
    s.close()
    s2 = LOPQSearcherLMDB(m, path, id_lambda=str)
    assert s2.get_nb_indexed() == 2
    s2.close()


def test_lmdb_file_walker_round_trip(tmp_path):
    """lopq/lmdb_read.py: the B+tree walk of a data.mdb against the module's own writer (no LMDB here: unpinned) -- keys in byte order
    over several leaf pages and two branch levels, values on overflow pages, the newer of the two meta pages, entries count, damage."""
    from columbiaimagesearch_amd.lopq import lmdb_read as LR
    import array
    rs = np.random.RandomState(3)
    items = {}
    for i in range(30000):
        cell = array.array("H", [int(rs.randint(0, 16)), int(rs.randint(0, 16))]).tobytes()
        items[cell + str(int(rs.randint(0, 10**9))).encode()] = bytes(rs.randint(0, 256, size=8).astype(np.uint8))
    items[b"\x00\x00\x00\x00big"] = bytes(rs.randint(0, 256, size=9000).astype(np.uint8))  # three overflow pages
    counts, depth = LR.write_env(str(tmp_path / "db"), list(items.items()))
    assert depth >= 3 and counts["leaf"] > 100
    env = LR.Env(str(tmp_path / "db"))
    got = list(env.items(b"index"))
    assert [k for k, _ in got] == sorted(items) and all(items[k] == v for k, v in got)
    assert env.entries(b"index") == len(items) and env.meta["txnid"] == 7 and list(env.items(b"other")) == []
    LR.write_env(str(tmp_path / "empty"), [])
    assert list(LR.Env(str(tmp_path / "empty")).items()) == []
    raw = bytearray(open(str(tmp_path / "db" / "data.mdb"), "rb").read())
    raw[5 * 4096] ^= 0xFF  # a page that carries another page's number
    open(str(tmp_path / "db" / "data.mdb"), "wb").write(bytes(raw))
    with pytest.raises(LR.LMDBFormatError, match="page 5"):
        list(LR.Env(str(tmp_path / "db")).items())


def test_existing_lmdb_index_opens_read_only_without_the_lmdb_module(tmp_path):
    """A directory that holds the reference's LMDB files is opened through lopq/lmdb_read.py when py-lmdb is missing: the same cells in the
    same key order as a searcher fed the same items, inserts refused (host logic only)."""
    from columbiaimagesearch_amd.lopq import LOPQSearcherLMDB, lmdb_read
    try:
        import lmdb  # noqa: F401
        pytest.skip("py-lmdb is installed: the directory is opened by LMDB itself")
    except ImportError:
        pass
    m, z, Q = _lmdb_fixture()
    n = 2000
    codes = [((int(c[0]), int(c[1])), tuple(int(v) for v in f)) for c, f in zip(z["coarse"][:n], z["fine"][:n])]
    ids = ["sha1_%05d" % (i * 7919 % n) for i in range(n)]
    ref = LOPQSearcherLMDB(m, str(tmp_path / "log"), id_lambda=str)
    ref.add_codes(codes, ids)
    lmdb_read.write_env(str(tmp_path / "mdb"), [(ref.encode_cell(c[0]) + i.encode(), ref.encode_fine_codes(c[1])) for c, i in zip(codes, ids)])
    ro = LOPQSearcherLMDB(m, str(tmp_path / "mdb"), id_lambda=str)
    assert ro.get_nb_indexed() == ref.get_nb_indexed() == n
    for cell in sorted({c for c, _ in codes}):
        assert ro.get_cell(cell) == ref.get_cell(cell)
    with pytest.raises(ImportError, match="read-only"):
        ro.add_codes(codes[:1], ["x"])
    assert sorted(os.listdir(str(tmp_path / "mdb"))) == ["data.mdb"]
    ro.close(); ref.close()


def test_lmdb_path_never_shadows_an_existing_lmdb_index(tmp_path):
    """an lmdb_path that already holds the reference's LMDB files (data.mdb) must not be opened as an empty log when
    the lmdb module is missing, and a directory with both stores is refused (lopq/lopq/search.py:416-417 opens what is there)."""
    from columbiaimagesearch_amd.lopq import LOPQSearcherLMDB
    from columbiaimagesearch_amd.lopq import kvlog
    m, z, Q = _lmdb_fixture()
    try:
        import lmdb  # noqa: F401
        have_lmdb = True
    except ImportError:
        have_lmdb = False
    p1 = tmp_path / "only_mdb"
    p1.mkdir()
    (p1 / "data.mdb").write_bytes(b"\0" * 64)
    if not have_lmdb:
        # (round 4: such a directory is walked by lopq/lmdb_read.py; 64 zero bytes are not an LMDB file)
        with pytest.raises(ValueError, match="no valid meta page"):
            LOPQSearcherLMDB(m, str(p1), id_lambda=str)
        assert not (p1 / kvlog.FILE_NAME).exists()
    p2 = tmp_path / "both"
    p2.mkdir()
    (p2 / "data.mdb").write_bytes(b"\0" * 64)
    (p2 / kvlog.FILE_NAME).write_bytes(kvlog.MAGIC)
    with pytest.raises(RuntimeError, match="both"):
        LOPQSearcherLMDB(m, str(p2), id_lambda=str)
    # a log of the first format (bare records) is read and rewritten in the grouped one
    p3 = tmp_path / "v1"
    p3.mkdir()
    import struct
    k = LOPQSearcherLMDB.encode_cell((1, 2)) + b"abc"
    (p3 / kvlog.FILE_NAME).write_bytes(kvlog.MAGIC_V1 + struct.pack("<II", len(k), m.M) + k + bytes(range(m.M)))
    s = LOPQSearcherLMDB(m, str(p3), id_lambda=str)
    assert s.get_nb_indexed() == 1 and s.get_cell((1, 2))[0][0] == "abc"
    s.close()
    assert (p3 / kvlog.FILE_NAME).read_bytes().startswith(kvlog.MAGIC)
    assert LOPQSearcherLMDB(m, str(p3), id_lambda=str).get_nb_indexed() == 1


@gpu
def test_lmdb_order_index_cold_start_answers_like_the_live_one(tmp_path):
    from columbiaimagesearch_amd.lopq import LOPQSearcherLMDB
    m, z, Q = _lmdb_fixture()
    n = 3000
    codes = [((int(c[0]), int(c[1])), tuple(int(v) for v in f)) for c, f in zip(z["coarse"][:n], z["fine"][:n])]
    ids = ["%040x" % (i * 2654435761 % (1 << 61)) for i in range(n)]
    path = str(tmp_path / "ix")
    live = LOPQSearcherLMDB(m, path, id_lambda=str)
    live.add_codes(codes, ids)
    want = [live.search(Q[i], quota=200, limit=30, with_dists=True) for i in range(5)]
    live.close()
    cold = LOPQSearcherLMDB(m, path, id_lambda=str)   # a new process would do exactly this
    assert cold.get_nb_indexed() == n
    for i in range(5):
        res, visited = cold.search(Q[i], quota=200, limit=30, with_dists=True)
        assert visited == want[i][1] and [(r.id, r.code, r.dist) for r in res] == [(r.id, r.code, r.dist) for r in want[i][0]]
    cold.close()


@gpu
def test_lmdb_order_index_refreshes_incrementally_when_new_keys_sort_last(tmp_path):
    """Round 4: after add_codes the key-ordered GPU index takes only the NEW keys when every one of them sorts behind the last key of
    its cell (ids that grow with time), and is rebuilt otherwise (an earlier key, or a replaced value); either way it answers like a
    searcher built from scratch over the same store (lopq/lopq/search.py:445-499: put = last write wins, cells in key order)."""
    from columbiaimagesearch_amd.lopq import LOPQSearcherLMDB
    m, z, Q = _lmdb_fixture()
    n = 2400
    codes = [((int(c[0]), int(c[1])), tuple(int(v) for v in f)) for c, f in zip(z["coarse"][:n], z["fine"][:n])]
    ids = ["%08d" % i for i in range(n)]  # zero-padded: byte order == numeric order

    def fresh(upto, extra=()):
        s = LOPQSearcherLMDB(m, None, id_lambda=str)
        s.add_codes(codes[:upto], ids[:upto])
        for c, i in extra:
            s.add_codes([c], [i])
        return s

    live = LOPQSearcherLMDB(m, None, id_lambda=str)
    live.add_codes(codes[:1500], ids[:1500])
    live.search(Q[0], quota=300, limit=20)                      # builds the device index
    dev0 = live._dev
    live.add_codes(codes[1500:1560], ids[1500:1560])            # later keys only: incremental, small enough for the cells' slack
    live.search(Q[0], quota=300, limit=20)
    assert live._dev is dev0 and live._dev.insert_counters()[0] >= 1   # same device index, the batch went in place
    live.add_codes(codes[1560:2000], ids[1560:2000])            # later keys again (a burst: in place or a device-side rebuild, same index object)
    got = [live.search(Q[i], quota=300, limit=20, with_dists=True) for i in range(4)]
    assert live._dev is dev0
    ref = fresh(2000)
    for i in range(4):
        want = ref.search(Q[i], quota=300, limit=20, with_dists=True)
        assert got[i][1] == want[1] and [(r.id, r.code, r.dist) for r in got[i][0]] == [(r.id, r.code, r.dist) for r in want[0]]
    early = (codes[7], "00000003x")                              # sorts in the middle of its cell: the rebuild route
    live.add_codes([early[0]], [early[1]])
    got = [live.search(Q[i], quota=300, limit=20, with_dists=True) for i in range(4)]
    assert live._dev is not dev0
    ref2 = fresh(2000, [early])
    for i in range(4):
        want = ref2.search(Q[i], quota=300, limit=20, with_dists=True)
        assert got[i][1] == want[1] and [(r.id, r.code, r.dist) for r in got[i][0]] == [(r.id, r.code, r.dist) for r in want[0]]
    dev1 = live._dev
    live.add_codes([codes[11]], [ids[5]])                        # an existing key gets a new value: rebuild
    live.search(Q[0], quota=300, limit=20)
    assert live._dev is not dev1
    for s in (live, ref, ref2):
        s.close()


@gpu
def test_apply_pca_works_before_the_lopq_parameters_exist():
    """The reference's training flow calls apply_PCA between fit_pca and fit (lopq/lopq/model.py:878-937,
    searcher_lopqhbase.py:340): a model that holds only pca_P / pca_mu must project like the fitted one."""
    from columbiaimagesearch_amd.lopq import LOPQModelPCA
    m, z, Q = _lmdb_fixture()
    only = LOPQModelPCA(V=m.V, M=m.M, renorm=bool(z["renorm"]), parameters=(None, None, None, None, z["pca_P"], z["pca_mu"]))
    np.testing.assert_array_equal(only.apply_PCA(Q[:50]), m.apply_PCA(Q[:50]))
    assert only.apply_PCA(Q[0]).shape == m.apply_PCA(Q[0]).shape
    with pytest.raises(ValueError):
        only.predict(Q[0])  # still no LOPQ parameters
