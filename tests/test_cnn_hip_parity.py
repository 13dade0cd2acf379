"""GPU parity of the DeepSentibank forward against the CPU restatement (oracle/cnn_oracle.py), seeded synthetic
weights.  float32 everywhere; the MFMA accumulates k-ascending in float32 like caffe's sgemm would, only the
order differs => tolerance 2e-4 relative to the feature scale (parity with caffe itself is unpinned)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net_and_weights():
    from oracle import cnn_oracle as C
    from columbiaimagesearch_amd.featurizer import SentiBankNet
    w = C.synthetic_weights(0)
    return SentiBankNet(w), w


def test_forward_matches_torch_cpu(net_and_weights):
    from oracle import cnn_oracle as C
    net, w = net_and_weights
    x = C.synthetic_images(5, seed=3)  # 5: exercises partial pixel tiles
    got = net.forward(x)
    ref = C.forward_torch(x, w)
    assert got.shape == (5, 4096) and got.dtype == np.float32
    scale = np.abs(ref).max()
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-4 * scale)
    assert ((got > 0) == (ref > 0)).mean() > 0.999  # post-ReLU read-out (reference :154)


def test_batch_equals_single_and_device_entry(net_and_weights):
    import torch
    from oracle import cnn_oracle as C
    net, w = net_and_weights
    x = C.synthetic_images(3, seed=4)
    a = net.forward(x)
    b = np.stack([net.forward(x[i:i + 1])[0] for i in range(3)])
    np.testing.assert_array_equal(a, b)  # batching must not change a single bit
    d = net.forward_dev(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(d.cpu().numpy(), a)


@pytest.mark.parametrize("arch", ["sentibank", "dlib"])
def test_forward_does_not_read_what_it_has_not_written(monkeypatch, arch):
    """Every workspace of a forward filled with NaN bytes beforehand (CIS_CNN_POISON, one workspace at a time and all together):
    the descriptors do not change.  Round 4: the pad floats of DeepSentibank's NHWC input rows were read (against zero weights)
    without ever being written; in memory that an earlier owner had filled with 0xff the products were NaN, the ReLU made them 0
    and the descriptors were off by 0.5 % -- only in a process that had run LOPQ searches before, only on the first small batch."""
    import torch
    if arch == "sentibank":
        from oracle import cnn_oracle as C
        from columbiaimagesearch_amd.featurizer import SentiBankNet
        net, mk = SentiBankNet(C.synthetic_weights(0)), lambda n: C.synthetic_images(n, seed=5)
    else:
        from oracle import dlib_oracle as D
        from columbiaimagesearch_amd.featurizer import DLibFaceNet
        net, mk = DLibFaceNet(D.synthetic_weights(1)), lambda n: D.synthetic_chips(n, seed=5)
    for n in (1, 5, 130):
        x = torch.from_numpy(np.ascontiguousarray(mk(n), dtype=np.float32)).cuda()
        monkeypatch.delenv("CIS_CNN_POISON", raising=False)
        net.forward_dev(x)
        want = net.forward_dev(x).clone()
        for mask in (1, 2, 4, 8, 16, 31):
            monkeypatch.setenv("CIS_CNN_POISON", str(mask))
            net.forward_dev(x)  # (a workspace that did not exist before this call is filled from the next call on)
            got = net.forward_dev(x)
            assert torch.equal(got, want), (arch, n, mask)
        monkeypatch.delenv("CIS_CNN_POISON")
    net.close()


def test_featurizer_surface(tmp_path, net_and_weights):
    """get_featurizer('sbpycaffe', conf, prefix) -> featurize(buffer) -> (4096,) float32, b64 round trip."""
    import io
    from PIL import Image
    from oracle import cnn_oracle as C
    from columbiaimagesearch_amd.featurizer import featB64decode, get_feat_size, get_featurizer, normfeatB64encode
    _, w = net_and_weights
    np.savez(tmp_path / "w.npz", **w)
    mean = np.zeros((3, 256, 256)) + np.array([104.0, 116.7, 122.7])[:, None, None]
    np.save(tmp_path / "mean.npy", mean)
    conf = {"SBF_sbcaffe_path": str(tmp_path / "w.npz"), "SBF_imgmean_path": str(tmp_path / "mean.npy")}
    f = get_featurizer("sbpycaffe", conf, prefix="SBF_")
    rs = np.random.RandomState(0)
    bufs = []
    for i in range(2):
        b = io.BytesIO()
        Image.fromarray(rs.randint(0, 255, (300 + 40 * i, 280, 3), dtype=np.uint8)).save(b, format="PNG")
        bufs.append(b.getvalue())
    feat = f.featurize(bufs[0])
    assert feat.shape == (get_feat_size("sbpycaffe"),) and feat.dtype == np.float32
    ref = C.forward_torch(f.preprocess_img(bufs[0])[None], w)[0]
    np.testing.assert_allclose(feat, ref, rtol=0, atol=2e-4 * np.abs(ref).max())
    both = f.featurize_batch(bufs)
    np.testing.assert_array_equal(both[0], feat)
    back = featB64decode(normfeatB64encode(feat), "sbpycaffe")
    np.testing.assert_allclose(np.linalg.norm(back), 1.0, rtol=1e-5)


def test_batched_extractor_rows_match_single_image_rows(tmp_path, net_and_weights):
    """GenericExtractor('full', 'sbpycaffe', ...).process_batch == [process_buffer(b) ...], bad images -> failed row."""
    import io
    from PIL import Image
    from columbiaimagesearch_amd.extractor import GenericExtractor
    from columbiaimagesearch_amd.featurizer import featB64decode
    _, w = net_and_weights
    np.savez(tmp_path / "w.npz", **w)
    np.save(tmp_path / "mean.npy", np.zeros((3, 256, 256)) + 110.0)
    conf = {"EX_sbcaffe_path": str(tmp_path / "w.npz"), "EX_imgmean_path": str(tmp_path / "mean.npy")}
    ex = GenericExtractor("full", "sbpycaffe", "image", "ext", "EX_", conf)
    assert ex.extr_str == "ext:sbpycaffe_feat_full_image" and ex.extr_str_processed.endswith("_processed")
    rs = np.random.RandomState(1)
    bufs = []
    for i in range(3):
        b = io.BytesIO()
        Image.fromarray(rs.randint(0, 255, (240 + 30 * i, 320, 3), dtype=np.uint8)).save(b, format="JPEG")
        bufs.append(b.getvalue())
    rows = ex.process_batch(bufs[:2] + [b"not an image"] + bufs[2:])
    assert rows[2] == {"ext:sbpycaffe_feat_full_image_failed": "1"}
    for r, b in zip([rows[0], rows[1], rows[3]], bufs):
        one = ex.process_buffer(b)
        assert r == one and r[ex.extr_str_processed] == "1"
        f = featB64decode(r[ex.extr_str], "sbpycaffe")
        assert f.shape == (4096,) and abs(np.linalg.norm(f) - 1.0) < 1e-5


def test_preprocess_pool_rows_equal_process_buffer_rows(tmp_path, net_and_weights):
    """Worker processes decode / resize into the shared ring, one forward for the batch: rows identical (same strings)
    to the per-image entry point; an undecodable buffer gets the failure row."""
    import io
    from PIL import Image
    from columbiaimagesearch_amd.extractor import GenericExtractor
    from columbiaimagesearch_amd.extractor.preprocess_pool import PreprocessPool
    _, w = net_and_weights
    np.savez(tmp_path / "w.npz", **w)
    np.save(tmp_path / "mean.npy", np.zeros((3, 256, 256)) + 110.0)
    conf = {"EX_sbcaffe_path": str(tmp_path / "w.npz"), "EX_imgmean_path": str(tmp_path / "mean.npy")}
    ex = GenericExtractor("full", "sbpycaffe", "image", "ext", "EX_", conf)
    rs = np.random.RandomState(4)
    bufs = []
    for i in range(7):
        b = io.BytesIO()
        Image.fromarray(rs.randint(0, 255, (200 + 17 * i, 300 - 11 * i, 3), dtype=np.uint8)).save(b, format="JPEG" if i % 2 else "PNG")
        bufs.append(b.getvalue())
    bufs.insert(3, b"garbage")
    pool = PreprocessPool(ex.featurizer, workers=3, slots=5)  # fewer slots than images: two rounds
    try:
        rows = ex.process_batch(bufs, pool=pool)
    finally:
        pool.close()
    assert rows[3] == ex.failed_out_dict()
    for k, b in enumerate(bufs):
        if k != 3:
            assert rows[k] == ex.process_buffer(b), k


def test_dlib_resnet_matches_torch_cpu(tmp_path):
    """dlib face ResNet (29 convolutions, residual adds with zero padding) vs the CPU restatement, synthetic weights."""
    from oracle import dlib_oracle as D
    from columbiaimagesearch_amd.featurizer import DLibFaceNet, get_feat_size, get_featurizer
    w = D.synthetic_weights(0)
    net = DLibFaceNet(w)
    chips = D.synthetic_chips(5, seed=2)
    got = net.forward(chips)
    ref = D.forward_torch(chips, w)
    assert got.shape == (5, 128) and got.dtype == np.float32
    np.testing.assert_allclose(got, ref, rtol=0, atol=3e-4 * np.abs(ref).max())
    one = np.stack([net.forward(chips[i:i + 1])[0] for i in range(5)])
    np.testing.assert_array_equal(got, one)
    np.savez(tmp_path / "rec.npz", **w)
    f = get_featurizer("dlib", {"D_rec_path": str(tmp_path / "rec.npz")}, prefix="D_")
    d = f.featurize_chips(chips[:2])
    assert d.dtype == np.float64 and d.shape == (2, get_feat_size("dlib"))
    np.testing.assert_array_equal(d, got[:2].astype(np.float64))


def test_dlib_direct_3x3_kernel_agrees_with_implicit_gemm(monkeypatch):
    """The direct 3x3 kernel (k_conv3x3_direct: the 35x35x32 and 17x17x64 residual branches) against the implicit-GEMM route
    of the same layers (CIS_CNN_NO_DIRECT), on batches whose tiles straddle image borders, and for every tile shape."""
    from oracle import dlib_oracle as D
    from columbiaimagesearch_amd.featurizer import DLibFaceNet
    net = DLibFaceNet(D.synthetic_weights(3))
    for n, seed in ((1, 7), (7, 8), (70, 9)):
        chips = D.synthetic_chips(n, seed=seed)
        monkeypatch.setenv("CIS_CNN_NO_DIRECT", "1")
        ref = net.forward(chips)
        monkeypatch.delenv("CIS_CNN_NO_DIRECT")
        tol = 3e-5 * np.abs(ref).max()  # same products, different float32 summation order
        for cfg in ("0", "1", "2"):
            monkeypatch.setenv("CIS_CNN_DIRECT_CFG", cfg)
            got = net.forward(chips)
            assert np.isfinite(got).all()
            np.testing.assert_allclose(got, ref, rtol=0, atol=tol)
            # a face's descriptor does not depend on its place in the batch: same accumulation order at every position
            np.testing.assert_array_equal(net.forward(chips[n - 1:n])[0], got[n - 1]) if n <= 7 else None
        monkeypatch.delenv("CIS_CNN_DIRECT_CFG")


def test_dlib_direct_first_layer_agrees_with_implicit_gemm(monkeypatch):
    """k_conv7x7s2_direct (first layer: raw chip in, normalisation + bias + ReLU fused, weights resident in registers) against the
    implicit-GEMM route of the same layer behind the separate normalisation pass (CIS_CNN_NO_DIRECT7), and against the CPU
    restatement; tiles that wrap output rows, the two-tile group at the end of an image, several batch sizes."""
    from oracle import dlib_oracle as D
    from columbiaimagesearch_amd.featurizer import DLibFaceNet
    w = D.synthetic_weights(5)
    net = DLibFaceNet(w)
    for n, seed in ((1, 1), (3, 2), (33, 3)):
        chips = D.synthetic_chips(n, seed=seed)
        monkeypatch.setenv("CIS_CNN_NO_DIRECT7", "1")
        ref = net.forward(chips)
        monkeypatch.delenv("CIS_CNN_NO_DIRECT7")
        got = net.forward(chips)
        assert np.isfinite(got).all()
        np.testing.assert_allclose(got, ref, rtol=0, atol=3e-5 * np.abs(ref).max())  # same products, another float32 summation order
        if n <= 3:
            cpu = D.forward_torch(chips, w)
            np.testing.assert_allclose(got, cpu, rtol=0, atol=3e-4 * np.abs(cpu).max())
    net.close()


def test_dlib_first_layer_with_the_max_pool_inside_equals_the_two_kernels(monkeypatch):
    """k_conv7x7s2_pool (round 5: the 7 x 7 / 2 convolution and max_pool<3,3,2,2> in one kernel, pooled from LDS) against
    k_conv7x7s2_direct + k_maxpool_nhwc_v4 (CIS_CNN_NO_POOL7): the same products in the same order and an exact max -- the descriptors
    are bit-identical; batch sizes that end inside a workgroup's image range, a chip's descriptor independent of its place."""
    from oracle import dlib_oracle as D
    from columbiaimagesearch_amd.featurizer import DLibFaceNet
    w = D.synthetic_weights(13)
    net = DLibFaceNet(w)
    for n, seed in ((1, 1), (3, 2), (34, 3)):
        chips = D.synthetic_chips(n, seed=seed)
        monkeypatch.setenv("CIS_CNN_NO_POOL7", "1")
        ref = net.forward(chips)
        monkeypatch.delenv("CIS_CNN_NO_POOL7")
        got = net.forward(chips)
        assert np.isfinite(got).all()
        np.testing.assert_array_equal(got, ref)
        np.testing.assert_array_equal(net.forward(chips[n - 1:n])[0], got[n - 1])
    net.close()


def test_dlib_fused_tail_agrees_with_the_six_launches_it_replaces(monkeypatch):
    """k_dlib_tail (round 5: the last block's second convolution -- its centre tap, all a 1 x 1 map ever meets --, pooled skip branch,
    add_prev, ReLU, global average and fc_no_bias in one launch) against the separate launches (CIS_CNN_NO_TAIL) and the CPU
    restatement; odd batch sizes (two chips per workgroup), a chip's descriptor independent of its place in the batch."""
    from oracle import dlib_oracle as D
    from columbiaimagesearch_amd.featurizer import DLibFaceNet
    w = D.synthetic_weights(11)
    net = DLibFaceNet(w)
    for n, seed in ((1, 4), (2, 5), (5, 6), (33, 7)):
        chips = D.synthetic_chips(n, seed=seed)
        monkeypatch.setenv("CIS_CNN_NO_TAIL", "1")
        ref = net.forward(chips)
        monkeypatch.delenv("CIS_CNN_NO_TAIL")
        got = net.forward(chips)
        assert np.isfinite(got).all()
        np.testing.assert_allclose(got, ref, rtol=0, atol=3e-5 * np.abs(ref).max())  # same products, another float32 summation order
        np.testing.assert_array_equal(net.forward(chips[n - 1:n])[0], got[n - 1])
        if n <= 5:
            cpu = D.forward_torch(chips, w)
            np.testing.assert_allclose(got, cpu, rtol=0, atol=3e-4 * np.abs(cpu).max())
    net.close()


def test_lane_streams_are_made_once_and_reused():
    """columbiaimagesearch_amd.streams.lane_streams: the same stream objects on every call (the pool grows, it is never remade), distinct
    streams, none of them the default stream; a forward on each of them gives the descriptors of the default stream."""
    import torch
    from oracle import dlib_oracle as D
    from columbiaimagesearch_amd.featurizer import DLibFaceNet
    from columbiaimagesearch_amd.streams import lane_streams
    a = lane_streams(3)
    b = lane_streams(5)
    assert [x.cuda_stream for x in a] == [x.cuda_stream for x in b[:3]]
    assert len({x.cuda_stream for x in b}) == 5 and torch.cuda.default_stream().cuda_stream not in {x.cuda_stream for x in b}
    net = DLibFaceNet(D.synthetic_weights(2))
    x = torch.as_tensor(D.synthetic_chips(160, seed=3)).cuda().float().contiguous()
    want = net.forward_dev(x).cpu().numpy()
    torch.cuda.synchronize()
    for s in lane_streams(4):
        with torch.cuda.stream(s):
            got = net.forward_dev(x)
        s.synchronize()
        np.testing.assert_array_equal(got.cpu().numpy(), want)
    net.close()


def test_dlib_forward_as_a_captured_graph_equals_the_launches(monkeypatch):
    """CIS_CNN_GRAPH=1: the forward of (input, n, output) is captured on its second call and replayed as one hipGraphLaunch from the
    third on -- the descriptors of every call equal the launches', bit for bit; new contents at the same addresses are seen (a graph
    bakes in addresses, not data); a switch flipped between calls (CIS_CNN_NO_TAIL) takes the launches of THAT route, not the cached
    graph; a view captures its own single chain."""
    import torch
    from oracle import dlib_oracle as D
    from columbiaimagesearch_amd.featurizer import DLibFaceNet
    net = DLibFaceNet(D.synthetic_weights(2))
    x = torch.as_tensor(D.synthetic_chips(160, seed=3)).cuda().float().contiguous()
    out = torch.empty((160, 128), device="cuda")
    monkeypatch.delenv("CIS_CNN_GRAPH", raising=False)
    want = net.forward_dev(x).cpu().numpy()
    monkeypatch.setenv("CIS_CNN_GRAPH", "1")
    for _ in range(4):  # launches, capture + launch, graph, graph
        out.fill_(-1.0)
        net.forward_dev(x, out)
        np.testing.assert_array_equal(out.cpu().numpy(), want)
    x2 = torch.as_tensor(D.synthetic_chips(160, seed=4)).cuda().float().contiguous()
    monkeypatch.delenv("CIS_CNN_GRAPH")
    want2 = net.forward_dev(x2).cpu().numpy()
    monkeypatch.setenv("CIS_CNN_GRAPH", "1")
    x.copy_(x2)
    net.forward_dev(x, out)
    np.testing.assert_array_equal(out.cpu().numpy(), want2)
    monkeypatch.setenv("CIS_CNN_NO_TAIL", "1")
    net.forward_dev(x, out)
    no_tail = out.cpu().numpy().copy()
    monkeypatch.delenv("CIS_CNN_GRAPH")
    net.forward_dev(x, out)
    np.testing.assert_array_equal(out.cpu().numpy(), no_tail)
    monkeypatch.delenv("CIS_CNN_NO_TAIL")
    monkeypatch.setenv("CIS_CNN_GRAPH", "1")
    v = net.view()
    side = torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(side):
            out.fill_(-1.0)
            v.forward_dev(x, out)
        side.synchronize()
        got = out.cpu().numpy()
    monkeypatch.delenv("CIS_CNN_GRAPH")
    with torch.cuda.stream(side):
        v.forward_dev(x, out)
    side.synchronize()
    np.testing.assert_array_equal(got, out.cpu().numpy())
    v.close()
    net.close()


@pytest.mark.parametrize("n", [128, 257, 512])
def test_dlib_batch_in_concurrent_parts_equals_the_single_chain(monkeypatch, n):
    """A batch of 128-512 chips runs as two parts on the handle's own streams (own workspaces, event fences on the caller's
    stream).  Part p of the default forward must equal, bit for bit, the single-chain forward (CIS_CNN_PARTS=1) of the same
    chips as their own batch (same batch size -> same tile / split-K choices); through the host entry point (NULL stream,
    copy right behind the fences), through forward_dev on a side torch stream, and with 3 and 4 parts."""
    import torch
    from oracle import dlib_oracle as D
    from columbiaimagesearch_amd.featurizer import DLibFaceNet
    net = DLibFaceNet(D.synthetic_weights(1))
    chips = D.synthetic_chips(n, seed=n)

    def single_chain(lo, hi):
        monkeypatch.setenv("CIS_CNN_PARTS", "1")
        out = net.forward(chips[lo:hi])
        monkeypatch.delenv("CIS_CNN_PARTS")
        return out

    for parts in (2, 3, 4):
        want = np.concatenate([single_chain(n * p // parts, n * (p + 1) // parts) for p in range(parts)])
        if parts == 2:
            monkeypatch.delenv("CIS_CNN_PARTS", raising=False)  # the default route of this batch size
        else:
            monkeypatch.setenv("CIS_CNN_PARTS", str(parts))
        got_host = net.forward(chips)
        np.testing.assert_array_equal(got_host, want)
        side = torch.cuda.Stream()
        x = torch.as_tensor(chips).cuda()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            y = x * 1.0                       # earlier work on the caller's stream that the parts must wait for
            out = net.forward_dev(y)
            out2 = out + 0.0                  # later work on the caller's stream that must wait for every part
        side.synchronize()
        np.testing.assert_array_equal(out2.cpu().numpy(), want)
        monkeypatch.delenv("CIS_CNN_PARTS", raising=False)
    # against the single chain of the whole batch: same products, float32 summation order of the other tile choice
    whole = single_chain(0, n)
    np.testing.assert_allclose(got_host, whole, rtol=0, atol=3e-5 * np.abs(whole).max())
    net.close()


def test_sentibank_batch_in_concurrent_parts_equals_the_single_chain(monkeypatch, net_and_weights):
    net, w = net_and_weights
    from oracle import cnn_oracle as C
    x = C.synthetic_images(9, seed=11)
    for parts in (2, 3):
        monkeypatch.setenv("CIS_CNN_PARTS", "1")
        want = np.concatenate([net.forward(x[9 * p // parts: 9 * (p + 1) // parts]) for p in range(parts)])
        monkeypatch.setenv("CIS_CNN_PARTS", str(parts))
        np.testing.assert_array_equal(net.forward(x), want)
    monkeypatch.delenv("CIS_CNN_PARTS")


def test_batch_ingest_matches_per_item_chain():
    """CNN -> L2 normalise -> LOPQ encode -> insert on the GPU == the same chain item by item through the host surfaces."""
    import torch
    from conftest import load_golden
    from oracle import dlib_oracle as D
    from columbiaimagesearch_amd.featurizer import DLibFaceNet
    from columbiaimagesearch_amd.ingest import BatchIngest
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    from test_lopq_hip_parity import hip_model
    z, X, Q = load_golden("c2")  # 128-d PCA model: the dlib descriptor size
    model = hip_model(z)
    net = DLibFaceNet(D.synthetic_weights(0))
    chips = D.synthetic_chips(6, seed=5)
    s = LOPQSearcherHIP(model)
    ing = BatchIngest(net, model, s, feat_dtype=torch.float64)
    x = torch.as_tensor(chips, dtype=torch.float32).cuda().contiguous()
    assert ing.ingest_batch(x, ids=np.arange(100, 106)) == 6 and s.get_nb_indexed() == 6
    feats = net.forward(chips).astype(np.float64)
    feats /= np.linalg.norm(feats, axis=1, keepdims=True)
    coarse, fine = model.predict_batch(feats)
    for i in range(6):
        items = s.get_cell((int(coarse[i, 0]), int(coarse[i, 1])))
        assert any(it[0] == 100 + i and tuple(it[1][1]) == tuple(int(v) for v in fine[i]) for it in items)


def test_pipelined_ingest_equals_batch_by_batch():
    """BatchIngest.ingest_batches (round 5): CNN forwards of the next batches in flight on views of the net while the current batch is
    encoded and inserted -- same counts, same cells in the same order as ingest_batch batch by batch (duplicates across batches incl.)."""
    import torch
    from conftest import load_golden
    from oracle import dlib_oracle as D
    from columbiaimagesearch_amd.featurizer import DLibFaceNet
    from columbiaimagesearch_amd.ingest import BatchIngest
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    from test_lopq_hip_parity import hip_model
    z, X, Q = load_golden("c2")
    model = hip_model(z)
    net = DLibFaceNet(D.synthetic_weights(0))
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    xs = [(torch.rand((n, 150, 150, 3), generator=g, device="cuda") * 255).contiguous() for n in (7, 1, 33, 12, 5)]
    xs.append(xs[2])  # a whole batch again under new ids, and ...
    ids = [torch.arange(1000 * i, 1000 * i + x.shape[0], device="cuda") for i, x in enumerate(xs)]
    ids.append(ids[0])  # ... one under ids that are stored already: every item a duplicate
    xs.append(xs[0])
    a, b = LOPQSearcherHIP(model), LOPQSearcherHIP(model)
    ia, ib = BatchIngest(net, model, a, feat_dtype=torch.float64), BatchIngest(net, model, b, feat_dtype=torch.float64)
    want = [ia.ingest_batch(x, ids=i) for x, i in zip(xs, ids)]
    got = ib.ingest_batches(xs, ids=ids, lanes=3)
    assert got == want and want[-1] == 0 and a.get_nb_indexed() == b.get_nb_indexed() == sum(want)
    V = model.V
    for c in range(V * V):
        ca, cb = a.get_cell((c // V, c % V)), b.get_cell((c // V, c % V))
        assert [i for i, _ in ca] == [i for i, _ in cb] and [tuple(x[1]) for _, x in ca] == [tuple(x[1]) for _, x in cb], c
    assert ib.ingest_batches(xs[:2], ids=[ids[0] + 50000, ids[1] + 50000], lanes=1) == [7, 1]  # one lane: the serial chain
    ib.close()


def test_batch_ingest_into_searchers_without_a_tuple_returning_device_insert(tmp_path):
    """BatchIngest must serve every searcher the package ships -- LOPQSearcherLMDB (host key / value store, string
    ids, no device entry point), GridSearcher (add_codes_dev returns a count, not a tuple) and non-integer ids on the HIP
    searcher -- with the same cells and codes as LOPQSearcherHIP gets."""
    import torch
    import torch.distributed as dist
    from conftest import load_golden
    from oracle import dlib_oracle as D
    from columbiaimagesearch_amd.distributed import GridSearcher
    from columbiaimagesearch_amd.featurizer import DLibFaceNet
    from columbiaimagesearch_amd.ingest import BatchIngest
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP, LOPQSearcherLMDB
    from test_lopq_hip_parity import hip_model
    z, X, Q = load_golden("c2")
    model = hip_model(z)
    net = DLibFaceNet(D.synthetic_weights(0))
    x = torch.as_tensor(D.synthetic_chips(5, seed=9), dtype=torch.float32).cuda().contiguous()
    ref = LOPQSearcherHIP(model)
    assert BatchIngest(net, model, ref, feat_dtype=torch.float64).ingest_batch(x, ids=np.arange(5)) == 5
    coarse, fine = BatchIngest(net, model, ref, feat_dtype=torch.float64).encode_batch_dev(x)
    coarse, fine = coarse.cpu().numpy().view(np.uint16), fine.cpu().numpy()
    sha = ["sha1_%d" % i for i in range(5)]
    # LMDB-order searcher: string ids, host store
    lm = LOPQSearcherLMDB(model, str(tmp_path / "ix"), id_lambda=str)
    assert BatchIngest(net, model, lm, feat_dtype=torch.float64).ingest_batch(x, ids=sha) == 5 and lm.get_nb_indexed() == 5
    # HIP searcher with non-integer ids (slots in the Python mirror)
    hs = LOPQSearcherHIP(model)
    assert BatchIngest(net, model, hs, feat_dtype=torch.float64).ingest_batch(x, ids=sha) == 5
    for i in range(5):
        cell = (int(coarse[i, 0]), int(coarse[i, 1]))
        want = tuple(int(v) for v in fine[i])
        assert any(it[0] == sha[i] and tuple(it[1][1]) == want for it in lm.get_cell(cell))
        assert any(it[0] == sha[i] and tuple(it[1][1]) == want for it in hs.get_cell(cell))
    lm.close()
    # GridSearcher over a world-1 group (1 query group x 1 cell shard): add_codes_dev returns an int
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        gs = GridSearcher(model, 1)
        ing = BatchIngest(net, model, gs, feat_dtype=torch.float64)
        assert ing.ingest_batch(x) == 5 and gs.get_nb_indexed() == 5
        assert ing.ingest_batch(x, ids=np.arange(5)) == 0  # the same ids in the same cells: duplicates (search.py:356-364)
    finally:
        dist.destroy_process_group()


def test_c5_sentibank_ingest_against_oracle(net_and_weights):
    """BASELINE config C5 at its true shapes: DeepSentibank batch -> L2 normalise (float32, featsio.py:13-22) ->
    LOPQModelPCA 4096 -> 256, V=16, M=16 (the reference-fitted c3full model) -> insert.  Bars: the CNN features
    against the CPU restatement (float tolerance), the device normalisation against numpy's within 2 float32 ulp, and
    -- the parity claim -- the codes of the GPU chain BIT-EXACT against oracle.compute_codes on the same normalised
    HIP features; then the index answers queries like the oracle index built from those codes."""
    import torch
    from conftest import load_golden
    from oracle import cnn_oracle as C
    from oracle import lopq_oracle as O
    from columbiaimagesearch_amd.ingest import BatchIngest, l2_normalize_dev
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    from test_lopq_hip_parity import hip_model
    net, w = net_and_weights
    z, X, Q = load_golden("c3full")
    model = hip_model(z)
    om = O.OracleModel.from_npz(z)
    n = 48
    imgs = C.synthetic_images(n, seed=9)
    x = torch.from_numpy(imgs).cuda()
    feats = net.forward_dev(x)
    ref = C.forward_torch(imgs[:4], w)
    np.testing.assert_allclose(feats[:4].cpu().numpy(), ref, rtol=0, atol=2e-4 * np.abs(ref).max())
    normed = l2_normalize_dev(feats).contiguous()
    f_host = feats.cpu().numpy()
    want_normed = np.stack([f / np.linalg.norm(f) for f in f_host])  # featsio.normfeatB64encode's arithmetic
    got_normed = normed.cpu().numpy()
    assert got_normed.dtype == np.float32
    np.testing.assert_allclose(got_normed, want_normed, rtol=2.4e-7, atol=0)
    s = LOPQSearcherHIP(model)
    ing = BatchIngest(net, model, s)
    coarse_d, fine_d = ing.encode_batch_dev(x)
    coarse = coarse_d.cpu().numpy().view(np.uint16)
    fine = fine_d.cpu().numpy()
    oc, of = O.compute_codes(om, got_normed)
    np.testing.assert_array_equal(coarse, oc)
    np.testing.assert_array_equal(fine, of)
    ids = np.arange(5000, 5000 + n)
    assert ing.ingest_batch(x, ids=ids) == n and s.get_nb_indexed() == n
    oi = O.OracleCSRIndex(om, oc, of, ids=ids)
    r = s.search_batch(got_normed[:8], quota=30, limit=20)
    for qi in range(8):
        wid, wd, wv = oi.search(got_normed[qi], quota=30, limit=20)
        k = int(r["n_found"][qi])
        assert k == len(wid) and int(r["visited"][qi]) == wv
        np.testing.assert_array_equal(r["ids"][qi, :k], wid)
        np.testing.assert_allclose(r["dists"][qi, :k], wd, rtol=1e-9, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sentibank", "dlib"])
def test_views_share_the_weights_and_overlap_batches(kind):
    """cis_cnn_create_view (round 5): a view computes what its base computes (same weights), batches in flight on three handles and three
    streams give the descriptors of the serial forwards bit for bit, a view outlives nothing: after the base is destroyed through the C
    API it answers an error instead of touching freed weights."""
    import torch
    from columbiaimagesearch_amd import _lib
    from columbiaimagesearch_amd.featurizer import DLibFaceNet, SentiBankNet
    from columbiaimagesearch_amd.featurizer.synthetic import dlib_weights, sentibank_weights
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    if kind == "dlib":
        net = DLibFaceNet(dlib_weights(1)); xs = [(torch.rand((n, 150, 150, 3), generator=g, device="cuda") * 255).contiguous() for n in (130, 9, 64)]
    else:
        net = SentiBankNet(sentibank_weights(1)); xs = [(torch.randn((n, 3, 227, 227), generator=g, device="cuda") * 50).contiguous() for n in (40, 3, 17)]
    views = [net.view(), net.view()]
    # reference: every batch alone through a view (one chain), everything synchronised
    want = []
    for x in xs:
        want.append(views[0].forward_dev(x).clone())
        torch.cuda.synchronize()
    # base == view on the same batch
    for x, w in zip(xs, want):
        assert torch.equal(net.forward_dev(x), w)
    torch.cuda.synchronize()
    # three batches in flight, twice around, nothing synchronised in between
    handles, streams = [net] + views, [torch.cuda.Stream() for _ in range(3)]
    outs = [None] * 6
    for i in range(6):
        with torch.cuda.stream(streams[i % 3]):
            outs[i] = handles[i % 3].forward_dev(xs[i % 3])
    torch.cuda.synchronize()
    for i in range(6):
        assert torch.equal(outs[i], want[i % 3]), i
    # the host entry point works on a view too
    assert np.array_equal(views[1].forward(xs[1].cpu().numpy()), want[1].cpu().numpy())
    # a base destroyed under its views (C API): the views answer an error
    h = net._h
    _lib.lib().cis_cnn_destroy(h)
    net._h = None
    out = torch.empty_like(want[1])
    rc = _lib.lib().cis_cnn_forward_dev(views[0]._h, xs[1].data_ptr(), xs[1].shape[0], out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and "base was destroyed" in _lib.last_error()
    for v in views:
        v.close()


@pytest.mark.parametrize("arch", ["sentibank", "dlib"])
def test_feature_bits_are_pinned(arch):
    """The float32 summation ORDER of a forward is part of what an index is built with: a descriptor near a quantiser boundary encodes
    differently when the order changes (round 5 changed the fc layers' K split from 4 to 8: features of rounds 1-4 and of round 5 differ
    in their last bits, INTEGRATION.md "feature kernel versions").  The sha1 of the descriptors of a seeded network on a seeded batch is
    pinned in tests/golden/cnn_feature_pins.json (made on an MI355X by this test's own code: tests/golden/make_cnn_feature_pins.py),
    so that the next order change is a deliberate one: regenerate the pins AND bump FEATURE_KERNEL_VERSION."""
    import hashlib
    import json
    from columbiaimagesearch_amd import featurizer as F
    sys_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    pins = json.load(open(os.path.join(sys_path, "cnn_feature_pins.json")))
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_cnn_feature_pins", os.path.join(sys_path, "make_cnn_feature_pins.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    got = mk.feature_sha1(arch)
    assert pins["feature_kernel_version"] == F.FEATURE_KERNEL_VERSION
    assert got == pins[arch], "the %s descriptors changed bits (%s, pinned %s): regenerate the pins and bump FEATURE_KERNEL_VERSION" % (arch, got, pins[arch])
