#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REAL reference.

Runs only in the build container (needs /root/reference).  The reference's vendored ``lopq``
package is python-2 source; it is converted *in a temporary directory outside the repo*
(lib2to3 + the integer-division sites listed in SURVEY.md section 8c), imported from there, and
used to produce input/output vectors.  Only data (``*.npz``) is written into the repo -- never
reference source.

    python tests/golden/make_golden.py            # regenerate every fixture
    python tests/golden/make_golden.py tiny c2    # regenerate some

Inputs are regenerated from seeds by ``tests/golden_inputs.py`` so the fixtures stay small; each
fixture stores a checksum of the inputs it was made from.
"""
import hashlib
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import golden_inputs as gi  # noqa: E402

REF = "/root/reference/lopq/lopq"


def import_reference():
    tmp = tempfile.mkdtemp(prefix="lopq_ref_")
    dst = os.path.join(tmp, "lopq")
    shutil.copytree(REF, dst)
    files = [os.path.join(dst, f) for f in ("model.py", "search.py", "utils.py", "eval.py", "__init__.py")]
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n"] + files,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    fixes = [(r"D / num_buckets", "D // num_buckets"), (r"M / 2", "M // 2"),
             (r"len\(arr\) / 2", "len(arr) // 2"), (r"len\(x\) / splits", "len(x) // splits"),
             (r"size / \(4", "size // (4"), (r"N / num_procs", "N // num_procs")]
    for f in files[:3]:
        src = open(f).read()
        for a, b in fixes:
            src = re.sub(a, b, src)
        open(f, "w").write(src)
    sys.path.insert(0, tmp)
    import lopq  # noqa
    return tmp


def model_arrays(m):
    nf = m.num_fine_splits
    out = {
        "Cs": np.stack([m.Cs[0], m.Cs[1]]),
        "Rs": np.stack([m.Rs[0], m.Rs[1]]),
        "mus": np.stack([m.mus[0], m.mus[1]]),
        "subs": np.stack([np.stack([m.subquantizers[s][j] for j in range(nf)]) for s in range(2)]),
        "num_fine_splits": np.int64(nf),
        "has_pca": np.bool_(hasattr(m, "pca_P") and m.pca_P is not None),
        "renorm": np.bool_(getattr(m, "renorm", False)),
    }
    if out["has_pca"]:
        out["pca_P"] = np.asarray(m.pca_P)
        out["pca_mu"] = np.asarray(m.pca_mu)
    return out


def sha1(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def codes_to_arrays(codes):
    coarse = np.array([[int(c) for c in code[0]] for code in codes], dtype=np.uint16)
    fine = np.array([[int(f) for f in code[1]] for code in codes], dtype=np.uint8)
    return coarse, fine


def run_searches(searcher, Q, settings, multiseq_n=64):
    """For each (quota, limit) run the reference search; store ragged results padded with -1."""
    from lopq.search import multisequence
    out = {}
    model = searcher.model
    is_pca = hasattr(model, "pca_P") and model.pca_P is not None
    for quota, limit in settings:
        L = quota if limit is None else limit
        ids = -np.ones((len(Q), L), dtype=np.int64)
        dists = np.full((len(Q), L), np.nan)
        nres = np.zeros(len(Q), dtype=np.int64)
        visited = np.zeros(len(Q), dtype=np.int64)
        nretr = np.zeros(len(Q), dtype=np.int64)
        for qi, q in enumerate(Q):
            res, vis = searcher.search(q, quota=quota, limit=limit, with_dists=True)
            res = list(res)
            nres[qi] = len(res)
            visited[qi] = vis
            for r, item in enumerate(res):
                ids[qi, r] = item.id
                dists[qi, r] = item.dist
            xq = model.apply_PCA(q) if is_pca else q
            retrieved, _ = searcher.get_result_quota(xq, quota)
            nretr[qi] = len(retrieved)
        tag = "q%d_l%s" % (quota, "N" if limit is None else str(limit))
        out["s_%s_ids" % tag] = ids
        out["s_%s_dists" % tag] = dists
        out["s_%s_n" % tag] = nres
        out["s_%s_visited" % tag] = visited
        out["s_%s_retrieved" % tag] = nretr
    # multisequence order
    ms = -np.ones((len(Q), multiseq_n, 2), dtype=np.int64)
    msd = np.full((len(Q), multiseq_n), np.nan)
    for qi, q in enumerate(Q):
        xq = model.apply_PCA(q) if is_pca else q
        for k, (d, cell) in enumerate(multisequence(xq, model.Cs)):
            if k >= multiseq_n:
                break
            ms[qi, k] = cell
            msd[qi, k] = d
    out["multiseq_cells"] = ms
    out["multiseq_dists"] = msd
    return out


def aux_outputs(model, X, n=32):
    """project / tables / reconstruct for the first n vectors."""
    is_pca = hasattr(model, "pca_P") and model.pca_P is not None
    Xp = model.apply_PCA(X[:n]) if is_pca else X[:n]
    proj, tabs, recon = [], [], []
    for x in Xp:
        c = model.predict_coarse(x)
        proj.append(model.project(x, c))
        tabs.append(np.stack(model.get_subquantizer_distances(x, c)))
        code = (c, model.predict_fine(x, c))
        recon.append(model.reconstruct(code))
    out = {"aux_project": np.stack(proj), "aux_tables": np.stack(tabs), "aux_reconstruct": np.stack(recon)}
    if is_pca:
        out["aux_pca"] = model.apply_PCA(X[:256])
        out["aux_pca_single"] = np.stack([model.apply_PCA(x) for x in X[:8]])
    return out


def near_tie_vectors(model, n, seed, dtype):
    """Vectors engineered to sit next to a coarse decision boundary (midpoint of the two nearest
    centroids of half 0 and of half 1, plus noise of a few float32 ulps)."""
    rs = np.random.RandomState(seed)
    C0, C1 = model.Cs
    h = C0.shape[1]
    out = np.zeros((n, 2 * h), dtype=np.float64)
    for i in range(n):
        for s, C in enumerate((C0, C1)):
            a, b = rs.choice(C.shape[0], 2, replace=False)
            mid = 0.5 * (C[a].astype(np.float64) + C[b].astype(np.float64))
            out[i, s * h:(s + 1) * h] = mid + rs.randn(h) * 1e-7 * (i % 4)
    return out.astype(dtype)


def make_c1(outdir):
    """BASELINE config C1: LOPQModel V=8, M=4 on 100k x 128 float32, 100 queries."""
    from lopq import LOPQModel, LOPQSearcher
    from lopq.utils import compute_codes_notparallel
    X, Q = gi.c1_inputs()
    m = LOPQModel(V=8, M=4)
    t = time.time()
    m.fit(X, n_init=1, random_state=1234)
    print("c1 fit %.1fs" % (time.time() - t))
    t = time.time()
    codes = compute_codes_notparallel(X, m)
    enc_s = time.time() - t
    print("c1 encode %.1fs (%.0f vec/s)" % (enc_s, len(X) / enc_s))
    coarse, fine = codes_to_arrays(codes)
    s = LOPQSearcher(m)
    s.add_codes(codes)
    d = model_arrays(m)
    d.update(inputs_sha1=np.array(sha1(X) + sha1(Q)), coarse_sha1=np.array(sha1(coarse)), fine_sha1=np.array(sha1(fine)),
             coarse_head=coarse[:4096], fine_head=fine[:4096], nb_indexed=np.int64(s.get_nb_indexed()),
             ref_encode_vec_per_s=np.float64(len(X) / enc_s))
    T = near_tie_vectors(m, 512, 77, np.float32)
    tc, tf = codes_to_arrays(compute_codes_notparallel(T, m))
    d.update(tie_X=T, tie_coarse=tc, tie_fine=tf)
    t = time.time()
    d.update(run_searches(s, Q, [(10, 10), (1000, 10), (10000, 10), (1000, None)]))
    srch_s = time.time() - t
    print("c1 searches %.1fs" % srch_s)
    d.update(aux_outputs(m, X))
    np.savez_compressed(os.path.join(outdir, "c1.npz"), **d)


def make_pca_fixture(outdir, name, V, M, pca_dims, n_index, n_train, nq, inputs, settings, pca_subsample=None, train_X=None):
    from lopq import LOPQModelPCA, LOPQSearcher
    from lopq.utils import compute_codes_notparallel
    X, Q = inputs
    m = LOPQModelPCA(V=V, M=M, renorm=True)
    t = time.time()
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        m.fit(X[:n_train] if train_X is None else train_X, pca_dims=pca_dims, n_init=1, random_state=4321, pca_subsample=pca_subsample)
    print("%s fit %.1fs  Cs %s Rs %s" % (name, time.time() - t, m.Cs[0].dtype, m.Rs[0].dtype))
    Xi = X[:n_index]
    with contextlib.redirect_stdout(io.StringIO()):
        codes = compute_codes_notparallel(Xi, m)
    coarse, fine = codes_to_arrays(codes)
    s = LOPQSearcher(m)
    s.add_codes(codes)
    d = model_arrays(m)
    d.update(inputs_sha1=np.array(sha1(X) + sha1(Q)), coarse=coarse, fine=fine,
             n_index=np.int64(n_index), nb_indexed=np.int64(s.get_nb_indexed()))
    Tp = near_tie_vectors(m, 128, 78, np.float32)  # near ties live in PCA space ...
    # ... map them back through the (orthonormal-column) PCA so that apply_PCA lands next to them
    T = (np.dot(Tp.astype(np.float64), m.pca_P.T) + m.pca_mu).astype(X.dtype)
    with contextlib.redirect_stdout(io.StringIO()):
        tc, tf = codes_to_arrays(compute_codes_notparallel(T, m))
    d.update(tie_X=T, tie_coarse=tc, tie_fine=tf)
    with contextlib.redirect_stdout(io.StringIO()):
        d.update(run_searches(s, Q[:nq], settings))
    d.update(aux_outputs(m, X))
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **d)


def make_c2(outdir):
    """C2-shaped: 128-d float64 unit vectors (dlib-like), LOPQModelPCA V=16, M=8, renorm."""
    make_pca_fixture(outdir, "c2", 16, 8, 128, 50000, 30000, 64, gi.c2_inputs(),
                     [(10, 10), (1000, 100), (10000, 100)])


def make_c4(outdir):
    """Bench model (config C4): 128-d descriptor-like unit vectors, LOPQModelPCA V=16, M=8, renorm."""
    make_pca_fixture(outdir, "c4", 16, 8, 128, 50000, 100000, 64, gi.c4_inputs(),
                     [(10, 10), (1000, 100), (10000, 100)])


def make_c3(outdir):
    """C3-shaped, scaled down: float32 non-negative features, PCA 320 -> 128, V=16, M=16."""
    make_pca_fixture(outdir, "c3", 16, 16, 128, 20000, 20000, 32, gi.c3_inputs(),
                     [(10, 10), (1000, 100), (10000, 100)])


def make_c3b(outdir):
    """C3 true sub-vector shape (h=128, w=16): PCA 288 -> 256, V=2, M=16."""
    make_pca_fixture(outdir, "c3b", 2, 16, 256, 8000, 8000, 16, gi.c3b_inputs(),
                     [(10, 10), (1000, 100)])


def make_c3full(outdir):
    """C3 at its TRUE shape: float32 non-negative 4096-d features, PCA 4096 -> 256 (pca_dims of
    conf/conf_search_sbpycaffe_release.json:12), V=16, M=16 (h=128, w=16).  The reference fits everything
    (PCA on 2500 of the training vectors: its per-sample np.outer loop costs 0.1 s per 4096-d sample; the LOPQ stages on all
    20000 training vectors of golden_inputs.c3full_train -- every local rotation has more points than dimensions)."""
    make_pca_fixture(outdir, "c3full", 16, 16, 256, 4000, 6000, 32, gi.c3full_inputs(),
                     [(10, 10), (1000, 100), (10000, 100)], pca_subsample=2500, train_X=gi.c3full_train())


def make_pk(outdir):
    """What production stores (SURVEY.md section 8b gotcha iii / 8f row 2): the model OBJECT pickled by the storer
    (cufacesearch/storer/local.py:58) and a per-update codes dict {id: [coarse, fine]} built exactly as
    searcher_lopqhbase.py:503-512 does, both pickled from the real reference classes (module ``lopq.model``), plus
    the expected codes / search results as arrays.  Protocol 2 = the highest a python-2 writer could have used."""
    import pickle
    from lopq import LOPQModel, LOPQModelPCA, LOPQSearcher
    from lopq.utils import compute_codes_notparallel
    import contextlib, io
    X, Q = gi.pk_inputs()
    sub = os.path.join(outdir, "pk")
    os.makedirs(sub, exist_ok=True)
    d = {"inputs_sha1": np.array(sha1(X) + sha1(Q))}
    for tag, m, fitkw in (("lopq", LOPQModel(V=4, M=4, subquantizer_clusters=16), {}),
                          ("lopq_pca", LOPQModelPCA(V=4, M=4, subquantizer_clusters=16, renorm=True), {"pca_dims": 16})):
        Xm = X[:, :16] if tag == "lopq" else X
        Qm = Q[:, :16] if tag == "lopq" else Q
        with contextlib.redirect_stdout(io.StringIO()):
            m.fit(Xm[:2000], n_init=1, random_state=11, **fitkw)
            codes = compute_codes_notparallel(Xm, m)
        det_ids = ["%040x_%d" % (i * 2654435761 % (1 << 61), i % 3) for i in range(len(Xm))]  # sha1-like strings
        codes_dict = dict()
        for i, code in enumerate(codes):  # searcher_lopqhbase.py:506-512
            codes_dict[det_ids[i]] = [code.coarse, code.fine]
        with open(os.path.join(sub, "model_%s.pkl" % tag), "wb") as f:
            pickle.dump(m, f, protocol=2)
        if tag == "lopq":  # the reference's .mat exchange format (lopq/lopq/model.py:712-728), written by the reference
            m.export_mat(os.path.join(sub, "model_lopq.mat"))
        with open(os.path.join(sub, "codes_%s.pkl" % tag), "wb") as f:
            pickle.dump(codes_dict, f, protocol=2)
        s = LOPQSearcher(m)
        s.add_codes_from_dict(codes_dict)
        coarse, fine = codes_to_arrays(codes)
        d["%s_coarse" % tag], d["%s_fine" % tag] = coarse, fine
        d["%s_nb_indexed" % tag] = np.int64(s.get_nb_indexed())
        d["%s_attrs" % tag] = np.array(sorted(m.__dict__))
        # ranked ids as positions in det_ids (ids are strings)
        pos_of = {k: i for i, k in enumerate(det_ids)}
        for quota, limit in ((20, 10), (500, 50)):
            ids = -np.ones((len(Qm), limit), dtype=np.int64)
            dists = np.full((len(Qm), limit), np.nan)
            vis = np.zeros(len(Qm), dtype=np.int64)
            for qi, q in enumerate(Qm):
                with contextlib.redirect_stdout(io.StringIO()):
                    res, v = s.search(q, quota=quota, limit=limit, with_dists=True)
                vis[qi] = v
                for r, item in enumerate(res):
                    ids[qi, r] = pos_of[item.id]
                    dists[qi, r] = item.dist
            d["%s_q%d_l%d_ids" % (tag, quota, limit)] = ids
            d["%s_q%d_l%d_dists" % (tag, quota, limit)] = dists
            d["%s_q%d_l%d_visited" % (tag, quota, limit)] = vis
    np.savez_compressed(os.path.join(sub, "expected.npz"), **d)


def make_tiny(outdir):
    """Edge cases: float64 centroids, K=16, empty cells, duplicates, explicit ids, limit=None."""
    from lopq import LOPQModel, LOPQSearcher
    from lopq.utils import compute_codes_notparallel
    X, Q = gi.tiny_inputs()
    m = LOPQModel(V=4, M=4, subquantizer_clusters=16)
    m.fit(X[:3000], n_init=1, random_state=7)
    print("tiny Cs dtype", m.Cs[0].dtype)
    codes = compute_codes_notparallel(X, m)
    coarse, fine = codes_to_arrays(codes)
    # index only a slice so that several cells stay empty, with explicit ids and duplicates
    sel = np.nonzero((coarse[:, 0] != 1) & (coarse[:, 1] != 2))[0][:600]
    ids = (1000 + sel * 3).tolist()
    s = LOPQSearcher(m)
    s.add_codes([codes[i] for i in sel], ids)
    n1 = s.get_nb_indexed()
    # re-add the first 50 (same ids -> no-ops) and 20 copies of item sel[0] under fresh ids (ties)
    s.add_codes([codes[i] for i in sel[:50]], ids[:50])
    n2 = s.get_nb_indexed()
    dup_ids = list(range(900000, 900020))
    s.add_codes([codes[sel[0]]] * 20, dup_ids)
    d = model_arrays(m)
    d.update(inputs_sha1=np.array(sha1(X) + sha1(Q)), coarse=coarse, fine=fine, sel=sel,
             ids=np.array(ids, dtype=np.int64), dup_ids=np.array(dup_ids, dtype=np.int64),
             nb_after_first=np.int64(n1), nb_after_readd=np.int64(n2), nb_indexed=np.int64(s.get_nb_indexed()))
    Qx = np.concatenate([Q, X[sel[:4]]])  # the last queries hit the duplicated item exactly
    d.update(run_searches(s, Qx, [(1, None), (1, 5), (10, 10), (50, None), (100000, 40)], multiseq_n=16))
    d.update(aux_outputs(m, X))
    np.savez_compressed(os.path.join(outdir, "tiny.npz"), **d)


MAKERS = {"tiny": make_tiny, "c1": make_c1, "c2": make_c2, "c3": make_c3, "c3b": make_c3b, "c4": make_c4,
          "c3full": make_c3full, "pk": make_pk}

if __name__ == "__main__":
    tmp = import_reference()
    try:
        names = sys.argv[1:] or list(MAKERS)
        for n in names:
            MAKERS[n](HERE)
            out = os.path.join(HERE, n + ".npz")
            print("wrote", n, (os.path.getsize(out) // 1024) if os.path.exists(out) else "", "KiB")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
