#!/usr/bin/env python3
"""CPU baselines timed beside the GPU path (bench.py's ``cpu_baseline`` leg).  TEST INFRASTRUCTURE ONLY.

Worker processes of the all-core baselines SURVEY.md section 8(d) asks for: the reference's own deployment is N
single-threaded Python processes (gunicorn x16 for search, 14-16 extractor processes: conf/conf_extr_*_release.json),
so "all cores" = one single-threaded worker per core, each running the oracle's restatement for a fixed time budget:

    python oracle/cpu_bench.py search  <dir> <seconds> <worker> <n_workers>   vectorised numpy search (OracleCSRIndex.search)
    python oracle/cpu_bench.py encode  <dir> <seconds> <worker> <n_workers>   vectorised numpy encode (compute_codes)
    python oracle/cpu_bench.py cnn1    <dir> <seconds> <worker> <n_workers>   torch-CPU DeepSentibank, batch 1, 1 thread

<dir> holds what bench.py exported: model.npz (fixture layout), queries.npy, fine_sorted.npy / ids_sorted.npy /
offsets.npy (cell-contiguous index, memory-mapped so the workers share the pages), enc_x.npy.  Each worker prints one
JSON line {"n": units done, "s": seconds}.  `run_pool` launches the workers and adds them up.
"""
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _single_thread_env():
    env = dict(os.environ)
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        env[k] = "1"
    env["HIP_VISIBLE_DEVICES"] = ""  # the workers never touch the GPU
    env["CUDA_VISIBLE_DEVICES"] = ""
    return env


def run_pool(mode, workdir, seconds, n_workers):
    """Start n_workers single-threaded workers at once; -> (units per second over the pool, units, slowest worker s)."""
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), mode, workdir, str(seconds), str(w), str(n_workers)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=_single_thread_env(), cwd=REPO)
             for w in range(n_workers)]
    n, slowest = 0, 0.0
    for p in procs:
        out, _ = p.communicate()
        line = [l for l in out.decode().splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            raise RuntimeError("cpu_bench worker failed (mode %s)" % mode)
        r = json.loads(line[-1])
        n += r["n"]
        slowest = max(slowest, r["s"])
    return n / slowest, n, slowest


def export_index(workdir, oracle_index, queries, model_npz_path, enc_x=None):
    import numpy as np
    os.makedirs(workdir, exist_ok=True)
    np.save(os.path.join(workdir, "fine_sorted.npy"), oracle_index.fine)
    np.save(os.path.join(workdir, "ids_sorted.npy"), oracle_index.ids)
    np.save(os.path.join(workdir, "offsets.npy"), oracle_index.offsets)
    np.save(os.path.join(workdir, "queries.npy"), queries)
    if enc_x is not None:
        np.save(os.path.join(workdir, "enc_x.npy"), enc_x)
    with open(os.path.join(workdir, "model_path.txt"), "wt") as f:
        f.write(model_npz_path)


def _model(workdir):
    import numpy as np
    from oracle import lopq_oracle as O
    z = np.load(open(os.path.join(workdir, "model_path.txt")).read().strip())
    return O.OracleModel.from_npz(z)


def _worker(mode, workdir, seconds, w, nw):
    sys.path.insert(0, REPO)
    import numpy as np
    t_budget = float(seconds)
    if mode == "search":
        from oracle import lopq_oracle as O
        ix = O.OracleCSRIndex.__new__(O.OracleCSRIndex)
        ix.model = _model(workdir)
        ix.fine = np.load(os.path.join(workdir, "fine_sorted.npy"), mmap_mode="r")
        ix.ids = np.load(os.path.join(workdir, "ids_sorted.npy"), mmap_mode="r")
        ix.offsets = np.load(os.path.join(workdir, "offsets.npy"))
        Q = np.load(os.path.join(workdir, "queries.npy"))
        quota, limit = int(os.environ.get("CIS_CPU_QUOTA", 10000)), int(os.environ.get("CIS_CPU_LIMIT", 100))
        ix.search(Q[w % len(Q)], quota=quota, limit=limit)  # warm the page cache / numpy
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < t_budget:
            ix.search(Q[(w + n * nw) % len(Q)], quota=quota, limit=limit)
            n += 1
        return n, time.perf_counter() - t0
    if mode == "encode":
        from oracle import lopq_oracle as O
        m = _model(workdir)
        X = np.load(os.path.join(workdir, "enc_x.npy"))
        blk = 2048
        O.compute_codes(m, X[:blk])
        n, t0, a = 0, time.perf_counter(), (w * blk) % max(len(X) - blk, 1)
        while time.perf_counter() - t0 < t_budget:
            O.compute_codes(m, X[a:a + blk])
            n += min(blk, len(X) - a)
            a = (a + blk * nw) % max(len(X) - blk, 1)
        return n, time.perf_counter() - t0
    if mode == "cnn1":
        import torch
        torch.set_num_threads(1)
        from oracle import cnn_oracle as C
        wts = C.synthetic_weights(0)
        x = C.synthetic_images(1, seed=w)
        C.forward_torch(x, wts)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < t_budget:
            C.forward_torch(x, wts)
            n += 1
        return n, time.perf_counter() - t0
    raise SystemExit("unknown mode " + mode)


if __name__ == "__main__":
    n, s = _worker(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]))
    print(json.dumps({"n": n, "s": s}))
