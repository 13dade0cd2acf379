#!/usr/bin/env python3
"""Headline benchmark: LOPQ queries/sec at recall@10 on a 10M-vector index (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the search hot path over one batch of 8192 synthetic queries that are
already resident in HBM: PCA -> coarse ranking -> multisequence plan -> ADC tables -> ADC scan +
top-k -> merge (-> RCCL all-gather + merge when the index is sharded by coarse cell over N GPUs).
Workload = BASELINE config C4 (10M x 128-d, LOPQModelPCA V=16, M=8, renorm) with the reference
API's operating point quota=10000, limit=100 (cufacesearch/searcher/searcher_lopqhbase.py:833-838).
The data are descriptor-like (anisotropic mixture with a decaying spectrum, codes almost all distinct);
the LOPQ model is the one the reference itself fitted on this generator (tests/golden/c4.npz).

N > 1 is strong scaling: the same 10M index is sharded by coarse cell, every rank sees the whole
query batch, scans its own cells and the per-shard top-`limit` lists are all-gathered over RCCL.

Extra objects in the JSON line: "roofline" for the ADC scan kernel (algorithmic bytes =
candidates x M, time from HIP events recorded on the launch stream inside the library) and
"cpu_baseline" = the oracle (numpy restatement of the reference, reference-shaped per-candidate
loop, 1 core) timed on this host on a bounded sample of the same queries, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402

NQ = 8192          # queries per step
QUOTA, LIMIT = 10000, 100
N_CHUNKS = 80      # the database is generated in 80 equal chunks with per-chunk seeds
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def load_model():
    from columbiaimagesearch_amd.lopq import LOPQModelPCA
    z = np.load(os.path.join(REPO, "tests", "golden", "c4.npz"))
    nf = int(z["num_fine_splits"])
    subs = tuple([z["subs"][s, j] for j in range(nf)] for s in range(2))
    params = ((z["Cs"][0], z["Cs"][1]), (z["Rs"][0], z["Rs"][1]), (z["mus"][0], z["mus"][1]), subs,
              z["pca_P"], z["pca_mu"])
    return LOPQModelPCA(renorm=bool(z["renorm"]), parameters=params), z


def mixture_centers(device="cpu"):
    """Parameters of the descriptor-like generator (tests/golden_inputs.descriptor_params: the distribution the
    reference fitted tests/golden/c4.npz on), as tensors on `device`."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import golden_inputs as gi
    basis, scale, centers, mean = gi.descriptor_params(128)
    return {"noise_map": torch.as_tensor(scale[:, None] * basis.T, device=device),  # z -> (z * scale) . basis^T
            "centers": torch.as_tensor(centers, device=device), "mean": torch.as_tensor(mean, device=device)}


def gen_chunk(P, chunk, n, device):
    """n unit-norm float64 128-d vectors of chunk `chunk` (identical on every rank)."""
    g = torch.Generator(device=device)
    g.manual_seed(1000 + chunk)
    comp = torch.randint(0, P["centers"].shape[0], (n,), generator=g, device=device)
    z = torch.randn((n, P["centers"].shape[1]), generator=g, device=device, dtype=torch.float64)
    x = P["mean"] + 0.7 * P["centers"][comp] + 0.7 * (z @ P["noise_map"])
    return x / x.norm(dim=1, keepdim=True)


def make_queries(x0, batch, nq, device):
    """Perturbed database points of chunk 0 (so that a true neighbour exists)."""
    g = torch.Generator(device=device)
    g.manual_seed(77000 + batch)
    idx = torch.randint(0, x0.shape[0], (nq,), generator=g, device=device)
    q = x0[idx] + 0.05 * torch.randn((nq, x0.shape[1]), generator=g, device=device, dtype=torch.float64) / np.sqrt(x0.shape[1])
    return (q / q.norm(dim=1, keepdim=True)).contiguous()


def exact_nn(queries, centers_dev, n_total, chunk_n, device):
    """True nearest neighbour ids (exact L2 on the unit sphere = max dot), streamed over chunks."""
    best = torch.full((queries.shape[0],), -2.0, device=device, dtype=torch.float32)
    arg = torch.zeros(queries.shape[0], dtype=torch.int64, device=device)
    qf = queries.float()
    for c in range(n_total // chunk_n):
        x = gen_chunk(centers_dev, c, chunk_n, device).float()
        s = qf @ x.t()
        v, i = s.max(dim=1)
        upd = v > best
        best = torch.where(upd, v, best)
        arg = torch.where(upd, i + c * chunk_n, arg)
    return arg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=int(os.environ.get("CIS_BENCH_N", 10_000_000)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cnn", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if os.environ.get("CIS_BENCH_ONE_DEVICE") == "1":
        local_rank = 0  # functional test of the N > 1 protocol on a 1-GPU box (with CIS_BENCH_BACKEND=gloo: RCCL refuses shared devices)
    backend = os.environ.get("CIS_BENCH_BACKEND", "nccl")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    from columbiaimagesearch_amd import _lib
    from columbiaimagesearch_amd.distributed import ShardedSearcher, greedy_cell_owner
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    _lib.check(_lib.lib().cis_set_device(local_rank))
    # CIS_BENCH_FORCE_DIST=1 runs the sharded code path (process group, all-gather, merge) even with one
    # rank -- a smoke test of the N > 1 path on a 1-GPU box
    use_dist = world > 1 or os.environ.get("CIS_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        if "RANK" not in os.environ:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(backend, rank=0, world_size=1, device_id=device if backend == "nccl" else None)
        else:
            dist.init_process_group(backend, device_id=device if backend == "nccl" else None)  # nccl == RCCL on ROCm

    model, z = load_model()
    N = args.n - args.n % (N_CHUNKS * world)
    chunk_n = N // N_CHUNKS
    centers = mixture_centers(device)

    # ---- build: data-parallel encode on the GPUs, codes all-gathered, index sharded by cell -----
    t_build = time.time()
    my_chunks = [c for c in range(N_CHUNKS) if c * world // N_CHUNKS == rank]
    coarse_l, fine_l, ev = [], [], []
    for c in my_chunks:
        x = gen_chunk(centers, c, chunk_n, device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()  # predict_batch_dev launches on torch's current stream: these events bracket the encode kernels only
        co, fi = model.predict_batch_dev(x)
        e1.record()
        ev.append((e0, e1))
        coarse_l.append(co)
        fine_l.append(fi)
    coarse = torch.cat(coarse_l)
    fine = torch.cat(fine_l)
    torch.cuda.synchronize()
    encode_s = sum(a.elapsed_time(b) for a, b in ev) / 1e3  # without the synthetic data generation
    if world > 1:
        from columbiaimagesearch_amd.distributed import all_gather_stack
        coarse = all_gather_stack(coarse).reshape(-1, 2)
        fine = all_gather_stack(fine).reshape(-1, fine.shape[1])
    coarse_h = coarse.cpu().numpy().view(np.uint16)
    fine_h = fine.cpu().numpy()
    V = model.V
    cell = coarse_h[:, 0].astype(np.int64) * V + coarse_h[:, 1]
    counts = np.bincount(cell, minlength=V * V)
    if use_dist:
        # cells -> ranks by greedy balance of the cell populations (identical table on every rank)
        sharded = ShardedSearcher(model, owner=greedy_cell_owner(counts, world))
        searcher = sharded.local
    else:
        sharded = None
        searcher = LOPQSearcherHIP(model)
    searcher.add_codes_array(coarse_h, fine_h, ids=np.arange(N, dtype=np.int64), dedup=False)
    build_s = time.time() - t_build

    # ---- queries (resident in HBM before the timed region) --------------------------------------
    x0 = gen_chunk(centers, 0, chunk_n, device)
    n_batches = args.warmup + args.steps
    qbatches = [make_queries(x0, b, NQ, device) for b in range(min(n_batches, 8))]

    def step(q):
        if sharded is None:
            return searcher.search_batch_dev(q, quota=QUOTA, limit=LIMIT)
        return sharded.search_batch_dev(q, quota=QUOTA, limit=LIMIT)  # partial scan -> RCCL all-gather -> merge

    for b in range(args.warmup):
        step(qbatches[b % len(qbatches)])
    # timed region: only the pair of HIP events around the scan kernel (roofline); the per-stage events are small bubbles
    # between kernels, so the stage breakdown is taken from a few extra steps after the timed region
    searcher.set_profiling(True, scan_only=True)
    searcher.read_profile()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cand = 0
    out = None
    for b in range(args.steps):
        out = step(qbatches[(args.warmup + b) % len(qbatches)])
        cand += searcher.last_stats()["candidates"]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    prof = searcher.read_profile()
    searcher.set_profiling(True)
    n_stage = min(args.steps, 5)
    for b in range(n_stage):
        step(qbatches[b % len(qbatches)])
    torch.cuda.synchronize()
    stage_prof = searcher.read_profile()
    searcher.set_profiling(False)
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ct = torch.tensor([cand], device=device, dtype=torch.int64)
        dist.all_reduce(ct)
        cand_all = int(ct.item())
    else:
        cand_all = cand

    # ---- recall@10 (lopq/lopq/eval.py:92-143 semantics), untimed, rank 0 -----------------------
    recall10 = None
    qr = qbatches[0][:1024].contiguous()
    res = step(qr)
    if rank == 0:
        nn = exact_nn(qr, centers, N, chunk_n, device)
        recall10 = float((res["ids"][:, :10] == nn[:, None]).any(dim=1).float().mean().item())

    # ---- CPU baseline + parity spot check: oracle on a bounded sample (rank 0, N=1) ------------
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import lopq_oracle as O
        om = O.OracleModel.from_npz(z)
        oix = O.OracleCSRIndex(om, coarse_h, fine_h)
        qh = qr.cpu().numpy()
        gi, gd = res["ids"].cpu().numpy(), res["dists"].cpu().numpy()
        n_loop, t_loop, ok, max_rel = 0, 0.0, True, 0.0
        while t_loop < 12.0 and n_loop < 256:
            tq = time.perf_counter()
            ids, dd, _ = oix.search_loop(qh[n_loop], quota=QUOTA, limit=LIMIT)
            t_loop += time.perf_counter() - tq
            ok = ok and bool((gi[n_loop, :len(ids)] == ids).all())
            max_rel = max(max_rel, float(np.max(np.abs(gd[n_loop, :len(ids)] - dd) / np.maximum(dd, 1e-300))))
            n_loop += 1
        n_vec, t_vec = 0, 0.0
        while t_vec < 5.0 and n_vec < 1024:
            tq = time.perf_counter()
            ids, dd, _ = oix.search(qh[n_vec], quota=QUOTA, limit=LIMIT)
            t_vec += time.perf_counter() - tq
            ok = ok and bool((gi[n_vec, :len(ids)] == ids).all())
            n_vec += 1
        cpu = {"value": n_loop / t_loop, "unit": "queries/s", "cores": 1, "kind": "port",
               "sample": "%d queries of the timed workload (quota=%d, limit=%d, 10M index) through the oracle's "
                         "reference-shaped per-candidate loop (search.py:166-175); vectorised numpy restatement: "
                         "%.1f queries/s on %d queries" % (n_loop, QUOTA, LIMIT, n_vec / t_vec, n_vec)}
        parity = {"queries_checked": max(n_loop, n_vec), "ids_bit_exact": ok, "max_rel_dist_err": max_rel}

    # ---- second half of the BASELINE metric: CNN descriptors/s (DeepSentibank forward, batch 256) -------
    cnn = None
    if rank == 0 and not args.no_cnn:
        from columbiaimagesearch_amd.featurizer.synthetic import sentibank_weights as synthetic_weights  # seeded (trained weights are not in the tree)
        from columbiaimagesearch_amd.featurizer import SentiBankNet
        del x0
        torch.cuda.empty_cache()
        net = SentiBankNet(synthetic_weights(0))
        B = 256
        gcn = torch.Generator(device=device)
        gcn.manual_seed(5)
        xb = (torch.randn((B, 3, 227, 227), generator=gcn, device=device) * 50.0).contiguous()
        ob = torch.empty((B, 4096), device=device)
        for _ in range(2):
            net.forward_dev(xb, ob)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        reps = 8
        for _ in range(reps):
            net.forward_dev(xb, ob)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - tc) / reps
        flop = 2.0 * 720310816 * B
        cnn = {"metric": "CNN descriptors/sec (DeepSentibank forward to fc7, batch 256, synthetic weights)",
               "value": B / dt, "unit": "descriptors/s", "ms_per_batch": dt * 1e3, "dtype": "f32",
               "roofline": {"bound": "mfma", "achieved": flop / dt / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                            "frac": flop / dt / 157.3e12, "flop_per_image": 2 * 720310816}}
        net.close()

    if rank == 0:
        M = model.M
        launches = max(prof["scan_launches"], 1)
        scan_s = prof["scan_kernel_ms"] / 1e3  # HIP events right around the k_adc_scan2 launches, on their stream
        algo_bytes = cand * M  # this rank's scan kernel
        achieved = algo_bytes / scan_s / 1e9 if scan_s > 0 else 0.0
        traffic = None
        tpath = os.path.join(REPO, "profiles", "scan_traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        line = {
            "metric": "queries/sec @ recall@10 on 10M LOPQ index",
            "value": NQ * args.steps / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "recall_at_10": recall10,
            "config": {"workload": "C4: %d x 128-d float64 unit vectors (descriptor-like anisotropic mixture), LOPQModelPCA V=16 M=8 "
                                   "renorm, %d queries/step, quota=%d limit=%d" % (N, NQ, QUOTA, LIMIT),
                       "index_vectors": N, "queries_per_step": NQ, "quota": QUOTA, "limit": LIMIT,
                       "sharding": "by coarse cell over %d GPU(s)%s" % (world, ", RCCL all-gather merge" if world > 1 else ""),
                       "candidates_per_query": cand_all / float(NQ * args.steps)},
            "roofline": {"bound": "hbm", "kernel": "k_adc_scan2", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes / launches,
                         "avg_launch_ms": prof["scan_kernel_ms"] / launches, "launches": launches},
            "stage_ms_per_step": {k: stage_prof[k] / n_stage for k in ("front_ms", "tables_ms", "scan_ms", "merge_ms", "scan_kernel_ms")},
            "cnn": cnn,
            "cpu_baseline": cpu,
            "parity": parity,
            "build": {"encode_s": encode_s, "total_s": build_s, "encode_vectors_per_s": len(my_chunks) * chunk_n / encode_s},
        }
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the last thing on stdout: flush what native libraries (the RCCL banner) still hold in C stdio first
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
