#!/usr/bin/env python3
"""Headline benchmark: LOPQ queries/sec at recall@10 on a 10M-vector index (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the search hot path over one batch of 8192 synthetic queries that are
already resident in HBM: PCA -> coarse ranking -> multisequence plan -> ADC tables -> ADC scan +
top-k -> merge (-> RCCL all-gather + merge when the index is sharded by coarse cell over N GPUs).
Operating point of the reference API: quota=10000, limit=100 (cufacesearch/searcher/searcher_lopqhbase.py:833-838).

--config selects the BASELINE.json workload of the HEADLINE line (default c4 = the one the metric is quoted on):
  c4  10M x 128-d float64, LOPQModelPCA V=16 M=8 renorm (model fitted by the reference: tests/golden/c4.npz);
      descriptor-like data (anisotropic mixture with a decaying spectrum, codes almost all distinct)
  c2  the same generator and model at 1M vectors (dlib-descriptor shape)
  c3  1M x 4096-d float32 >= 0 (post-ReLU-like), LOPQModelPCA 4096 -> 256, V=16 M=16 renorm (model fitted by the
      reference on this generator: tests/golden/c3full.npz)
Without --config, at N = 1, the line also carries "configs": {"c2", "c3", "c5"} -- short runs of the other single-GPU
BASELINE configurations (their own ms/step, scan roofline, parity flag against the oracle) and the C5 chain at its true
shapes (DeepSentibank batch 256 -> L2 normalise -> PCA 4096 -> 256, V=16 M=16 encode -> insert into the resident c3 index).

N > 1: the HEADLINE is BASELINE config C4's layout -- the same 10M index sharded by coarse cell over all N GPUs (S = N cell
shards, one query group) -- answered by the ROUTED protocol (distributed.RoutedSearcher, round 5): a step is N x 8192 queries over
the job, every rank the home of 8192; a query goes to the owners of the cells it visits only (one all-to-all out, one back):
"scaling": "weak" (per-GPU work fixed as N grows: the query load grows, the index does not).  It is the headline only after it
reproduced the all-gather protocol's answers on a batch; that protocol (every rank sees the whole batch of 8192
queries, scans its own cells, the per-shard top-`limit` lists are all-gathered over RCCL and merged: strong scaling) stays in the
line as "allgather", and is the headline with --no-routed or when the routed leg fails.  A further
object "grid" in the same line carries the R x S layout (R query groups, each one copy of the index sharded over S GPUs;
--cell-shards, default 2 from 4 GPUs on, whole copies at 2): its value counts every group's queries (weak in the query load).

Extra objects in the JSON line: "roofline" for the ADC scan kernel (algorithmic bytes = candidates x M, time from
HIP events recorded on the launch stream inside the library; `frac` is SURVEY.md 8(d)'s accounting and can exceed 1 -- see
`binding` for what the kernel is bound by); "cpu_baseline" = the oracle (numpy restatement of the
reference) timed on this host on bounded samples of the same workload, rank 0 at N=1 only: `value` is the
reference-shaped per-candidate loop on 1 core, the other fields are the all-core vectorised search, encode
(1 core loop / all cores) and the torch-CPU DeepSentibank forward (batch 1 x cores, batch 256); "cnn" / "dlib" =
the descriptor networks (MFMA roofline); "pcie_inclusive" = the same step through the host-pointer entry point.
"""
import argparse
import contextlib
import json
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before the HIP runtime initialises: see columbiaimagesearch_amd/_lib.py
import sys
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402

NQ = 8192          # queries per step
QUOTA, LIMIT = 10000, 100
N_CHUNKS = 80      # the database is generated in 80 equal chunks with per-chunk seeds
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
F32_MFMA_PEAK_TFLOPS = 157.3
SENTIBANK_MAC, DLIB_MAC = 720310816, 270854144  # multiply-accumulates per image / face (oracle/cnn_oracle.py, oracle/dlib_oracle.py)

CONFIGS = {
    "c4": {"n": 10_000_000, "fixture": "c4", "gen": "descriptor", "d_in": 128, "label": "C4"},
    "c2": {"n": 1_000_000, "fixture": "c4", "gen": "descriptor", "d_in": 128, "label": "C2"},
    "c3": {"n": 1_000_000, "fixture": "c3full", "gen": "relu_mixture", "d_in": 4096, "label": "C3"},
}


def load_model(fixture):
    from columbiaimagesearch_amd.lopq import LOPQModelPCA
    z = np.load(os.path.join(REPO, "tests", "golden", fixture + ".npz"))
    nf = int(z["num_fine_splits"])
    subs = tuple([z["subs"][s, j] for j in range(nf)] for s in range(2))
    params = ((z["Cs"][0], z["Cs"][1]), (z["Rs"][0], z["Rs"][1]), (z["mus"][0], z["mus"][1]), subs,
              z["pca_P"], z["pca_mu"])
    return LOPQModelPCA(renorm=bool(z["renorm"]), parameters=params), z


def mixture_centers(gen, device="cpu"):
    """Parameters of the synthetic generators, as tensors on `device`: "descriptor" = tests/golden_inputs.descriptor_params
    (the distribution the reference fitted tests/golden/c4.npz on); "relu_mixture" = golden_inputs.gmm_unit(.., 4096, 64,
    seed 6, nonneg) whose 64 centres the reference fitted tests/golden/c3full.npz on."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import golden_inputs as gi
    if gen == "relu_mixture":
        centers = np.random.RandomState(6).randn(64, 4096)
        return {"gen": gen, "centers": torch.as_tensor(centers, device=device, dtype=torch.float32)}
    basis, scale, centers, mean = gi.descriptor_params(128)
    return {"gen": gen, "noise_map": torch.as_tensor(scale[:, None] * basis.T, device=device),  # z -> (z * scale) . basis^T
            "centers": torch.as_tensor(centers, device=device), "mean": torch.as_tensor(mean, device=device)}


def gen_chunk(P, chunk, n, device):
    """n unit-norm vectors of chunk `chunk` (identical on every rank): float64 128-d, or float32 4096-d >= 0."""
    g = torch.Generator(device=device)
    g.manual_seed(1000 + chunk)
    comp = torch.randint(0, P["centers"].shape[0], (n,), generator=g, device=device)
    if P["gen"] == "relu_mixture":
        x = torch.randn((n, P["centers"].shape[1]), generator=g, device=device, dtype=torch.float32)
        x = torch.clamp_(x.mul_(0.35).add_(P["centers"][comp]), min=0.0)
        return x.div_(x.norm(dim=1, keepdim=True).clamp_(min=1e-12))
    z = torch.randn((n, P["centers"].shape[1]), generator=g, device=device, dtype=torch.float64)
    x = P["mean"] + 0.7 * P["centers"][comp] + 0.7 * (z @ P["noise_map"])
    return x / x.norm(dim=1, keepdim=True)


def make_queries(x0, batch, nq, device):
    """Perturbed database points of chunk 0 (so that a true neighbour exists)."""
    g = torch.Generator(device=device)
    g.manual_seed(77000 + batch)
    idx = torch.randint(0, x0.shape[0], (nq,), generator=g, device=device)
    q = x0[idx] + 0.05 * torch.randn((nq, x0.shape[1]), generator=g, device=device, dtype=x0.dtype) / np.sqrt(x0.shape[1])
    return (q / q.norm(dim=1, keepdim=True)).contiguous()


def exact_nn(queries, centers_dev, n_total, chunk_n, device, pca=None):
    """True nearest neighbour ids in the space LOPQ works in (lopq/lopq/eval.py:92-143): exact L2 between unit vectors
    = max dot, streamed over chunks; `pca` = (mu, P) float32 tensors applies (x - mu) . P + renormalisation first."""
    def space(x):
        x = x.float()
        if pca is not None:
            x = (x - pca[0]) @ pca[1]
            x = x / x.norm(dim=1, keepdim=True)
        return x
    best = torch.full((queries.shape[0],), -2.0, device=device, dtype=torch.float32)
    arg = torch.zeros(queries.shape[0], dtype=torch.int64, device=device)
    qf = space(queries)
    for c in range(n_total // chunk_n):
        x = space(gen_chunk(centers_dev, c, chunk_n, device))
        s = qf @ x.t()
        v, i = s.max(dim=1)
        upd = v > best
        best = torch.where(upd, v, best)
        arg = torch.where(upd, i + c * chunk_n, arg)
    return arg


# ---- a phase that hangs (a collective whose peer never arrives) must say which one it was -------------------------------------
class Watchdog(object):
    """`with wd.phase("all-gather of the packed hits", 120): ...` -- a phase that outlives its allowance prints its name on
    stderr and ends the process (exit code 3) instead of hanging the node until the driver's limit."""

    def __init__(self, rank):
        self.rank = rank
        self._lock = threading.Lock()
        self._name, self._deadline = None, None
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def _run(self):
        while True:
            time.sleep(1.0)
            with self._lock:
                name, dl = self._name, self._deadline
            if name is not None and time.monotonic() > dl:
                sys.stderr.write("[bench rank %d] phase '%s' did not finish within its allowance -- a rank is missing from a "
                                 "collective or a kernel hangs; giving up\n" % (self.rank, name))
                sys.stderr.flush()
                os._exit(3)

    @contextlib.contextmanager
    def phase(self, name, seconds):
        with self._lock:
            prev = (self._name, self._deadline)
            self._name, self._deadline = name, time.monotonic() + seconds
        try:
            yield
        finally:
            with self._lock:
                self._name, self._deadline = prev


class Ctx(object):
    pass


def pipe_streams(device, n):
    """The process's lane streams (columbiaimagesearch_amd/streams.py): made back to back at the first call, each with a first submission
    right away.  The GPU dispatches from four hardware pipes and HIP deals a process's streams onto them in the order they first submit
    work (queue number modulo 4: tools/r06_queue_probe.py) -- two streams on one pipe do not overlap.  Streams made one after the other sit
    on different pipes; streams made leg by leg, after a varying number of others, may not (round 6: four search lanes moved the CNN
    lanes of a later leg onto shared pipes, dlib three in flight 0.61 -> 0.55 of the MFMA peak).  Every leg with batches in flight takes
    its streams from here."""
    from columbiaimagesearch_amd.streams import lane_streams
    lane_streams(4, device)
    return lane_streams(n, device)


def pmc_source(path, d):
    """Where a replayed PMC summary came from: file, the commit it was copied in at, the kernel symbol, and whether the kernel sources it
    was measured on are the ones this build was compiled from (sha1 of the csrc files, recorded on the GPU box by the collecting tool)."""
    out = {"file": os.path.relpath(path, REPO), "commit": d.get("commit"), "kernel": d.get("kernel")}
    want = d.get("kernel_sources_sha1")
    if want:
        sys.path.insert(0, os.path.join(REPO, "tools"))
        try:
            from pmc_stamp import kernel_sources_sha1
            have = kernel_sources_sha1()
            changed = sorted(f for f in want if want[f] != have.get(f))
            out["sources_match_this_build"] = not changed
            if changed:
                out["changed_since"] = changed
        except Exception as e:
            out["sources_match_this_build"] = None
            out["error"] = repr(e)
    else:
        out["sources_match_this_build"] = None   # a summary from before round 6: no record of the kernel build
    return out


def scan_binding(cfg_name, scan_name, cand, cand_items, M, scan_s, launches):
    """What binds the scan kernel.  `model`: the share of the launch time that each resource's MINIMUM accounts for (conflict-free
    LDS, the bare gather-and-add loop) -- a lower bound per resource.  `measured`: the SQ counters of the same kernel on the same
    workload from this round's rocprofv3 --pmc passes (profiles/scan_binding_<config>.json, made by tools/pmc_scan.py: VALU
    busy, LDS busy incl. bank conflicts) -- PMC counters need their own passes and cannot be read inside a timed run."""
    G = {"k_adc_scan2": 2, "k_adc_scan3": 4, "k_adc_scan4": 4}.get(scan_name, 1)
    K = 256
    pairs_rows = cand / float(G) / 64.0                 # wave-rows: 64 candidates x G queries
    moved = cand * M / float(G) + cand_items * M * K * 4 + cand_items * LIMIT * 8   # codes once per workgroup + float32 tables staged + survivors
    lds_cycles = pairs_rows * M * 2.0                   # one ds_read_b64 per sub-quantizer and row, 2 LDS cycles when conflict-free
    valu_instr = pairs_rows * ((2 * M + 4) if G == 2 else (3 * M + 4))  # address + packed add(s) per sub-quantizer, unpack + compare
    clk, n_cu = 2.4e9, 256
    t_lds = lds_cycles / (n_cu * clk)
    t_valu = valu_instr * 4.0 / (4 * n_cu * clk)        # a wave64 VALU instruction occupies its SIMD for 4 cycles
    t_mem = moved / (HBM_PEAK_GBS * 1e9)
    fr = (lambda t: t / scan_s) if scan_s > 0 else (lambda t: None)
    model = {"unit": "fraction of the launch time each resource's minimum accounts for (1.0 = bound by it)",
             "hbm_moved_bytes": {"bytes_per_launch": moved / launches, "frac": fr(t_mem),
                                 "note": "cand*M/G + float32 tables staged + survivors, against 8 TB/s (mostly L2 / Infinity Cache hits)"},
             "lds_gather": {"wave_reads_per_launch": pairs_rows * M / launches, "frac": fr(t_lds),
                            "note": "CONFLICT-FREE model: ds_read_b64 per (row of 64 candidates x G queries, sub-quantizer), 2 cycles each, 256 CUs at 2.4 GHz"},
             "valu_issue": {"wave_instr_per_launch": valu_instr / launches, "frac": fr(t_valu),
                            "note": "minimal gather-and-add loop only (no selection / append), 4 cycles per wave64 instruction, 1024 SIMDs"}}
    out = {"queries_per_workgroup": G, "model": model, "measured": None}
    p = os.path.join(REPO, "profiles", "scan_binding_%s.json" % cfg_name)
    if os.path.exists(p):
        try:
            out["measured"] = json.load(open(p))
            out["measured"]["source_file"] = "profiles/scan_binding_%s.json" % cfg_name
        except Exception as e:
            out["measured"] = {"error": repr(e)}
    fracs = {k: model[k]["frac"] or 0.0 for k in ("hbm_moved_bytes", "lds_gather", "valu_issue")}
    if out["measured"] and out["measured"].get("lds_conflict_ratio") is not None and model["lds_gather"]["frac"] is not None:
        # the same minimum at the MEASURED conflict ratio (conflict cycles / active cycles): what the gathers cost on this kernel
        cr = float(out["measured"]["lds_conflict_ratio"])
        if 0.0 <= cr < 1.0:
            model["lds_gather"]["frac_at_measured_conflicts"] = model["lds_gather"]["frac"] / (1.0 - cr)
    if out["measured"] and "valu_busy_frac" in out["measured"]:
        fracs = {"valu_issue": out["measured"]["valu_busy_frac"], "lds_gather": out["measured"].get("lds_busy_frac", 0.0),
                 "hbm_moved_bytes": out["measured"].get("hbm_frac", fracs["hbm_moved_bytes"])}
        out["binds_source"] = "measured"
    else:
        out["binds_source"] = "model"
    out["binds"] = max(fracs, key=lambda k: fracs[k] or 0.0)
    # binding_frac: the launch's MINIMUM time under the resource that binds it / the launch time -- <= 1 by construction, the roofline
    # fraction of a kernel that no byte count describes.  Minima of the NECESSARY work only: the bare gather-and-add instructions at the
    # VALU's issue rate (2 cycles per wave64 instruction and SIMD: MI355X_MICROARCH.md, tools/probes/valu_rate.hip), the gathers at the
    # MEASURED bank-conflict ratio (conflicts are the layout's, not the launch's, to lose), the measured fabric traffic at 8 TB/s.
    t_valu2 = valu_instr * 2.0 / (4 * n_cu * clk)
    mins = {"valu_issue": t_valu2, "lds_gather": t_lds, "hbm": t_mem}
    mm = out["measured"] or {}
    if mm.get("lds_conflict_ratio") is not None and 0.0 <= float(mm["lds_conflict_ratio"]) < 1.0:
        mins["lds_gather"] = t_lds / (1.0 - float(mm["lds_conflict_ratio"]))
    if mm.get("hbm_bytes_per_launch"):
        mins["hbm"] = float(mm["hbm_bytes_per_launch"]) * launches / (HBM_PEAK_GBS * 1e9)
    res = max(mins, key=lambda k: mins[k])
    out["binding_resource"] = res
    out["binding_min_ms_per_launch"] = {k: v / launches * 1e3 for k, v in mins.items()}
    out["binding_frac"] = min(1.0, mins[res] / scan_s) if scan_s > 0 else None
    if out["measured"] and "error" not in out["measured"]:
        out["pmc_source"] = pmc_source(p, out["measured"])
    return out


def search_leg(ctx, cfg_name, n_vectors, S, steps, warmup, scaling, oracle_rows):
    """Build the index of one configuration (encode on the GPUs, device-side insert, R x S layout with S cell shards) and time
    `steps` query batches.  Returns (result dict (rank 0's view), state for the follow-up legs)."""
    import torch.distributed as dist
    from columbiaimagesearch_amd.distributed import GridSearcher, greedy_cell_owner
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    cfg = CONFIGS[cfg_name]
    device, rank, world, wd = ctx.device, ctx.rank, ctx.world, ctx.wd
    model, z = load_model(cfg["fixture"])
    N = n_vectors * (world if scaling == "weak" else 1)
    N -= N % (N_CHUNKS * world)
    chunk_n = N // N_CHUNKS
    centers = mixture_centers(cfg["gen"], device)
    st = Ctx()
    st.cfg_name, st.cfg, st.model, st.z, st.N, st.chunk_n, st.centers = cfg_name, cfg, model, z, N, chunk_n, centers

    # ---- build: data-parallel encode on the GPUs, codes routed to the owners of their cells -----
    t_build = time.time()
    R = world // S                      # query groups
    g_q, s_c = rank // S, rank % S      # this rank: cell shard s_c of query group g_q
    my_slice = s_c * R + g_q            # GridSearcher.slice_number: keeps the cells in the order of a single index
    my_chunks = [c for c in range(N_CHUNKS) if c * world // N_CHUNKS == my_slice]
    coarse_l, fine_l, ev = [], [], []
    # encode parity at full size: rows sampled from EVERY chunk are kept (host copies) and their codes are checked against
    # the oracle's compute_codes after the timed region (the oracle index of the search spot check is built from HIP codes)
    n_sample = max(oracle_rows // N_CHUNKS + 1, 1) if oracle_rows else 0
    sample_x, sample_pos = [], []
    # the generator's chunks are encoded in groups of about 65536 vectors, the library's own pass size (one call per 12500-row
    # chunk of the 1M configurations leaves the chip half empty: c2 75 -> ~150 M, c3 8.7 -> ~10 M vectors/s)
    group = max(1, 65536 // chunk_n)
    with wd.phase("%s: encode of the index build" % cfg_name, 600):
        for g0 in range(0, len(my_chunks), group):
            xs = []
            for c in my_chunks[g0:g0 + group]:
                x = gen_chunk(centers, c, chunk_n, device)
                xs.append(x)
                if n_sample:
                    sel = torch.as_tensor(np.random.RandomState(4242 + c).choice(chunk_n, n_sample, replace=False), device=device)
                    sample_x.append(x[sel].cpu().numpy())
                    sample_pos.append(sel.cpu().numpy() + (c - my_chunks[0]) * chunk_n)
            x = xs[0] if len(xs) == 1 else torch.cat(xs)
            del xs
            if g0 == 0:  # one untimed call first: the workspaces' hipMalloc and the code-object load (41.6 ms cold against 3.6 ms: profiles/r04w_encode_calls.txt)
                model.predict_batch_dev(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()  # predict_batch_dev launches on torch's current stream: these events bracket the encode kernels only
            co, fi = model.predict_batch_dev(x)
            e1.record()
            ev.append((e0, e1))
            coarse_l.append(co)
            fine_l.append(fi)
            del x
        coarse = torch.cat(coarse_l)
        fine = torch.cat(fine_l)
        del coarse_l, fine_l
        torch.cuda.synchronize()
    encode_s = sum(a.elapsed_time(b) for a, b in ev) / 1e3  # without the synthetic data generation
    V = model.V
    # this rank encoded chunks [first, first + len(my_chunks)): ids are positions in the whole database
    ids_dev = torch.arange(my_chunks[0] * chunk_n, (my_chunks[-1] + 1) * chunk_n, dtype=torch.int64, device=device)
    t_ins = time.perf_counter()
    if ctx.use_dist:
        # R x S grid (distributed.GridSearcher): R query groups, each holding one copy of the index sharded by cell over S ranks.
        # cells -> shards by greedy balance of the cell populations: the table must be identical on every rank, so the
        # per-cell counts are summed over the ranks first (V*V int64); then every code travels once per copy, to its owner --
        # device buffers in, RCCL all-gather along the column + all-to-all inside the row, device-side merge into the owner's
        # index (no host copy of the codes)
        cell = coarse[:, 0].to(torch.int64).bitwise_and_(0xFFFF) * V + coarse[:, 1].to(torch.int64).bitwise_and_(0xFFFF)
        ct_all = torch.bincount(cell, minlength=V * V)
        del cell
        if ctx.backend != "nccl":
            ct_all = ct_all.cpu()
        with wd.phase("%s: all-reduce of the per-cell counts (V*V int64)" % cfg_name, 180):
            dist.all_reduce(ct_all)
        with wd.phase("%s: process groups of the %d x %d grid (dist.new_group)" % (cfg_name, R, S), 180):
            sharded = GridSearcher(model, S, owner=greedy_cell_owner(ct_all.cpu().numpy(), S) if S > 1 else None)
        assert sharded.slice_number == my_slice
        searcher = sharded.local
        with wd.phase("%s: routed insert (column all-gather + all-to-all of the code records)" % cfg_name, 300):
            sharded.add_codes_dev(coarse, fine, ids_dev, dedup=False)  # column all-gather, then routed inside the query group
            torch.cuda.synchronize()
    else:
        sharded = None
        searcher = LOPQSearcherHIP(model)
        searcher.add_codes_dev(coarse, fine, ids_dev, dedup=False)  # device-side merge (csrc/lopq_index.hip)
        torch.cuda.synchronize()
    insert_s = time.perf_counter() - t_ins
    st.coarse_h = st.fine_h = None
    if oracle_rows:  # host copies only for the oracle legs
        st.coarse_h = coarse.cpu().numpy().view(np.uint16)
        st.fine_h = fine.cpu().numpy()
    del coarse, fine, ids_dev
    build_s = time.time() - t_build
    st.sample_x, st.sample_pos, st.searcher, st.sharded = sample_x, sample_pos, searcher, sharded

    # ---- queries (resident in HBM before the timed region) --------------------------------------
    x0 = gen_chunk(centers, 0, chunk_n, device)
    n_batches = warmup + steps
    qbatches = [make_queries(x0, b, NQ, device) for b in range(min(n_batches, 8))]
    del x0
    st.qbatches = qbatches

    def step(q):
        if sharded is None:
            return searcher.search_batch_dev(q, quota=QUOTA, limit=LIMIT)
        return sharded.search_batch_dev(q, quota=QUOTA, limit=LIMIT)  # partial scan -> RCCL all-gather -> merge
    st.step = step
    # Steps are independent batches: `pipeline` of them are in flight at once, each through its own VIEW of the index (shared
    # codes / ids in HBM, private per-batch workspaces: cis_index_create_view) on its own HIP stream -- the launch-bound small
    # kernels of one batch's front end, tables and slot building fill the tail of the previous batch's scan and its merge.
    P = max(1, ctx.pipeline)
    if sharded is None:
        lanes = [(searcher, torch.cuda.current_stream(device))]
        for ls in pipe_streams(device, P - 1):
            lanes.append((searcher.view(), ls))
    elif sharded.row is not None:  # the sharded searcher rotates its partial searches over its own lanes (search_begin)
        sharded.row.pipeline_depth = P
        lanes = list(sharded.row.lanes())
    else:
        P = 1
        lanes = [(searcher, torch.cuda.current_stream(device))]
    st.lanes = lanes if sharded is None else lanes[:1]  # (release_state closes the views it owns; the sharded searcher keeps its own)

    def qb(b):  # a step = one batch of NQ queries PER QUERY GROUP: group g answers its own batches, the S ranks of a group the same
        return qbatches[(b * R + g_q) % len(qbatches)]

    with wd.phase("%s: warm-up steps (first collectives of the search path)" % cfg_name, 300):
        for b in range(warmup):
            step(qb(b))
        if sharded is not None and sharded.row is not None:  # every lane of the pipelined form once (workspaces of the views)
            for b in range(len(lanes)):
                sharded.search_end(sharded.search_begin(qb(b), quota=QUOTA, limit=LIMIT))
        torch.cuda.synchronize()
    # timed region: only the pair of HIP events around the scan kernel (roofline); the per-stage events are small bubbles
    # between kernels, so the stage breakdown is taken from a few extra steps after the timed region
    for sv, _ in lanes:
        sv.set_profiling(True, scan_only=True)
        sv.read_profile()
    if P > 1 and sharded is None:  # the views' workspaces warm up too
        for k in range(1, P):
            with torch.cuda.stream(lanes[k][1]):
                lanes[k][0].search_batch_dev(qb(0), quota=QUOTA, limit=LIMIT)
        torch.cuda.synchronize()
        for sv, _ in lanes:
            sv.read_profile()
    # every lane sees every distinct batch once before the timed region: the per-batch workspaces are grow-only, a growth is a hipFree
    # (waits for the device) + hipMalloc of up to hundreds of MB, and a batch with a few more candidates than the warm-up batch
    # grew them INSIDE the timed loop (the 4.33 ms/step c2 leg of profiles/archive/r04f: root cause)
    with wd.phase("%s: workspace warm-up of every lane" % cfg_name, 300):
        if sharded is None:
            for k in range(P):
                with torch.cuda.stream(lanes[k][1]):
                    for q in qbatches:
                        lanes[k][0].search_batch_dev(q, quota=QUOTA, limit=LIMIT)
            torch.cuda.synchronize()
        elif sharded.row is not None:
            for k in range(len(lanes) * len(qbatches)):
                sharded.search_end(sharded.search_begin(qbatches[k % len(qbatches)], quota=QUOTA, limit=LIMIT))
            torch.cuda.synchronize()
        for sv, _ in lanes:
            sv.read_profile()
    from columbiaimagesearch_amd import _lib as _cl

    def run_steps(first):
        """Exactly `steps` steps (batches first ... first + steps - 1); returns (candidates, work items) of this rank."""
        cand = cand_items = 0
        if sharded is None:
            for b in range(steps):
                sv, stream = lanes[b % P]
                with torch.cuda.stream(stream):
                    sv.search_batch_dev(qb(first + b), quota=QUOTA, limit=LIMIT)
                ls = sv.last_stats()
                cand += ls["candidates"]
                cand_items += ls["items"]
            return cand, cand_items
        # steps are independent batches: the exchange + merge of batch b (side stream, RCCL) overlap the partial search
        # of batch b+1 (compute stream)
        def stats_of(hh):
            ls = hh.get("searcher", searcher).last_stats() if isinstance(hh, dict) else searcher.last_stats()
            return ls["candidates"], ls["items"]
        h = sharded.search_begin(qb(first), quota=QUOTA, limit=LIMIT)
        c_, i_ = stats_of(h)
        cand += c_
        cand_items += i_
        # check=False: no host read in the exchange (fixed-size payload all-gather, offsets by a kernel); the overflow flags of
        # all steps are read once after the loop
        for b in range(1, steps):
            h2 = sharded.search_begin(qb(first + b), quota=QUOTA, limit=LIMIT)
            c_, i_ = stats_of(h2)
            cand += c_
            cand_items += i_
            st.exchange_flags.append(sharded.search_end(h, check=False).get("overflow"))
            h = h2
        st.exchange_flags.append(sharded.search_end(h, check=False).get("overflow"))
        return cand, cand_items

    # Timed region: REPETITIONS of exactly `steps` steps, each bracketed by barrier + synchronize on both sides (the contract's
    # bracket) and timed twice -- host wall clock (what `value` is made of: the max over ranks, then the MEDIAN over repetitions) and a
    # pair of HIP events around the same steps (start on the launch stream, end after every lane's stream joined it).  One
    # repetition of 20 steps is 8 ms: a single host hiccup used to decide the figure; repetitions continue until >= 0.5 s are
    # timed (at least 5, at most 400).  wall >> events, or a workspace allocation inside the region, marks the leg `suspect`.
    st.exchange_flags = []
    main_stream = torch.cuda.current_stream(device)
    lane_streams = [s_ for _, s_ in lanes if s_ is not None and s_ != main_stream] if sharded is None else \
                   [s_ for s_ in (sharded.row._lane_streams() if sharded.row is not None else [])]
    walls, evs, cand, cand_items, reps = [], [], 0, 0, 0
    min_reps = int(os.environ.get("CIS_BENCH_MIN_REPS", 5))
    min_timed_s = float(os.environ.get("CIS_BENCH_MIN_TIMED_S", 0.5))
    alloc0 = _cl.alloc_stats()
    with wd.phase("%s: timed steps" % cfg_name, 900):
        while True:
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(main_stream)
            c_, i_ = run_steps(warmup + reps * steps)
            for s_ in lane_streams:
                main_stream.wait_stream(s_)
            if sharded is not None and getattr(sharded.row, "_side", None) is not None:
                main_stream.wait_stream(sharded.row._side)
            e1.record(main_stream)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t1 = time.perf_counter()
            el = t1 - t0
            if world > 1:  # the slowest rank's time of THIS repetition; every rank then takes the same stop decision
                tdev = device if ctx.backend == "nccl" else "cpu"
                tt = torch.tensor([el], device=tdev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            walls.append(el)
            evs.append(e0.elapsed_time(e1) / 1e3)
            cand += c_
            cand_items += i_
            reps += 1
            if reps >= 400 or (reps >= min_reps and sum(walls) >= min_timed_s):
                break
        scan_name = searcher.last_stats()["scan_kernel"]
    alloc1 = _cl.alloc_stats()
    steps_total = steps * reps
    prof = searcher.read_profile()
    for sv, _ in lanes[1:]:
        pv = sv.read_profile()
        for k in prof:
            prof[k] += pv[k]
        sv.set_profiling(False)
    searcher.set_profiling(True)
    n_stage = max(5, min(steps, 20))  # one batch at a time: the scan launch alone on the chip (roofline.avg_launch_ms) and the stage split
    with wd.phase("%s: stage-profile steps" % cfg_name, 300):
        for b in range(n_stage):
            step(qb(b))
        torch.cuda.synchronize()
    stage_prof = searcher.read_profile()
    searcher.set_profiling(False)
    walls_s = sorted(walls)
    elapsed = walls_s[len(walls_s) // 2] if len(walls_s) % 2 else 0.5 * (walls_s[len(walls_s) // 2 - 1] + walls_s[len(walls_s) // 2])  # median repetition
    ev_s = sorted(evs)
    ev_med = ev_s[len(ev_s) // 2]
    n_alloc = alloc1[0] - alloc0[0]
    timing = {"repetitions": reps, "steps_per_repetition": steps, "timed_s": sum(walls),
              "ms_per_step": {"median": elapsed / steps * 1e3, "min": walls_s[0] / steps * 1e3, "max": walls_s[-1] / steps * 1e3},
              "hip_event_ms_per_step": {"median": ev_med / steps * 1e3, "min": ev_s[0] / steps * 1e3, "max": ev_s[-1] / steps * 1e3},
              "wall_over_events": elapsed / ev_med if ev_med > 0 else None,
              "outlier_repetitions": sum(1 for w in walls if w > 1.5 * elapsed),
              "workspace_allocations_in_timed_region": n_alloc, "workspace_bytes_allocated_in_timed_region": alloc1[1] - alloc0[1]}
    timing["suspect"] = bool(n_alloc > 0 or (ev_med > 0 and elapsed > 1.5 * ev_med + 0.2e-3 * steps))
    if timing["suspect"]:
        sys.stderr.write("[bench rank %d] %s: SUSPECT TIMING -- wall %.3f ms/step against %.3f ms/step between the HIP events, %d workspace "
                         "allocation(s) inside the timed region\n" % (rank, cfg_name, elapsed / steps * 1e3, ev_med / steps * 1e3, n_alloc))
    if world > 1:
        with wd.phase("%s: all-reduce of the candidate counts" % cfg_name, 120):
            tdev = device if ctx.backend == "nccl" else "cpu"
            ct = torch.tensor([cand], device=tdev, dtype=torch.int64)
            dist.all_reduce(ct)
            cand_all = int(ct.item())
    else:
        cand_all = cand

    # ---- recall@10 (lopq/lopq/eval.py:92-143 semantics), untimed, rank 0 -----------------------
    recall10 = None
    qr = qbatches[0][:1024].contiguous()
    with wd.phase("%s: recall batch" % cfg_name, 300):
        res = step(qr)
        torch.cuda.synchronize()
    st.qr, st.res = qr, res
    if rank == 0:
        pca = None
        if cfg["gen"] == "relu_mixture":  # the 4096 -> 256 PCA changes the metric: neighbours are defined after it
            pca = (torch.as_tensor(z["pca_mu"], device=device, dtype=torch.float32),
                   torch.as_tensor(z["pca_P"], device=device, dtype=torch.float32))
        nn = exact_nn(qr, centers, N, chunk_n, device, pca)
        recall10 = float((res["ids"][:, :10] == nn[:, None]).any(dim=1).float().mean().item())

    M = model.M
    launches = max(prof["scan_launches"], 1)   # scan launches of the timed region (several batches in flight)
    pipe_launch_ms = prof["scan_kernel_ms"] / launches
    iso_launches = max(stage_prof["scan_launches"], 1)
    iso_launch_ms = stage_prof["scan_kernel_ms"] / iso_launches   # the same launch alone on the chip (one batch at a time, right after the timed region)
    algo_bytes = cand * M  # this rank's scan kernel, SURVEY.md 8(d): every (candidate, query) pair counts M code bytes
    algo_per_launch = algo_bytes / launches
    achieved = algo_per_launch / (iso_launch_ms / 1e3) / 1e9 if iso_launch_ms > 0 else 0.0
    pipe_achieved = algo_per_launch / (pipe_launch_ms / 1e3) / 1e9 if pipe_launch_ms > 0 else 0.0
    # `frac` = SURVEY.md 8(d)'s accounting over the KERNEL's own duration (HIP events around the launch, one batch at a time), uncapped.
    # The accounting re-counts a code for every query although the G queries of a workgroup share one load and a 10M index streams
    # through L2 / Infinity Cache, so it can exceed 1 and is not a physical bound: `physical_frac` = the bytes that
    # really crossed the fabric (PMC, own rocprofv3 passes: profiles/scan_traffic_<config>.json) over the same duration, `binding`
    # = what the kernel runs against.  The physically HBM-bound regime is measured by the c4x leg (`configs.c4x`).
    traffic, traffic_note = None, "PMC passes are separate rocprofv3 runs: none committed for this config"
    tp = os.path.join(REPO, "profiles", "scan_traffic_%s.json" % cfg_name)
    if os.path.exists(tp):
        try:
            traffic = float(json.load(open(tp))["hbm_bytes_per_launch"])
            traffic_note = "from profiles/scan_traffic_%s.json (separate rocprofv3 --pmc passes of the same command, gfx950 corrections applied), per full launch" % cfg_name
        except Exception as e:
            traffic_note = "profiles/scan_traffic_%s.json unreadable: %r" % (cfg_name, e)
    Dm, hm = model.dim, model.dim // 2
    # encode: SURVEY.md 8(d)'s algorithmic flop per vector -- 2 D_in D (PCA) + 3 V D (coarse) + 4 h^2 (rotation) + 3 K D (fine) --
    # against the float64 peak (AMD's MI355X figure, 78.6 TFLOP/s vector = matrix; the arithmetic that decides a code is
    # float64 / numpy-ordered, the fine and large-V coarse stages prefilter on the float32 matrix cores and re-check exactly)
    enc_flop = 2.0 * cfg["d_in"] * Dm + 3.0 * model.V * Dm + 4.0 * hm * hm + 3.0 * 256 * Dm
    enc_rate = len(my_chunks) * chunk_n / encode_s
    result = {
        "value": R * NQ * steps / elapsed,
        "unit": "queries/s",
        "ms_per_step": elapsed / steps * 1e3,
        "steps": steps,
        "recall_at_10": recall10,
        "config": {"workload": "%s: %d x %d-d %s unit vectors (%s), LOPQModelPCA %d -> %d V=%d M=%d renorm (fitted by the "
                               "reference, tests/golden/%s.npz), %d queries/step, quota=%d limit=%d"
                               % (cfg["label"], N, cfg["d_in"], "float32 >= 0" if cfg["gen"] == "relu_mixture" else "float64",
                                  "post-ReLU-like 64-component mixture" if cfg["gen"] == "relu_mixture" else "descriptor-like anisotropic mixture",
                                  cfg["d_in"], model.dim, model.V, M, cfg["fixture"], NQ, QUOTA, LIMIT),
                   "name": cfg_name, "index_vectors": N, "queries_per_step": NQ, "quota": QUOTA, "limit": LIMIT,
                   "sharding": "%d query group(s) x %d cell shard(s): each group holds the whole index sharded by coarse cell over %d GPU(s)%s "
                               "and answers its own %d queries per step" % (R, S, S, ", RCCL all-gather merge inside the group" if S > 1 else "", NQ),
                   "parallelism": "grid %dx%d" % (R, S), "query_groups": R, "cell_shards": S, "queries_per_step_all_groups": R * NQ,
                   "batches_in_flight": P, "index_scaling": scaling, "query_load_scaling": "weak (x%d query groups)" % R if R > 1 else "fixed",
                   "candidates_per_query": cand_all / float(R * NQ * steps_total)},
        "roofline": {"bound": "hbm", "kernel": scan_name, "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": None, "accounting_frac": achieved / HBM_PEAK_GBS,
                     "frac_note": "frac = binding_frac: the launch's minimum time under the resource that binds it (LDS gathers at the measured "
                                  "conflict ratio / VALU issue of the bare gather-and-add loop / measured fabric bytes at 8 TB/s) over its duration, "
                                  "<= 1 by construction.  accounting_frac = achieved / peak = SURVEY.md 8(d)'s accounting (every (candidate, query) "
                                  "pair = M bytes) over the kernel's own launch duration (HIP events on its stream, one batch at a time): > 1 = not a "
                                  "physical bound (codes are shared by the queries of a workgroup and served from L2 / Infinity Cache).  "
                                  "physical_frac = PMC fabric bytes over the same duration.  The physically HBM-bound regime: configs.c4x",
                     "traffic": traffic, "traffic_note": traffic_note,
                     "physical_frac": (traffic / (iso_launch_ms / 1e3) / 1e9 / HBM_PEAK_GBS) if (traffic and iso_launch_ms > 0) else None,
                     "algorithmic_bytes_per_launch": algo_per_launch,
                     "avg_launch_ms": iso_launch_ms, "launches": iso_launches,
                     # with several batches in flight a scan launch shares the chip with the other batches' kernels: its HIP-event
                     # duration grows although the job gets faster -- the timed region's own launches, for the record:
                     "pipeline_avg_launch_ms": pipe_launch_ms, "pipeline_launches": launches,
                     "pipeline_accounting_frac": pipe_achieved / HBM_PEAK_GBS, "batches_in_flight": P,
                     "binding": scan_binding(cfg_name, scan_name, cand / float(launches) * iso_launches,
                                             cand_items / float(launches) * iso_launches, M, stage_prof["scan_kernel_ms"] / 1e3, iso_launches)},
        "timing": timing,
        "stage_ms_per_step": {k: stage_prof[k] / n_stage for k in ("front_ms", "tables_ms", "scan_ms", "merge_ms", "scan_kernel_ms")},
        "encode": {"value": enc_rate, "unit": "vectors/s", "vectors": len(my_chunks) * chunk_n,
                   "note": "cis_encode_dev on this rank's share of the index build, HIP events around the encode calls (after one untimed warm-up call)",
                   "roofline": {"bound": "mfma", "achieved": enc_rate * enc_flop / 1e12, "peak": 78.6, "unit": "TFLOP/s",
                                "frac": enc_rate * enc_flop / 78.6e12, "flop_per_vector": enc_flop, "dtype": "f64"}},
        "build": {"encode_s": encode_s, "insert_s": insert_s, "total_s": build_s,
                  "insert": "device-side merge of %d codes into the HBM index (cis_index_add_dev%s)"
                            % (len(my_chunks) * chunk_n, " after the RCCL all-to-all" if ctx.use_dist else "")},
    }
    rb = result["roofline"]["binding"]
    result["roofline"]["binding_frac"] = rb.get("binding_frac")
    result["roofline"]["binding_resource"] = rb.get("binding_resource")
    result["roofline"]["frac"] = rb.get("binding_frac")
    result["roofline"]["pmc_source"] = rb.get("pmc_source") or {"file": None, "note": "no PMC summary committed for this configuration"}
    st.exchange_flags = [f for f in (getattr(st, "exchange_flags", None) or []) if f is not None]
    if st.exchange_flags:
        n_over = int(torch.stack([f.reshape(()) for f in st.exchange_flags]).sum().item())
        result["config"]["exchange"] = ("packed all-gather of a fixed %.1f x even share per rank, offsets on the device, no host read per batch; "
                                        "%d of %d steps exceeded the fixed size (those are repeated with the exact size outside a timed loop)"
                                        % (1.5, n_over, len(st.exchange_flags)))
        result["config"]["exchange_overflows"] = n_over
    st.R = R
    return result, st


def oracle_parity(st, t_loop_budget, n_loop_max, t_vec_budget, n_vec_max):
    """Parity spot check of the timed workload against the oracle (numpy restatement of the reference, pinned to golden vectors):
    the sampled rows' codes and the first queries' ranked ids / distances.  Also times the oracle (cpu_baseline).  Rank 0, N = 1."""
    from oracle import lopq_oracle as O
    om = O.OracleModel.from_npz(st.z)
    sx = np.concatenate(st.sample_x)
    sp = np.concatenate(st.sample_pos)
    oc, of = O.compute_codes(om, sx)
    enc_ok = bool((oc == st.coarse_h[sp]).all() and (of == st.fine_h[sp]).all())
    oix = O.OracleCSRIndex(om, st.coarse_h, st.fine_h)
    qh = st.qr.cpu().numpy()
    gi, gd = st.res["ids"].cpu().numpy(), st.res["dists"].cpu().numpy()
    n_loop, t_loop, ok, max_rel = 0, 0.0, True, 0.0
    while t_loop < t_loop_budget and n_loop < n_loop_max:
        tq = time.perf_counter()
        ids, dd, _ = oix.search_loop(qh[n_loop], quota=QUOTA, limit=LIMIT)
        t_loop += time.perf_counter() - tq
        ok = ok and bool((gi[n_loop, :len(ids)] == ids).all())
        max_rel = max(max_rel, float(np.max(np.abs(gd[n_loop, :len(ids)] - dd) / np.maximum(dd, 1e-300))))
        n_loop += 1
    n_vec, t_vec = 0, 0.0
    while t_vec < t_vec_budget and n_vec < n_vec_max:
        tq = time.perf_counter()
        ids, dd, _ = oix.search(qh[n_vec], quota=QUOTA, limit=LIMIT)
        t_vec += time.perf_counter() - tq
        ok = ok and bool((gi[n_vec, :len(ids)] == ids).all())
        max_rel = max(max_rel, float(np.max(np.abs(gd[n_vec, :len(ids)] - dd) / np.maximum(dd, 1e-300)))) if len(ids) else max_rel
        n_vec += 1
    parity = {"queries_checked": max(n_loop, n_vec), "ids_bit_exact": ok, "max_rel_dist_err": max_rel,
              "encode_rows_checked": int(sp.shape[0]), "encode_rows_from_chunks": N_CHUNKS, "encode_codes_bit_exact": enc_ok}
    return parity, dict(om=om, oix=oix, qh=qh, n_loop=n_loop, t_loop=t_loop, n_vec=n_vec, t_vec=t_vec)


def cpu_baseline_leg(ctx, st, orc, with_cnn):
    """The oracle timed on this host's cores on bounded samples of the headline workload."""
    import shutil
    import tempfile
    from oracle import cpu_bench
    from oracle import lopq_oracle as O
    om, oix, qh = orc["om"], orc["oix"], orc["qh"]
    cores = max(1, min(len(os.sched_getaffinity(0)), 64))
    from columbiaimagesearch_amd.extractor.preprocess_pool import cpu_allowance
    allowance = cpu_allowance()  # the container's cgroup CPU time (16 cores on the GPU boxes seen so far, of 256 visible)
    # encode, reference-shaped per-vector loop (lopq/lopq/utils.py:203-218), 1 core
    enc_x = gen_chunk(st.centers, 1, 8192, ctx.device).cpu().numpy()
    n_el, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 4.0 and n_el < len(enc_x):
        O.compute_codes_loop(om, enc_x[n_el:n_el + 16])
        n_el += 16
    enc_loop = n_el / (time.perf_counter() - t0)
    # all cores: one single-threaded worker per core (the reference deploys N single-threaded processes)
    workdir = tempfile.mkdtemp(prefix="cis_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        cpu_bench.export_index(workdir, oix, qh, os.path.join(REPO, "tests", "golden", st.cfg["fixture"] + ".npz"), enc_x)
        srch_all, n_sa, _ = cpu_bench.run_pool("search", workdir, 6.0, cores)
        enc_all, n_ea, _ = cpu_bench.run_pool("encode", workdir, 4.0, cores)
        cnn1_all, n_c1, _ = (None, 0, 0) if not with_cnn else cpu_bench.run_pool("cnn1", workdir, 5.0, cores)
    finally:
        shutil.rmtree(workdir, ignore_errors=True)
    cnn256 = None
    if with_cnn:
        from oracle import cnn_oracle as C
        wts = C.synthetic_weights(0)
        xi = C.synthetic_images(256, seed=2)
        torch.set_num_threads(cores)
        C.forward_torch(xi[:32], wts)
        t0 = time.perf_counter()
        C.forward_torch(xi, wts)
        cnn256 = 256 / (time.perf_counter() - t0)
    cpu = {"value": orc["n_loop"] / orc["t_loop"], "unit": "queries/s", "cores": 1, "kind": "port",
           "sample": "%d queries of the timed workload (quota=%d, limit=%d, %d-vector index) through the oracle's "
                     "reference-shaped per-candidate loop (search.py:166-175)" % (orc["n_loop"], QUOTA, LIMIT, st.N),
           "search_vectorised_1core_qps": orc["n_vec"] / orc["t_vec"],
           "search_vectorised_allcore_qps": srch_all, "allcore_workers": cores, "cgroup_cpu_allowance_cores": allowance,
           "encode_loop_1core_vps": enc_loop, "encode_vectorised_allcore_vps": enc_all,
           "cnn_torch_cpu_batch1_x_cores_ips": cnn1_all, "cnn_torch_cpu_batch256_ips": cnn256,
           "samples": "vectorised search: %d queries on 1 core, %d queries over %d single-threaded workers (6 s); encode: %d "
                      "vectors per-vector loop on 1 core, %d vectors over the workers (4 s); DeepSentibank torch-CPU: %d "
                      "images batch 1 over the workers (5 s), one batch of 256 on %d threads"
                      % (orc["n_vec"], n_sa, cores, n_el, n_ea, n_c1, cores)}
    # the loop restatement against the reference's own wall time on C1 (the reference ran in the build
    # container when the fixture was made: tests/golden/c1.npz:ref_encode_vec_per_s; this host is a different machine, so the
    # ratio is a note, not the +-20 % validation -- that one is `tools/validate_loop_restatement.py`, run in the build container)
    try:
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import golden_inputs as gi
        z1 = np.load(os.path.join(REPO, "tests", "golden", "c1.npz"))
        om1 = O.OracleModel.from_npz(z1)
        x1 = gi.c1_inputs()[0][:4096]
        if "ref_encode_vec_per_s" in z1:
            n1, t1s = 0, time.perf_counter()
            while time.perf_counter() - t1s < 3.0 and n1 < len(x1):
                O.compute_codes_loop(om1, x1[n1:n1 + 64])
                n1 += 64
            loop1 = n1 / (time.perf_counter() - t1s)
            cpu["c1_encode_loop_vps_this_host"] = loop1
            cpu["c1_encode_reference_vps_build_container"] = float(z1["ref_encode_vec_per_s"])
            cpu["c1_loop_over_reference"] = loop1 / float(z1["ref_encode_vec_per_s"])
            vp = os.path.join(REPO, "profiles", "loop_restatement_validation.json")
            if os.path.exists(vp):
                cpu["c1_like_for_like"] = json.load(open(vp))
    except Exception as e:  # a timing note must never cost the bench line
        cpu["c1_loop_over_reference_error"] = repr(e)
    return cpu


class _RestrictedCells(object):
    """The rows of a few cells of a cell-contiguous array, addressed by ABSOLUTE positions of the whole layout (what
    OracleCSRIndex.search slices and gathers): the oracle's search then runs on an index of 200 M rows of which only the
    visited cells were brought to the host."""

    def __init__(self, starts, blocks):
        self.starts = np.asarray(starts, dtype=np.int64)
        self.blocks = blocks
        self.ends = self.starts + np.asarray([len(b) for b in blocks], dtype=np.int64)

    def __getitem__(self, key):
        if isinstance(key, slice):
            k = int(np.searchsorted(self.starts, key.start, side="right")) - 1
            assert k >= 0 and key.stop <= self.ends[k], "the oracle asked for a cell that was not brought to the host"
            return self.blocks[k][key.start - self.starts[k]:key.stop - self.starts[k]]
        pos = np.asarray(key, dtype=np.int64)
        k = np.searchsorted(self.starts, pos, side="right") - 1
        out = np.empty(pos.shape, dtype=self.blocks[0].dtype)
        for kk in np.unique(k):
            sel = k == kk
            out[sel] = self.blocks[kk][pos[sel] - self.starts[kk]]
        return out


def c4x_leg(ctx, want_oracle, steps, warmup):
    """The HBM-resident regime: the C4 model over an index whose codes exceed the 256 MB Infinity Cache several
    times -- 200 M x 8-byte codes = 1.6 GB (+ 1.6 GB of ids), encoded from generated vectors like the other configurations.
      batch       8192 queries per step at quota 10000: one ~N/256-candidate cell per query through the batch scan kernel
                  (SURVEY.md 8(d)'s accounting: every (candidate, query) pair = M bytes);
      exhaustive  quota = N ("exhaustive quota=N, full scan, roofline run"; lopq/lopq/search.py:128-133 consumes
                  whole cells until the quota) for ONE query and for a pair: every code byte is needed once per launch, so the
                  algorithmic bytes N x M are also the physical minimum, and `roofline.frac` = N x M / kernel time / 8 TB/s is a
                  PHYSICAL HBM figure (the PMC FETCH_SIZE of the same kernel: profiles/scan_traffic_c4x.json);
    parity: quota-10000 queries against the oracle on the visited cells (brought to the host), the exhaustive query against the
    batch kernels' answer on the same index (and against the oracle's blocked exhaustive search in tests/ and with --c4x-oracle)."""
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    device = ctx.device
    model, z = load_model("c4")
    centers = mixture_centers("descriptor", device)
    N = int(os.environ.get("CIS_BENCH_C4X_N", 200_000_000))
    N -= N % N_CHUNKS
    chunk_n = N // N_CHUNKS
    V, M = model.V, model.M
    searcher = LOPQSearcherHIP(model)
    t_build = time.perf_counter()
    ev, coarse_l, fine_l = [], [], []
    sub = 1 << 20
    model.predict_batch_dev(gen_chunk(centers, 0, min(chunk_n, sub), device))  # untimed: workspaces
    with ctx.wd.phase("c4x: build of the %d-vector index" % N, 600):
        for c in range(N_CHUNKS):
            x = gen_chunk(centers, c, chunk_n, device)
            co_c, fi_c = [], []
            for a in range(0, chunk_n, sub):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                co, fi = model.predict_batch_dev(x[a:a + sub])
                e1.record()
                ev.append((e0, e1))
                co_c.append(co)
                fi_c.append(fi)
            co, fi = torch.cat(co_c), torch.cat(fi_c)
            searcher.add_codes_dev(co, fi, torch.arange(c * chunk_n, (c + 1) * chunk_n, dtype=torch.int64, device=device), dedup=False)
            coarse_l.append(co)
            fine_l.append(fi)
            del x, co_c, fi_c
        torch.cuda.synchronize()
    build_s = time.perf_counter() - t_build
    encode_s = sum(a.elapsed_time(b) for a, b in ev) / 1e3
    coarse_all, fine_all = torch.cat(coarse_l), torch.cat(fine_l)   # 2.4 GB of the 288: kept for the parity legs
    del coarse_l, fine_l
    x0 = gen_chunk(centers, 0, min(chunk_n, 1 << 20), device)
    qb = [make_queries(x0, b, NQ, device) for b in range(4)]
    del x0

    def timed(fn, min_reps, min_s, max_reps=400):
        """median / min / max of the HIP-event time of fn() over repetitions (>= min_reps, until min_s seconds are timed)"""
        ts, tot = [], 0.0
        while len(ts) < max_reps and (len(ts) < min_reps or tot < min_s):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            fn(len(ts))
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 1e3)
            tot += ts[-1]
        s_ = sorted(ts)
        return s_[len(s_) // 2], s_[0], s_[-1], len(ts)

    from columbiaimagesearch_amd import _lib as _cl
    out = {"metric": "queries/sec on a 200M-vector index (HBM-resident regime: 1.6 GB of codes against 256 MB of Infinity Cache)",
           "config": {"workload": "c4x: %d x 128-d float64 unit vectors (descriptor-like anisotropic mixture), LOPQModelPCA 128 -> 128 V=16 M=8 "
                                  "(tests/golden/c4.npz): %.2f GB of codes + %.2f GB of ids in HBM" % (N, N * M / 1e9, N * 8 / 1e9),
                      "name": "c4x", "index_vectors": N, "limit": LIMIT},
           "build": {"total_s": build_s, "encode_s": encode_s, "encode_vectors_per_s": N / encode_s}}
    # ---- batch: 8192 queries per step, quota 10000 ---------------------------------------------------------------------------------
    with ctx.wd.phase("c4x: batch steps", 600):
        for b in range(max(1, warmup)):
            searcher.search_batch_dev(qb[b % 4], quota=QUOTA, limit=LIMIT)
        for q in qb:
            searcher.search_batch_dev(q, quota=QUOTA, limit=LIMIT)   # workspaces at their largest
        torch.cuda.synchronize()
        searcher.set_profiling(True, scan_only=True)
        searcher.read_profile()
        a0 = _cl.alloc_stats()
        cand_box = [0]

        def batch_steps(rep):
            for b in range(steps):
                searcher.search_batch_dev(qb[(rep * steps + b) % 4], quota=QUOTA, limit=LIMIT)
                cand_box[0] += searcher.last_stats()["candidates"]
        med, lo, hi, reps = timed(batch_steps, 3, 0.5)
        a1 = _cl.alloc_stats()
        prof = searcher.read_profile()
        searcher.set_profiling(False)
        scan_name = searcher.last_stats()["scan_kernel"]
    launches = max(prof["scan_launches"], 1)
    launch_ms = prof["scan_kernel_ms"] / launches
    algo = cand_box[0] * M / launches
    out.update({"value": NQ * steps / med, "unit": "queries/s", "ms_per_step": med / steps * 1e3, "steps": steps,
                "timing": {"repetitions": reps, "ms_per_step": {"median": med / steps * 1e3, "min": lo / steps * 1e3, "max": hi / steps * 1e3},
                           "workspace_allocations_in_timed_region": a1[0] - a0[0]}})
    out["config"].update({"queries_per_step": NQ, "quota": QUOTA, "candidates_per_query": cand_box[0] / float(NQ * steps * reps)})
    out["roofline"] = {"bound": "hbm", "kernel": scan_name, "achieved": algo / (launch_ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": algo / (launch_ms / 1e3) / 1e9 / HBM_PEAK_GBS, "accounting_frac": algo / (launch_ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                       "algorithmic_bytes_per_launch": algo, "avg_launch_ms": launch_ms, "launches": launches, "traffic": None,
                       "frac_note": "SURVEY.md 8(d) accounting of the batch leg (every (candidate, query) pair = M bytes; ~32 queries share a cell's "
                                    "codes, four per workgroup): not a physical bound -- the physical figure is `exhaustive.roofline`"}
    # ---- exhaustive: quota = N, one / two / four / eight queries (the HBM-streaming route, csrc/lopq_stream.hip) --------------------
    tp = os.path.join(REPO, "profiles", "scan_traffic_c4x.json")
    pmc = None
    if os.path.exists(tp):
        try:
            pmc = json.load(open(tp))
        except Exception:
            pmc = None
    exh = {}
    with ctx.wd.phase("c4x: exhaustive queries", 600):
        for nq in (1, 2, 4, 8):   # one query, a pair, four per slot (one launch each), eight = two launches of four
            q = qb[0][:nq].contiguous()
            for _ in range(3):
                searcher.search_batch_dev(q, quota=N, limit=LIMIT)
            torch.cuda.synchronize()
            searcher.set_profiling(True, scan_only=True)
            searcher.read_profile()
            med, lo, hi, reps = timed(lambda rep: searcher.search_batch_dev(q, quota=N, limit=LIMIT), 5, 0.25)
            prof = searcher.read_profile()
            searcher.set_profiling(False)
            kname = searcher.last_stats()["scan_kernel"]
            k_ms = prof["scan_kernel_ms"] / max(prof["scan_launches"], 1)
            passes = -(-nq // 4)  # slots hold up to four queries: eight queries = two passes over the codes inside one launch
            phys = float(N) * M * passes   # every code byte once per pass: the algorithmic bytes ARE the physical minimum here
            r = {"bound": "hbm", "kernel": kname, "achieved": phys / (k_ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": phys / (k_ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                 "accounting_frac": nq * float(N) * M / (k_ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                 "queries_per_pass": min(nq, 4), "passes_per_launch": passes,
                 "algorithmic_bytes_per_launch": phys, "avg_launch_ms": k_ms, "launches": prof["scan_launches"],
                 "traffic": None,
                 "frac_note": "PHYSICAL: N x M code bytes (each needed once per launch, 1.6 GB >> L2 + Infinity Cache) / the kernel's own duration "
                              "(HIP events around the launch) / 8 TB/s; accounting_frac counts them once per query of the launch"}
            if pmc and str(nq) in pmc.get("per_nq", {}) and "hbm_bytes_per_launch" in pmc["per_nq"][str(nq)]:
                r["pmc_source"] = pmc_source(tp, dict(pmc, kernel=pmc["per_nq"][str(nq)].get("kernel")))
                r["traffic"] = pmc["per_nq"][str(nq)]["hbm_bytes_per_launch"]
                r["traffic_frac"] = r["traffic"] / (k_ms / 1e3) / 1e9 / HBM_PEAK_GBS
                r["traffic_note"] = "profiles/scan_traffic_c4x.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this kernel in their own passes (gfx950 correction applied)"
            exh["nq%d" % nq] = {"queries": nq, "quota": N, "ms_per_batch": {"median": med * 1e3, "min": lo * 1e3, "max": hi * 1e3},
                                "repetitions": reps, "queries_per_s": nq / med, "roofline": r,
                                "stream_counters": dict(zip(("served", "handed_back"), searcher.stream_counters()))}
        # the same single query at the API's default quota: one ~N/256-candidate cell
        q1 = qb[0][:1].contiguous()
        for _ in range(3):
            searcher.search_batch_dev(q1, quota=QUOTA, limit=LIMIT)
        med, lo, hi, reps = timed(lambda rep: searcher.search_batch_dev(q1, quota=QUOTA, limit=LIMIT), 5, 0.1)
        exh["single_query_quota_10000"] = {"ms": {"median": med * 1e3, "min": lo * 1e3, "max": hi * 1e3}, "kernel": searcher.last_stats()["scan_kernel"],
                                           "candidates": searcher.last_stats()["candidates"]}
    out["exhaustive"] = exh
    # ---- parity ---------------------------------------------------------------------------------------------------------------------
    par = {}
    with ctx.wd.phase("c4x: parity", 600):
        # (1) the exhaustive answer of the streaming route == the batch kernels' answer on the same index (k_adc_scan4 forced)
        q1 = qb[0][:2].contiguous()
        r_stream = searcher.search_batch_dev(q1, quota=N, limit=LIMIT)
        searcher.set_scan_mode(mode=5)
        r_batch = searcher.search_batch_dev(q1, quota=N, limit=LIMIT)
        other = searcher.last_stats()["scan_kernel"]
        searcher.set_scan_mode(mode=0)
        torch.cuda.synchronize()
        par["exhaustive_stream_equals_%s" % other] = bool(torch.equal(r_stream["ids"], r_batch["ids"]) and
                                                          torch.equal(r_stream["dists"].view(torch.int64), r_batch["dists"].view(torch.int64)) and
                                                          torch.equal(r_stream["visited"], r_batch["visited"]))
        if want_oracle:
            from oracle import lopq_oracle as O
            om = O.OracleModel.from_npz(z)
            # (2) quota-10000 queries against the oracle: the cells the oracle's own multisequence visits are brought to the host (rows
            # of a cell in insertion order = ascending id), the cell offsets are those of the whole index
            cell = coarse_all[:, 0].to(torch.int64).bitwise_and_(0xFFFF) * V + coarse_all[:, 1].to(torch.int64).bitwise_and_(0xFFFF)
            counts = torch.bincount(cell, minlength=V * V).cpu().numpy()
            offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
            nchk = 8
            qh = qb[0][:nchk].cpu().numpy()
            need = set()
            for qi in range(nchk):
                got = 0
                for _, (c0, c1) in O.multisequence(om, O.apply_pca(om, qh[qi]) if om.has_pca else qh[qi]):
                    cid = int(c0) * V + int(c1)
                    need.add(cid)
                    got += int(counts[cid])
                    if got >= QUOTA:
                        break
            need = sorted(need)
            blocks_f, blocks_i = [], []
            for cid in need:
                rows = torch.nonzero(cell == cid).reshape(-1)   # ascending row number = insertion order inside the cell
                blocks_f.append(fine_all[rows].cpu().numpy())
                blocks_i.append(rows.cpu().numpy())
            oix = O.OracleCSRIndex.__new__(O.OracleCSRIndex)
            oix.model, oix.offsets = om, offsets
            oix.fine = _RestrictedCells(offsets[need], blocks_f)
            oix.ids = _RestrictedCells(offsets[need], blocks_i)
            res_b = searcher.search_batch_dev(qb[0], quota=QUOTA, limit=LIMIT)      # 8192-query batch: k_adc_scan4
            res_s = [searcher.search_batch_dev(qb[0][qi:qi + 1].contiguous(), quota=QUOTA, limit=LIMIT) for qi in range(nchk)]  # one at a time
            torch.cuda.synchronize()
            ok_b = ok_s = True
            worst = 0.0
            for qi in range(nchk):
                ids, dd, vis = oix.search(qh[qi], quota=QUOTA, limit=LIMIT)
                gb = res_b["ids"][qi].cpu().numpy()
                gs = res_s[qi]["ids"][0].cpu().numpy()
                ok_b = ok_b and bool((gb[:len(ids)] == ids).all()) and int(res_b["visited"][qi]) == vis
                ok_s = ok_s and bool((gs[:len(ids)] == ids).all()) and int(res_s[qi]["visited"][0]) == vis
                worst = max(worst, float(np.max(np.abs(res_b["dists"][qi].cpu().numpy()[:len(ids)] - dd) / np.maximum(dd, 1e-300))))
            par.update({"oracle_queries_checked": nchk, "oracle_cells_on_host": len(need), "batch_ids_bit_exact": ok_b,
                        "single_query_ids_bit_exact": ok_s, "max_rel_dist_err": worst})
            if ctx.c4x_oracle_exhaustive:
                # (3) the exhaustive query against the oracle's blocked exhaustive search over all rows (tens of seconds of numpy)
                ch, fh = coarse_all.cpu().numpy().view(np.uint16), fine_all.cpu().numpy()
                t0 = time.perf_counter()
                ids, dd, vis = O.search_exhaustive_blocked(om, ch, fh, qh[0], LIMIT)
                par["exhaustive_oracle_s"] = time.perf_counter() - t0
                gi = r_stream["ids"][0].cpu().numpy()
                par["exhaustive_ids_bit_exact_vs_oracle"] = bool((gi[:len(ids)] == ids).all()) and int(r_stream["visited"][0]) == vis
                par["exhaustive_max_rel_dist_err"] = float(np.max(np.abs(r_stream["dists"][0].cpu().numpy()[:len(ids)] - dd) / np.maximum(dd, 1e-300)))
                del ch, fh
    out["parity"] = par
    out["parity_green"] = all(v for k, v in par.items() if isinstance(v, bool)) and par.get("max_rel_dist_err", 0.0) < 1e-9
    del coarse_all, fine_all
    searcher.close()
    torch.cuda.empty_cache()
    return out


def pcie_leg(st, resident_qps):
    """The same step through the host-pointer entry points (PCIe in and out), never `value`.  Two forms: the reference's call pattern
    (one blocking call per batch, pageable numpy arrays) and the deployment form of round 5 -- queries and results in pinned memory
    (cis_host_alloc), cis_index_search_async on three handles (the index and two views) so that one batch's copies overlap the
    others' searches."""
    from columbiaimagesearch_amd import _lib as L
    qh_all = st.qbatches[0].cpu().numpy()
    st.searcher.search_batch(qh_all, quota=QUOTA, limit=LIMIT)
    tp = time.perf_counter()
    reps = 5
    for _ in range(reps):
        st.searcher.search_batch(qh_all, quota=QUOTA, limit=LIMIT)
    dtp = (time.perf_counter() - tp) / reps
    out = {"pageable_blocking": {"value": NQ / dtp, "unit": "queries/s", "ms_per_step": dtp * 1e3,
                                 "note": "cis_index_search: one blocking call per batch, pageable numpy arrays in and out"}}
    n_lanes = max(2, int(os.environ.get("CIS_BENCH_PCIE_LANES", 3)))
    lanes = [st.searcher] + [st.searcher.view() for _ in range(n_lanes - 1)]
    try:
        qpin, outs = [], []
        for i in range(len(lanes)):
            q = L.pinned_empty(qh_all.shape, qh_all.dtype)
            q[...] = st.qbatches[i % len(st.qbatches)].cpu().numpy()
            qpin.append(q)
            outs.append({"ids": L.pinned_empty((NQ, LIMIT), np.int64), "dists": L.pinned_empty((NQ, LIMIT), np.float64),
                         "n_found": L.pinned_empty((NQ,), np.int32), "visited": L.pinned_empty((NQ,), np.int32)})
        for i, sv in enumerate(lanes):
            sv.search_batch_async(qpin[i], quota=QUOTA, limit=LIMIT, out=outs[i])
        for sv in lanes:
            sv.search_wait()
        want = st.searcher.search_batch(qpin[0], quota=QUOTA, limit=LIMIT)
        same = bool((want["ids"] == outs[0]["ids"]).all() and (want["n_found"] == outs[0]["n_found"]).all())
        nb = 36
        tp = time.perf_counter()
        for b in range(nb):
            sv = lanes[b % len(lanes)]
            sv.search_wait()     # the lane's previous batch has landed (its buffers are free again)
            sv.search_batch_async(qpin[b % len(lanes)], quota=QUOTA, limit=LIMIT, out=outs[b % len(lanes)])
        for sv in lanes:
            sv.search_wait()
        dta = (time.perf_counter() - tp) / nb
        out.update({"value": NQ / dta, "unit": "queries/s", "ms_per_step": dta * 1e3, "frac_of_resident": (NQ / dta) / resident_qps if resident_qps else None,
                    "bytes_per_step": int(qh_all.nbytes + NQ * LIMIT * 16 + NQ * 8), "results_equal_blocking_call": same,
                    "handles": n_lanes,
                    "note": "cis_index_search_async / _wait on several handles (the index + views), queries and results in pinned host memory "
                            "(cis_host_alloc); all copies on one copy stream (a copy-in beside a copy-out collapses to 11 GB/s on this platform), "
                            "overlapping the other handles' searches"})
    finally:
        for sv in lanes[1:]:
            sv.close()
    return out


def cnn_legs(ctx):
    """Second half of the BASELINE metric: CNN descriptors/s (batch 256, synthetic weights).  Every rank runs the forward on its own
    GPU (replicas: weights replicated, no collective); the legs are bracketed by a barrier, the time is the
    slowest rank's and the value the whole job's."""
    import torch.distributed as dist
    from columbiaimagesearch_amd.featurizer.synthetic import dlib_weights, sentibank_weights  # seeded (trained weights are not in the tree)
    from columbiaimagesearch_amd.featurizer import DLibFaceNet, SentiBankNet
    device, world = ctx.device, ctx.world
    torch.cuda.empty_cache()
    B = 256
    gcn = torch.Generator(device=device)
    gcn.manual_seed(5)

    def time_net(net, xb, ob, reps=8, lanes=1, stream=None):
        """Seconds per forward: `lanes` batches in flight -- the net and lanes - 1 views of it (shared weights, own workspaces), each on
        its own stream with its own input and output -- round-robin; median of five measurements of reps x lanes forwards."""
        handles = [net] + [net.view() for _ in range(lanes - 1)]
        # every lane on a lane stream (pipe_streams: different hardware pipes), none on the default stream
        streams = pipe_streams(device, lanes) if stream is None else [stream]
        xs = [xb] + [xb.clone() for _ in range(lanes - 1)]
        obs = [ob] + [torch.empty_like(ob) for _ in range(lanes - 1)]

        def go(k):
            for i in range(k):
                l = i % lanes
                with torch.cuda.stream(streams[l]):
                    handles[l].forward_dev(xs[l], obs[l])
        torch.cuda.synchronize()  # (the inputs were made on the default stream)
        go(2 * lanes)
        torch.cuda.synchronize()
        dts = []
        for _ in range(5):
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            tc = time.perf_counter()
            go(reps * lanes)
            torch.cuda.synchronize()
            dt_ = (time.perf_counter() - tc) / (reps * lanes)
            if world > 1:
                tt_ = torch.tensor([dt_], device=device if ctx.backend == "nccl" else "cpu", dtype=torch.float64)
                dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                dt_ = float(tt_.item())
            dts.append(dt_)
        same = True
        if lanes > 1:  # a forward gives the same descriptors alone and with others in flight
            ref = obs[0].clone()
            handles[0].forward_dev(xs[0], obs[0])
            torch.cuda.synchronize()
            same = bool(torch.equal(ref, obs[0]))
        for h in handles[1:]:
            h.close()
        return sorted(dts)[len(dts) // 2], same

    def roof(mac, b, dt):
        flop = 2.0 * mac * b
        return {"bound": "mfma", "achieved": flop / dt / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": flop / dt / (F32_MFMA_PEAK_TFLOPS * 1e12), "flop_per_image": 2 * mac}

    with ctx.wd.phase("CNN legs (replicas, barrier + all-reduce of the time)", 600):
        LANES = 4  # batches in flight, one per hardware pipe (like the search legs): consecutive forwards are in different layers and fill each other's rounds (three: dlib 0.608, four: 0.620)
        net = SentiBankNet(sentibank_weights(0))
        xb = (torch.randn((B, 3, 227, 227), generator=gcn, device=device) * 50.0).contiguous()
        dt1, _ = time_net(net, xb, torch.empty((B, 4096), device=device))
        dt, same = time_net(net, xb, torch.empty((B, 4096), device=device), lanes=LANES)
        cnn = {"metric": "CNN descriptors/sec (DeepSentibank forward to fc7, batch 256 per GPU, synthetic weights)",
               "value": world * B / dt, "unit": "descriptors/s", "n_gpus": world, "parallelism": "replicas, no collective",
               "ms_per_batch": dt * 1e3, "dtype": "f32", "roofline": roof(SENTIBANK_MAC, B, dt), "batches_in_flight": LANES,
               "same_descriptors_in_flight": same,
               "one_batch_at_a_time": {"value": world * B / dt1, "ms_per_batch": dt1 * 1e3, "frac": roof(SENTIBANK_MAC, B, dt1)["frac"]}}
        net.close()
        del xb
        net = DLibFaceNet(dlib_weights(0))
        xb = (torch.rand((B, 150, 150, 3), generator=gcn, device=device) * 255).contiguous()
        # (a single handle: two half-batch chains, one on the caller's stream and one on a stream of the handle's.  Where the handle's stream
        # lands among the four hardware pipes is not the caller's to choose: the batch is timed from each of the four lane streams -- one
        # of them shares the pipe and loses the overlap; the figure is the median, all four are in the line)
        dt1_by_lane = [time_net(net, xb, torch.empty((B, 128), device=device), stream=ls)[0] for ls in pipe_streams(device, 4)]
        dt1 = sorted(dt1_by_lane)[len(dt1_by_lane) // 2]
        # the same forward at 1024 chips per call: a launch of the 256-chip batch lasts 60-80 us, of which the ramp and the tail
        # of the workgroup rounds are a fifth -- reported next to the BASELINE batch, not instead of it
        B4 = 1024
        xb4 = (torch.rand((B4, 150, 150, 3), generator=gcn, device=device) * 255).contiguous()
        dt4, _ = time_net(net, xb4, torch.empty((B4, 128), device=device), reps=4)
        del xb4
        dt, same = time_net(net, xb, torch.empty((B, 128), device=device), lanes=LANES)  # (views: every handle runs its batch as one chain)
        dlib = {"metric": "CNN descriptors/sec (dlib face ResNet forward, batch 256 aligned chips per GPU, synthetic weights)",
                "value": world * B / dt, "unit": "descriptors/s", "n_gpus": world, "parallelism": "replicas, no collective",
                "ms_per_batch": dt * 1e3, "dtype": "f32", "roofline": roof(DLIB_MAC, B, dt), "batches_in_flight": LANES,
                "same_descriptors_in_flight": same,
                "one_batch_at_a_time": {"value": world * B / dt1, "ms_per_batch": dt1 * 1e3, "frac": roof(DLIB_MAC, B, dt1)["frac"],
                                        "ms_per_batch_by_lane_stream": [round(t * 1e3, 4) for t in dt1_by_lane]}}
        dlib["batch_1024"] = {"value": world * B4 / dt4, "unit": "descriptors/s", "ms_per_batch": dt4 * 1e3,
                              "frac": 2.0 * DLIB_MAC * B4 / dt4 / (F32_MFMA_PEAK_TFLOPS * 1e12)}
        net.close()
        del xb
    torch.cuda.empty_cache()
    return cnn, dlib


def ingest_leg(ctx, st, net_kind, n_ing=12):
    """BASELINE config C5 chain: batched CNN extract (batch 256) -> L2 normalise -> LOPQ encode -> insert with dedup into the
    resident index of `st`.  net_kind "sentibank" (4096-d float32 -> the PCA model: C5's true shapes) or "dlib" (128-d)."""
    from columbiaimagesearch_amd.featurizer.synthetic import dlib_weights, sentibank_weights
    from columbiaimagesearch_amd.featurizer import DLibFaceNet, SentiBankNet
    from columbiaimagesearch_amd.ingest import BatchIngest
    device, searcher = ctx.device, st.searcher
    torch.cuda.empty_cache()
    B = 256
    g5 = torch.Generator(device=device)
    g5.manual_seed(55)
    if net_kind == "sentibank":
        net5, net_name, mac = SentiBankNet(sentibank_weights(0)), "DeepSentibank fc7 (4096-d float32)", SENTIBANK_MAC
        xb5 = (torch.randn((B, 3, 227, 227), generator=g5, device=device) * 50.0).contiguous()
        fdt = None
    else:
        net5, net_name, mac = DLibFaceNet(dlib_weights(0)), "dlib face ResNet (128-d, cast to float64 like the reference's descriptors)", DLIB_MAC
        xb5 = (torch.rand((B, 150, 150, 3), generator=g5, device=device) * 255).contiguous()
        fdt = torch.float64
    ing = BatchIngest(net5, st.model, searcher, feat_dtype=fdt)
    n_before = searcher.get_nb_indexed()
    ing.nb_ingested = 1 << 40  # fresh ids above every stored one: the dedup lookup is answered by the per-cell maximum
    for _ in range(2):
        ing.ingest_batch(xb5)
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    for _ in range(n_ing):
        ing.ingest_batch(xb5)   # forward -> normalise -> encode -> device-side merge incl. refreshed cell statistics
    torch.cuda.synchronize()
    dt5_serial = (time.perf_counter() - t5) / n_ing
    # the same chain with the CNN forwards of the next batches in flight (views of the net on their own streams) while the current
    # batch is encoded and inserted: same ids, same insertion order (BatchIngest.ingest_batches)
    ing.ingest_batches([xb5] * 3, lanes=3)
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    ing.ingest_batches([xb5] * (2 * n_ing), lanes=3)
    torch.cuda.synchronize()
    dt5 = (time.perf_counter() - t5) / (2 * n_ing)
    t5 = time.perf_counter()
    for _ in range(n_ing):
        ing.encode_batch_dev(xb5)
    torch.cuda.synchronize()
    de5 = (time.perf_counter() - t5) / n_ing
    # ids BELOW the stored maximum: the dedup lookup really walks / probes the cell (the best case alone is not the figure)
    ing.nb_ingested = st.N // 2 + 1
    t5 = time.perf_counter()
    n_low = 4
    for _ in range(n_low):
        ing.ingest_batch(xb5)
    torch.cuda.synchronize()
    dl5 = (time.perf_counter() - t5) / n_low
    r5 = searcher.search_batch_dev(st.qbatches[0][:64].contiguous(), quota=QUOTA, limit=LIMIT)  # the grown index still answers
    torch.cuda.synchronize()
    flop = 2.0 * mac * B
    out = {"metric": "descriptors/s end to end: CNN forward (batch 256) -> L2 normalise -> LOPQ encode -> insert with dedup into the "
                     "resident index (BASELINE config C5 chain)",
           "value": B / dt5, "unit": "descriptors/s", "ms_per_batch": dt5 * 1e3, "net": net_name, "cnn_batches_in_flight": 3,
           "one_batch_at_a_time": {"value": B / dt5_serial, "ms_per_batch": dt5_serial * 1e3},
           "model": "LOPQModelPCA %d -> %d V=%d M=%d" % (st.cfg["d_in"], st.model.dim, st.model.V, st.model.M),
           "extract_encode_ms": de5 * 1e3, "insert_ms": (dt5_serial - de5) * 1e3,
           "insert_ms_ids_below_cell_maximum": (dl5 - de5) * 1e3,
           "resident_items_before": int(n_before), "resident_items_after": int(searcher.get_nb_indexed()), "batches": n_ing,
           "insert": "cis_index_add_dev: device-side insert, no host copy of codes",
           "searched_after": int((r5["n_found"] > 0).sum().item()),
           "roofline": {"bound": "mfma", "achieved": flop / dt5 / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flop / dt5 / (F32_MFMA_PEAK_TFLOPS * 1e12),
                        "note": "CNN forward flop over the time of the WHOLE chain (forward + normalise + encode + insert)"}}
    ing.close()
    net5.close()
    del xb5
    torch.cuda.empty_cache()
    return out


def single_query_leg(st):
    """What the reference's callers issue: ONE query per search call (searcher_lopqhbase.py:849-857, 964-970).  Median HIP-event time of
    search_batch_dev on one query of the timed workload (quota 10000, limit 100), 16 different queries in turn, one call at a time."""
    qs = [st.qbatches[0][i:i + 1].contiguous() for i in range(16)]
    for q in qs:
        st.searcher.search_batch_dev(q, quota=QUOTA, limit=LIMIT)
    torch.cuda.synchronize()
    ts = []
    for r in range(48):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        st.searcher.search_batch_dev(qs[r % 16], quota=QUOTA, limit=LIMIT)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    s_ = sorted(ts)
    return {"ms": {"median": s_[len(s_) // 2], "min": s_[0], "max": s_[-1]}, "calls": len(ts), "quota": QUOTA, "limit": LIMIT,
            "candidates": st.searcher.last_stats()["candidates"], "route": st.searcher.last_stats()["scan_kernel"] or "all-candidates select"}


def release_state(st):
    try:
        for sv, _ in (getattr(st, "lanes", None) or [])[1:]:
            sv.close()
        st.searcher.close()
    except Exception:
        pass
    for k in list(vars(st)):
        setattr(st, k, None)
    torch.cuda.empty_cache()


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _r(x, nd=3):
    if isinstance(x, float):
        return float("%.*g" % (nd + 2, x))
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def _roof_c(r):
    """The roofline object in its compact form: the contract's fields + the two extra fractions."""
    if not isinstance(r, dict):
        return None
    out = _pick(r, "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "physical_frac", "traffic_frac", "accounting_frac",
                "binding_frac", "binding_resource", "avg_launch_ms", "algorithmic_bytes_per_launch")
    if isinstance(r.get("pmc_source"), dict):
        out["pmc_source"] = _pick(r["pmc_source"], "file", "commit", "sources_match_this_build")   # (+ kernel symbol in #detail)
        if out["pmc_source"].get("commit"):
            out["pmc_source"]["commit"] = out["pmc_source"]["commit"][:10]
    out.setdefault("traffic", None)
    b = r.get("binding") or {}
    m = b.get("measured") or {}
    for k in ("valu_busy_frac", "lds_busy_frac", "lds_conflict_ratio"):
        if k in m:
            out[k] = m[k]
    return out


XGMI_LINK_GBS = 153.0   # one xGMI link of an MI355X (MI355X_MICROARCH.md / the task statement: 7 links x ~153 GB/s per GPU, point to point)
COLLECTIVE_LATENCY_US = 25.0   # what a small RCCL collective costs whatever its size: 21-30 us measured at world 1 on one MI355X (`--collectives-only`, round 6); a node's own figure is in `collectives`


def exchange_model(world, nq_total, nq_home, L, row_bytes):
    """Bytes the two protocols move per step and rank, and what they should cost on xGMI: every peer has its own link, so a collective
    costs (the largest per-peer share) / 153 GB/s + a fixed latency.  The sizes are those of columbiaimagesearch_amd/distributed.py:
      all-gather protocol  counts [nq] int32 + a FIXED exchange_stride(nq, L, world) x 32-byte records per rank, gathered by every rank;
      routed protocol      row counts [world] int32, the fixed query blocks [world, cap, row] (cap = route_capacity), an overflow word
                           (all-reduce), the ranked lists back (~1.3 owners per query x L x 32 bytes, by the rows actually sent)."""
    from columbiaimagesearch_amd.distributed import exchange_stride, route_capacity
    link = XGMI_LINK_GBS * 1e9
    stride = exchange_stride(nq_total, L, world)
    ag = {"counts_bytes_per_rank": nq_total * 4, "payload_bytes_per_rank": stride * 32,
          "received_bytes_per_rank": (world - 1) * (stride * 32 + nq_total * 4)}
    ag["projected_us"] = 2 * COLLECTIVE_LATENCY_US + (ag["counts_bytes_per_rank"] + ag["payload_bytes_per_rank"]) / link * 1e6
    cap = route_capacity(nq_home, row_bytes // 4, world)
    hits_back = int(1.3 * nq_home * L * 32)
    rt = {"query_block_bytes_per_peer": cap * row_bytes, "query_bytes_sent_per_rank": (world - 1) * cap * row_bytes,
          "hits_back_bytes_per_rank": hits_back, "hits_back_bytes_per_peer": hits_back // max(world, 1), "small_collectives": 2}
    rt["projected_us"] = 4 * COLLECTIVE_LATENCY_US + (rt["query_block_bytes_per_peer"] + rt["hits_back_bytes_per_peer"]) / link * 1e6
    return {"link_GBs": XGMI_LINK_GBS, "assumed_latency_us_per_collective": COLLECTIVE_LATENCY_US,
            "allgather": dict(ag, exchange_bytes_per_step=ag["counts_bytes_per_rank"] + ag["payload_bytes_per_rank"]),
            "routed": dict(rt, exchange_bytes_per_step=rt["query_bytes_sent_per_rank"] + rt["hits_back_bytes_per_rank"] + 8 * world)}


def collectives_leg(ctx, L=None, row_bytes=1024):
    """Bare collectives of the protocols' REAL payload sizes, before anything else touches the GPUs: a slow or hanging collective is
    named here (every one runs in its own watchdog phase) instead of inside a search step.  Every rank; returns rank 0's view."""
    import torch.distributed as dist
    from columbiaimagesearch_amd.distributed import exchange_stride, route_capacity
    device, rank, world, wd = ctx.device, ctx.rank, ctx.world, ctx.wd
    L = LIMIT if L is None else L
    dev = device if ctx.backend == "nccl" else "cpu"
    out = {"backend": ctx.backend, "world": world, "sizes": {}}

    def timed(key, name, fn, bytes_per_peer):
        with wd.phase("collectives-only: %s" % name, 120):
            for _ in range(3):
                fn()
            if dev != "cpu":
                torch.cuda.synchronize()
            ts = []
            for _ in range(20):
                dist.barrier()
                if dev != "cpu":
                    torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                if dev != "cpu":
                    torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            tt = torch.tensor([sorted(ts)[len(ts) // 2]], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            us = float(tt.item()) * 1e6
        out["sizes"][key] = {"what": name, "bytes_per_peer": int(bytes_per_peer), "median_us_max_over_ranks": us,
                              "GBs_per_link": (bytes_per_peer / (us * 1e-6) / 1e9) if us > 0 else None,
                              "projected_us": COLLECTIVE_LATENCY_US + bytes_per_peer / (XGMI_LINK_GBS * 1e9) * 1e6}

    for label, nq_total, nq_home in (("strong", NQ, max(NQ // world, 1)), ("weak", NQ * world, NQ)):
        stride = exchange_stride(nq_total if label == "strong" else NQ, L, world)
        if label == "strong":
            cnt = torch.zeros(NQ, dtype=torch.int32, device=dev)
            cnt_all = torch.empty(world * NQ, dtype=torch.int32, device=dev)
            timed("ag_counts", "all_gather counts [%d] int32" % NQ, lambda: dist.all_gather_into_tensor(cnt_all, cnt), NQ * 4)
            pay = torch.zeros(stride * 32, dtype=torch.uint8, device=dev)
            pay_all = torch.empty(world * stride * 32, dtype=torch.uint8, device=dev)
            timed("ag_payload", "all_gather payload %d x 32 B (all-gather protocol, %d queries per step)" % (stride, NQ),
                  lambda: dist.all_gather_into_tensor(pay_all, pay), stride * 32)
            small = torch.zeros(world, dtype=torch.int32, device=dev)
            small_o = torch.empty_like(small)
            timed("a2a_counts", "all_to_all row counts [%d] int32" % world, lambda: dist.all_to_all_single(small_o, small), 4)
            one = torch.zeros(1, dtype=torch.int32, device=dev)
            timed("allreduce_word", "all_reduce overflow word", lambda: dist.all_reduce(one, op=dist.ReduceOp.MAX), 4)
        cap = route_capacity(nq_home, row_bytes // 4, world)
        sq = torch.zeros(world * cap * row_bytes, dtype=torch.uint8, device=dev)
        rq = torch.empty_like(sq)
        timed("a2a_queries_" + label, "all_to_all query blocks [%d, %d, %d B] (routed, %s: %d home queries per rank)" % (world, cap, row_bytes, label, nq_home),
              lambda: dist.all_to_all_single(rq, sq), cap * row_bytes)
        per_peer = int(1.3 * nq_home * L * 32) // world // 32 * 32
        sh = torch.zeros(world * per_peer, dtype=torch.uint8, device=dev)
        rh = torch.empty_like(sh)
        timed("a2a_lists_" + label, "all_to_all ranked lists back %d B per peer (routed, %s)" % (per_peer, label), lambda: dist.all_to_all_single(rh, sh), per_peer)
        del sq, rq, sh, rh
    return out


def routed_leg(ctx, st, steps, warmup, strong=False):
    """N > 1, cells only (S = N): the ROUTED protocol (columbiaimagesearch_amd/distributed.py: RoutedSearcher) on the index of the headline
    leg.  A step = one batch of world x NQ queries over the job: every rank is the HOME of NQ of them (its own batch), finds the owners
    of the cells they visit, sends each query to those owners only (one all-to-all), answers what it receives and merges its home
    queries' lists.  Per-GPU work is fixed as N grows (NQ home queries, ~NQ received): weak scaling of the QUERY load over ONE 10M
    index sharded by coarse cell.  Before the timed region the protocol is compared with the all-gather protocol on one batch."""
    import torch.distributed as dist
    from columbiaimagesearch_amd.distributed import RoutedSearcher, home_slice
    device, rank, world, wd = ctx.device, ctx.rank, ctx.world, ctx.wd
    sh = st.sharded.row
    rt = RoutedSearcher(sh)
    qbs = st.qbatches
    tdev = device if ctx.backend == "nccl" else "cpu"

    def agree(flag):  # every rank takes the same decision
        t = torch.tensor([1 if flag else 0], device=tdev, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())
    with wd.phase("routed: equality with the all-gather protocol on one batch", 300):
        ref = sh.search_batch_dev(qbs[0], quota=QUOTA, limit=LIMIT)      # every rank: the same NQ queries
        lo, hi = home_slice(NQ, rank, world)
        got = rt.search_batch_dev(qbs[0][lo:hi].contiguous(), quota=QUOTA, limit=LIMIT, nq_total=NQ)
        torch.cuda.synchronize()
        same = bool(torch.equal(ref["ids"][lo:hi], got["ids"]) and torch.equal(ref["n_found"][lo:hi], got["n_found"]) and
                    torch.equal(ref["visited"][lo:hi], got["visited"]))
        dr, dg = ref["dists"][lo:hi], got["dists"]
        same = same and bool(torch.equal(dr[~torch.isnan(dr)], dg[~torch.isnan(dg)]))
        same = agree(same)

    nq_job = NQ if strong else NQ * world   # queries per step over the job
    lo_s, hi_s = home_slice(NQ, rank, world)
    strong_slices = [qb[lo_s:hi_s].contiguous() for qb in qbs] if strong else None

    def home(b):  # this rank's home batch of step b: weak = a whole batch of its own, strong = its slice of the job's ONE batch of NQ
        return strong_slices[b % len(qbs)] if strong else qbs[(b * world + rank) % len(qbs)]

    def run(first, n):
        h = rt.search_begin(home(first), quota=QUOTA, limit=LIMIT, nq_total=nq_job)
        for b in range(1, n):
            h2 = rt.search_begin(home(first + b), quota=QUOTA, limit=LIMIT, nq_total=nq_job)
            rt.search_end(h)
            h = h2
        rt.search_end(h)
    with wd.phase("routed: warm-up (every lane, every batch)", 300):
        run(0, max(warmup, 3 * len(qbs)))
        torch.cuda.synchronize()
    walls, reps = [], 0
    min_reps = int(os.environ.get("CIS_BENCH_MIN_REPS", 5))
    min_timed_s = float(os.environ.get("CIS_BENCH_MIN_TIMED_S", 0.5))
    fb0 = rt.fallbacks
    with wd.phase("routed: timed steps", 900):
        while True:
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(reps * steps, steps)
            torch.cuda.synchronize()
            dist.barrier()
            el = time.perf_counter() - t0
            tt = torch.tensor([el], device=tdev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            walls.append(float(tt.item()))
            reps += 1
            if reps >= 400 or (reps >= min_reps and sum(walls) >= min_timed_s):
                break
    med = sorted(walls)[len(walls) // 2]
    return {"metric": "queries/sec @ recall@10 on 10M LOPQ index",
            "value": nq_job * steps / med, "unit": "queries/s", "ms_per_step": med / steps * 1e3, "scaling": "strong" if strong else "weak",
            "equals_allgather_protocol": same, "fallbacks_in_timed_region": rt.fallbacks - fb0,
            "timing": {"repetitions": reps, "steps_per_repetition": steps, "timed_s": round(sum(walls), 4),
                       "ms_per_step": {"median": med / steps * 1e3, "min": min(walls) / steps * 1e3, "max": max(walls) / steps * 1e3}},
            "config": {"name": st.cfg_name, "index_vectors": st.N, "queries_per_step": nq_job, "home_queries_per_gpu_and_step": nq_job // world,
                       "quota": QUOTA, "limit": LIMIT, "query_groups": 1, "cell_shards": world, "parallelism": "cells 1x%d, routed" % world,
                       "batches_in_flight": 2,
                       "workload": "ONE copy of the %d-vector index sharded by coarse cell over %d GPUs; a step = %d queries over the job, every GPU "
                                   "the home of %d: owners of the visited cells found at home, ONE all-to-all carries a query to those owners only, "
                                   "ranked lists return (all-to-all) and are merged at home (distributed.RoutedSearcher)" % (st.N, world, nq_job, nq_job // world)}}


def compact_line(line):
    """The driver keeps the last ~8 KB of stdout: the FINAL line is a compact form (<= 4 KB) that still carries the contract's fields,
    `roofline`, `cpu_baseline` and value + roofline fraction of every leg; the full detail object is printed before it."""
    c = _pick(line, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data",
              "recall_at_10")
    c["vs_baseline"] = line.get("vs_baseline")
    cfg = line.get("config") or {}
    c["config"] = _pick(cfg, "name", "index_vectors", "queries_per_step", "quota", "limit", "parallelism", "batches_in_flight", "candidates_per_query")
    c["config"]["workload"] = (cfg.get("workload") or "")[:48]
    c["roofline"] = _roof_c(line.get("roofline"))
    t = line.get("timing") or {}
    c["timing"] = _pick(t, "repetitions", "wall_over_events", "workspace_allocations_in_timed_region", "suspect")
    if "ms_per_step" in t:
        c["timing"]["min_max_ms"] = [t["ms_per_step"]["min"], t["ms_per_step"]["max"]]
    if line.get("stage_ms_per_step"):
        c["stage_ms"] = {k[:-3]: v for k, v in line["stage_ms_per_step"].items()}
    e = line.get("encode")
    if e:
        c["encode"] = {"value": e["value"], "unit": e["unit"], "frac": e["roofline"]["frac"]}
    if line.get("pcie_inclusive"):
        c["pcie_inclusive"] = _pick(line["pcie_inclusive"], "value", "ms_per_step", "frac_of_resident")
    if line.get("single_query_quota_10000"):
        c["single_query_quota_10000_ms"] = line["single_query_quota_10000"]["ms"]["median"]
    for k in ("cnn", "dlib"):
        x = line.get(k)
        if x:
            c[k] = {"value": x["value"], "unit": x["unit"], "ms_per_batch": x["ms_per_batch"], "frac": x["roofline"]["frac"], "dtype": x["dtype"]}
            if "one_batch_at_a_time" in x:
                c[k]["in_flight"] = x["batches_in_flight"]
                c[k]["frac_one_at_a_time"] = x["one_batch_at_a_time"]["frac"]
            if "batch_1024" in x:
                c[k]["frac_batch_1024"] = x["batch_1024"]["frac"]
    cb = line.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = _pick(cb, "value", "unit", "cores", "kind", "search_vectorised_allcore_qps", "allcore_workers",
                                  "encode_vectorised_allcore_vps", "cnn_torch_cpu_batch1_x_cores_ips", "cnn_torch_cpu_batch256_ips")
        c["cpu_baseline"]["sample"] = (cb.get("sample") or "")[:24]
    else:
        c["cpu_baseline"] = None
    if line.get("parity"):
        c["parity"] = _pick(line["parity"], "queries_checked", "ids_bit_exact", "encode_codes_bit_exact", "max_rel_dist_err", "batch_ids_bit_exact",
                            "single_query_ids_bit_exact")
    if line.get("ingest"):
        c["ingest"] = _pick(line["ingest"], "value", "unit", "ms_per_batch", "insert_ms")
    if line.get("allgather"):
        a = line["allgather"]
        c["allgather"] = {"value": a["value"], "ms_per_step": a["ms_per_step"], "scaling": a["scaling"], "queries_per_step": a["config"]["queries_per_step"]}
    if line.get("strong"):
        c["strong"] = line["strong"]
    if line.get("exchange"):
        e = line["exchange"]
        c["exchange"] = {"link_GBs": e["link_GBs"], "bytes_per_step": line.get("exchange_bytes_per_step"),
                         "projected_us": {"allgather": round(e["allgather"]["projected_us"], 1), "routed_strong": round(e["routed"]["projected_us"], 1),
                                          "routed_weak": round(e["weak_routed"]["projected_us"], 1)}}
    if isinstance(line.get("collectives"), dict) and line["collectives"].get("sizes"):
        c["collectives_us"] = {k: round(v["median_us_max_over_ranks"], 1) for k, v in line["collectives"]["sizes"].items()}
    if line.get("routed"):
        r = line["routed"]
        c["routed"] = {"error": r["error"][:160]} if "error" in r else _pick(r, "equals_allgather_protocol", "fallbacks_in_timed_region", "value", "ms_per_step")
    g = line.get("grid")
    if g:
        c["grid"] = g if "error" in g else {"value": g["value"], "ms_per_step": g["ms_per_step"], "recall_at_10": g["recall_at_10"],
                                            "parallelism": g["config"]["parallelism"], "frac": g["roofline"]["frac"]}
    ex = line.get("exhaustive")
    if ex:
        c["exhaustive"] = {k: ({"ms": v["ms_per_batch"]["median"], "frac": v["roofline"]["frac"], "kernel_ms": v["roofline"]["avg_launch_ms"]}
                               if "roofline" in v else {"ms": v["ms"]["median"]}) for k, v in ex.items()}
    if line.get("batch_roofline"):
        c["batch_roofline"] = _roof_c(line["batch_roofline"])
    cs = line.get("configs")
    if cs:
        cc = {}
        for name, x in cs.items():
            if "error" in x:
                cc[name] = {"error": x["error"][:120]}
            elif name == "c5":
                cc[name] = {"value": x["value"], "unit": x["unit"], "ms_per_batch": x["ms_per_batch"], "insert_ms": x["insert_ms"], "frac": x["roofline"]["frac"]}
            elif name == "c4x":
                e1 = x["exhaustive"]["nq1"]
                cc[name] = {"value": x["value"], "ms_per_step": x["ms_per_step"], "index_vectors": x["config"]["index_vectors"],
                            "batch_accounting_frac": x["roofline"]["frac"], "batch_launch_ms": x["roofline"]["avg_launch_ms"],
                            "roofline": _roof_c(e1["roofline"]),
                            "exhaustive_ms": {k: v["ms_per_batch"]["median"] for k, v in x["exhaustive"].items() if "ms_per_batch" in v},
                            "exhaustive_queries_per_s": {k: v["queries_per_s"] for k, v in x["exhaustive"].items() if "queries_per_s" in v},
                            "exhaustive_kernel_ms": {k: v["roofline"]["avg_launch_ms"] for k, v in x["exhaustive"].items() if "roofline" in v},
                            "single_query_quota_10000_ms": x["exhaustive"]["single_query_quota_10000"]["ms"]["median"],
                            "parity_green": x.get("parity_green")}
            else:
                r = x["roofline"]
                cc[name] = {"value": x["value"], "ms_per_step": x["ms_per_step"], "recall_at_10": x["recall_at_10"], "kernel": r["kernel"], "frac": r["frac"],
                            "accounting_frac": r.get("accounting_frac"), "physical_frac": r.get("physical_frac"), "launch_ms": r["avg_launch_ms"],
                            "stage_ms": {k[:-3]: v for k, v in x["stage_ms_per_step"].items() if k != "scan_kernel_ms"}, "encode": x["encode"]["value"],
                            "encode_frac": x["encode"]["roofline"]["frac"], "parity_green": x.get("parity_green")}
                if (x.get("timing") or {}).get("suspect"):
                    cc[name]["suspect_timing"] = True
        c["configs"] = cc
    c["detail"] = "#detail line above"
    # the contract's top-level numbers keep five significant digits, everything nested four (the line must stay under 4 KB)
    out = {k: (_r(v, 2) if isinstance(v, (dict, list, tuple)) else _r(v, 3)) for k, v in c.items()}
    for drop in ("ingest", "timing", "stage_ms", "encode", "collectives_us", "grid"):   # never needed so far: the last resort of a line that grew
        if len(json.dumps(out)) <= 4000:
            break
        out.pop(drop, None)
    return out


def emit_lines(line, compact, detail_file=None):
    # the JSON lines are the last thing on stdout: flush what native libraries (the RCCL banner) still hold in C stdio first
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if detail_file:
        try:
            with open(detail_file, "w") as f:
                json.dump(line, f)
        except Exception as e:
            sys.stderr.write("[bench] could not write %s: %r\n" % (detail_file, e))
    print("#detail " + json.dumps(line), flush=True)
    print(json.dumps(compact), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS) + ["c4x"], default=os.environ.get("CIS_BENCH_CONFIG") or None,
                    help="headline workload (default c4 + short c2 / c3 / c5 / c4x sub-runs under `configs`); c4x = the HBM-resident regime alone "
                         "(200 M vectors: batch leg + exhaustive queries)")
    ap.add_argument("--c4x-oracle", action="store_true", help="c4x: also check the exhaustive query against the oracle's blocked exhaustive "
                                                              "search over all 200 M rows (tens of seconds of numpy)")
    ap.add_argument("--no-c4x", action="store_true", help="skip the c4x sub-run of the default line")
    ap.add_argument("--detail-file", default=os.environ.get("CIS_BENCH_DETAIL_FILE"),
                    help="also write the full detail object (what the `#detail` line carries) to this file")
    ap.add_argument("--n", type=int, default=int(os.environ.get("CIS_BENCH_N", 0)), help="index vectors (default: the config's)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="of the INDEX size with --gpus (weak: n vectors per GPU)")
    ap.add_argument("--cell-shards", type=int, default=int(os.environ.get("CIS_BENCH_CELL_SHARDS", 0)),
                    help="S of the second (`grid`) layout at N > 1: R = gpus / S query groups, each one copy of the index sharded by "
                         "cell over S GPUs.  Default: 2 from 4 GPUs on, 1 (whole copies) at 2.  The headline is always S = gpus")
    ap.add_argument("--no-grid", action="store_true", help="N > 1: skip the second layout")
    ap.add_argument("--collectives-only", action="store_true",
                    help="N > 1: time the bare RCCL collectives of the protocols' payload sizes (all-gather counts / payload, all-to-all query blocks / "
                         "ranked lists, the small ones), print them and stop -- the first thing to run on a new node")
    ap.add_argument("--no-routed", action="store_true", help="N > 1: skip the routed protocol (the headline is then the all-gather protocol at 8192 queries per step)")
    ap.add_argument("--pipeline", type=int, default=int(os.environ.get("CIS_BENCH_PIPELINE", 0)),
                    help="N = 1: query batches in flight at once (each through its own view of the index on its own stream; default 4, at N > 1 three partial searches per rank); 1 = one after the other")
    ap.add_argument("--no-configs", action="store_true", help="skip the c2 / c3 / c5 sub-runs of the default line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cnn", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-pointer leg (profiling runs)")
    args = ap.parse_args()
    explicit_config = args.config is not None
    only_c4x = args.config == "c4x"
    cfg_name = "c4" if (args.config is None or only_c4x) else args.config
    cfg = CONFIGS[cfg_name]
    if args.n <= 0:
        args.n = cfg["n"]

    ctx = Ctx()
    rank = ctx.rank = int(os.environ.get("RANK", 0))
    world = ctx.world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if os.environ.get("CIS_BENCH_ONE_DEVICE") == "1":
        local_rank = 0  # functional test of the N > 1 protocol on a 1-GPU box (with CIS_BENCH_BACKEND=gloo: RCCL refuses shared devices)
    backend = ctx.backend = os.environ.get("CIS_BENCH_BACKEND", "nccl")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    device = ctx.device = torch.device("cuda", local_rank)
    wd = ctx.wd = Watchdog(rank)
    # batches in flight: FOUR on one GPU (round 6, same-box A/B of three runs each: C4 21.8-21.9 -> 22.0-22.4 M queries/s, C2 20.6-20.7 ->
    # 20.7-21.0 M; five lose), three partial searches in flight per rank at N > 1 (the exchange runs beside them)
    ctx.pipeline = args.pipeline if args.pipeline > 0 else (4 if world == 1 else 3)
    from columbiaimagesearch_amd import _lib
    _lib.check(_lib.lib().cis_set_device(local_rank))
    # CIS_BENCH_FORCE_DIST=1 runs the sharded code path (process group, all-gather, merge) even with one
    # rank -- a smoke test of the N > 1 path on a 1-GPU box
    ctx.use_dist = world > 1 or os.environ.get("CIS_BENCH_FORCE_DIST") == "1"
    if ctx.use_dist:
        import datetime
        import torch.distributed as dist
        with wd.phase("init_process_group (%s, world %d)" % (backend, world), 300):
            kw = dict(timeout=datetime.timedelta(seconds=300))
            if "RANK" not in os.environ:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29533")
                dist.init_process_group(backend, rank=0, world_size=1, device_id=device if backend == "nccl" else None, **kw)
            else:
                dist.init_process_group(backend, device_id=device if backend == "nccl" else None, **kw)  # nccl == RCCL on ROCm

    collectives = None
    if ctx.use_dist and (world > 1 or args.collectives_only):
        try:
            collectives = collectives_leg(ctx)
        except Exception as e:
            collectives = {"error": repr(e)}
            sys.stderr.write("[bench rank %d] collectives leg failed: %r\n" % (rank, e))
        if rank == 0:
            sys.stderr.write("[bench] collectives (median us, max over ranks): %s\n" % json.dumps(collectives))
    if args.collectives_only:
        if rank == 0:
            line = {"metric": "bare collectives of the search protocols' payloads", "value": None, "unit": "us", "n_gpus": world, "steps": 20, "warmup": 3,
                    "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                    "config": {"workload": "collectives only: the all-gather and all-to-all payload sizes of an %d-query step at limit %d" % (NQ, LIMIT)},
                    "collectives": collectives, "exchange": exchange_model(world, NQ, max(NQ // world, 1), LIMIT, 1024) if world > 1 else None}
            emit_lines(line, line, args.detail_file)
        if ctx.use_dist:
            import torch.distributed as dist
            dist.destroy_process_group()
        return
    ctx.c4x_oracle_exhaustive = bool(args.c4x_oracle)
    solo = rank == 0 and world == 1
    want_oracle = solo and not args.no_cpu_baseline
    if only_c4x:
        if world != 1:
            raise SystemExit("--config c4x runs on one GPU")
        x = c4x_leg(ctx, want_oracle, max(2, min(args.steps, 5)), args.warmup)
        line = {"metric": x["metric"], "value": x["value"], "unit": x["unit"], "n_gpus": 1, "steps": x["steps"], "warmup": args.warmup,
                "ms_per_step": x["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
                "data": "synthetic", "config": x["config"], "roofline": x["exhaustive"]["nq1"]["roofline"], "batch_roofline": x["roofline"],
                "timing": x["timing"], "exhaustive": x["exhaustive"], "parity": x["parity"], "parity_green": x["parity_green"], "build": x["build"],
                "cpu_baseline": None}
        emit_lines(line, compact_line(line), args.detail_file)
        return
    # ---- headline: BASELINE C4's layout -- ONE copy of the index sharded by coarse cell over all N GPUs ---------------------------
    head, st = search_leg(ctx, cfg_name, args.n, world, args.steps, args.warmup, args.scaling, 4096 if want_oracle else 0)

    routed = routed_strong = None
    if world > 1 and st.sharded is not None and st.sharded.row is not None and not args.no_routed:
        try:
            routed = routed_leg(ctx, st, args.steps, args.warmup)
        except Exception as e:  # the all-gather headline stands on its own
            routed = {"error": repr(e)}
            sys.stderr.write("[bench rank %d] routed leg failed: %r\n" % (rank, e))
        try:   # the SAME job as N = 1 (8192 queries per step over all GPUs) through the routed protocol: like with like
            routed_strong = routed_leg(ctx, st, args.steps, args.warmup, strong=True)
        except Exception as e:
            routed_strong = {"error": repr(e)}
            sys.stderr.write("[bench rank %d] routed leg (strong) failed: %r\n" % (rank, e))
    pcie = pcie_leg(st, head["value"]) if solo and not args.no_pcie else None
    single = None
    if solo:
        try:
            single = single_query_leg(st)
        except Exception as e:  # a side leg must never cost the headline line
            sys.stderr.write("[bench] single-query leg failed: %r\n" % (e,))
    cpu = parity = None
    if want_oracle:
        parity, orc = oracle_parity(st, 10.0, 256, 4.0, 1024)
        cpu = cpu_baseline_leg(ctx, st, orc, not args.no_cnn)
        del orc
    ingest = None
    if not args.no_cnn and st.sharded is None:
        try:
            ingest = ingest_leg(ctx, st, "sentibank" if cfg["d_in"] == 4096 else "dlib")
        except Exception as e:  # a side leg must never cost the headline line
            ingest = None
            sys.stderr.write("[bench] ingest leg failed: %r\n" % (e,))
            torch.cuda.empty_cache()
        if ingest is not None and cfg["d_in"] != 4096:
            ingest["note"] = "dlib 128-d descriptors on this config's model; C5 at its true shapes (DeepSentibank 4096-d -> PCA 256, M=16) is configs.c5"
    release_state(st)

    # ---- N > 1: the R x S grid as a second object (never the headline) -----------------------------------------------------------
    grid = None
    if world > 1 and not args.no_grid:
        S2 = args.cell_shards if args.cell_shards > 0 else (2 if world >= 4 and world % 2 == 0 else 1)
        if world % S2 == 0 and S2 != world:
            try:
                g, gst = search_leg(ctx, cfg_name, args.n, S2, args.steps, args.warmup, args.scaling, 0)
                grid = {"metric": "queries/sec over R query groups x S cell shards (every group holds one whole copy of the index and "
                                  "answers its own 8192 queries per step: the query load grows with R)",
                        "value": g["value"], "unit": "queries/s", "ms_per_step": g["ms_per_step"], "scaling": "weak",
                        "recall_at_10": g["recall_at_10"], "config": g["config"], "roofline": g["roofline"],
                        "stage_ms_per_step": g["stage_ms_per_step"], "timing": g["timing"]}
                release_state(gst)
            except Exception as e:  # e.g. dist.new_group unavailable: the headline (S = N) stands on its own
                grid = {"error": repr(e), "layout": "%d x %d" % (world // S2, S2)}
                sys.stderr.write("[bench rank %d] grid layout failed: %r\n" % (rank, e))

    # ---- the other single-GPU BASELINE configurations, short (default line only) -------------------------------------------------
    configs = None
    if solo and not explicit_config and not args.no_configs:
        configs = {}
        sub_steps = max(5, min(args.steps, 30))  # (10 steps behind 2 warm-up steps were mostly the ramp of the three batches in flight: C2 17.9 against 19 M q/s)
        for name in ("c2", "c3"):
            try:
                r, sst = search_leg(ctx, name, CONFIGS[name]["n"], 1, sub_steps, max(2, min(args.warmup, 5)), "strong", 1024 if want_oracle else 0)
                sub = {"value": r["value"], "unit": "queries/s", "ms_per_step": r["ms_per_step"], "steps": sub_steps,
                       "recall_at_10": r["recall_at_10"], "config": r["config"], "roofline": r["roofline"],
                       "stage_ms_per_step": r["stage_ms_per_step"], "encode": r["encode"], "timing": r["timing"]}
                if want_oracle:
                    p, _ = oracle_parity(sst, 2.0, 8, 3.0, 64)
                    sub["parity"] = p
                    sub["parity_green"] = bool(p["ids_bit_exact"] and p["encode_codes_bit_exact"] and p["max_rel_dist_err"] < 1e-9)
                configs[name] = sub
                if name == "c3" and not args.no_cnn:
                    c5 = ingest_leg(ctx, sst, "sentibank", n_ing=8)
                    c5["config"] = {"workload": "C5 chain at its true shapes on one GPU: 256 x 3 x 227 x 227 float32 batches (randn * 50), seeded "
                                                "synthetic DeepSentibank weights -> fc7 4096-d -> L2 normalise -> LOPQModelPCA 4096 -> 256 V=16 M=16 "
                                                "(tests/golden/c3full.npz) -> insert with dedup into the resident %d-vector c3 index" % sst.N}
                    configs["c5"] = c5
                release_state(sst)
            except Exception as e:  # a sub-run must never cost the headline line
                configs[name] = {"error": repr(e)}
                sys.stderr.write("[bench] sub-config %s failed: %r\n" % (name, e))
                torch.cuda.empty_cache()

    if configs is not None and not args.no_c4x:
        try:
            configs["c4x"] = c4x_leg(ctx, want_oracle, 3, 1)
        except Exception as e:  # a sub-run must never cost the headline line
            configs["c4x"] = {"error": repr(e)}
            sys.stderr.write("[bench] sub-config c4x failed: %r\n" % (e,))
            torch.cuda.empty_cache()

    cnn = dlib = None
    if not args.no_cnn:
        try:
            cnn, dlib = cnn_legs(ctx)
        except Exception as e:  # (every rank fails alike or none does: the legs hold no data-dependent branch before their collectives)
            cnn = dlib = None
            sys.stderr.write("[bench rank %d] CNN legs failed: %r\n" % (rank, e))
            torch.cuda.empty_cache()

    if rank == 0:
        line = {
            "metric": "queries/sec @ recall@10 on 10M LOPQ index",
            "value": head["value"],
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "recall_at_10": head["recall_at_10"],
            "config": head["config"],
            "roofline": head["roofline"],
            "timing": head["timing"],
            "stage_ms_per_step": head["stage_ms_per_step"],
            "encode": head["encode"],
            "pcie_inclusive": pcie,
            "single_query_quota_10000": single,
            "cnn": cnn,
            "dlib": dlib,
            "cpu_baseline": cpu,
            "parity": parity,
            "ingest": ingest,
            "build": head["build"],
            "grid": grid,
            "configs": configs,
        }
        if routed is not None:
            # N > 1: the headline is the ROUTED protocol when it ran and reproduced the all-gather protocol's answers -- same index
            # layout (one copy, sharded by coarse cell over all N GPUs), per-GPU work fixed as N grows (weak scaling of the query
            # load); the all-gather protocol at 8192 queries per step over the whole job (strong) stays in the line as `allgather`.
            if "error" not in routed and routed["equals_allgather_protocol"] and routed["fallbacks_in_timed_region"] == 0:
                line["allgather"] = {"value": head["value"], "ms_per_step": head["ms_per_step"], "scaling": args.scaling, "config": head["config"],
                                     "timing": head["timing"], "recall_at_10": head["recall_at_10"]}
                line.update({"value": routed["value"], "ms_per_step": routed["ms_per_step"], "scaling": "weak", "config": routed["config"],
                             "timing": routed["timing"]})
                line["routed"] = {"equals_allgather_protocol": True, "fallbacks_in_timed_region": 0}
                line["roofline"] = dict(line["roofline"], note="k_adc_scan launches of the all-gather leg on the same index (the routed leg runs the same kernels on the queries a rank receives)")
            else:
                line["routed"] = routed
        if world > 1:
            # The SAME job as N = 1 -- 8192 queries per step over all GPUs -- next to the weak headline: N = 1 -> N compares like with like.
            strong = {"queries_per_step": NQ,
                      "allgather": {"value": head["value"], "ms_per_step": head["ms_per_step"]}}
            if routed_strong is not None:
                strong["routed"] = ({"error": routed_strong["error"][:160]} if "error" in routed_strong else
                                    {"value": routed_strong["value"], "ms_per_step": routed_strong["ms_per_step"],
                                     "fallbacks_in_timed_region": routed_strong["fallbacks_in_timed_region"],
                                     "equals_allgather_protocol": routed_strong["equals_allgather_protocol"]})
            line["strong"] = strong
            row_bytes = int(cfg["d_in"]) * (4 if cfg["gen"] == "relu_mixture" else 8)
            ex = exchange_model(world, NQ, max(NQ // world, 1), LIMIT, row_bytes)
            ex["weak_routed"] = exchange_model(world, NQ * world, NQ, LIMIT, row_bytes)["routed"]
            line["exchange"] = ex
            line["exchange_bytes_per_step"] = {"allgather": ex["allgather"]["exchange_bytes_per_step"], "routed_strong": ex["routed"]["exchange_bytes_per_step"],
                                               "routed_weak": ex["weak_routed"]["exchange_bytes_per_step"], "per": "rank and step (sent)"}
        if collectives is not None:
            line["collectives"] = collectives
    if ctx.use_dist:
        import torch.distributed as dist
        with wd.phase("final barrier", 120):
            dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit_lines(line, compact_line(line), args.detail_file)


if __name__ == "__main__":
    main()
