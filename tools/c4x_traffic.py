#!/usr/bin/env python3
"""HBM / fabric bytes and SQ counters per k_adc_stream launch from the passes of tools/c4x_pmc.sh -> gpurun_out/scan_traffic_c4x.json
(bench.py reads profiles/scan_traffic_c4x.json: configs.c4x.roofline.traffic).
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes of a wide coalesced streaming read (16 B per
lane: exactly this kernel's loads) -> doubled; WRITE_SIZE as reported; both in KB."""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_stamp import kernel_sources_sha1
tag, N = sys.argv[1], int(float(sys.argv[2]))
CLK, SIMDS, CUS = 2.4e9, 1024, 256


def rows(path):
    out = []
    for line in open(path).read().splitlines()[1:]:
        parts = line.rsplit(",", 4)
        if len(parts) == 5 and "k_adc_stream" in parts[0] and "false" in parts[0].replace(" ", "").split("<")[1]:
            out.append((parts[0], parts[2], float(parts[3]), int(float(parts[1])) if parts[1].replace(".", "").isdigit() else 0))
    return out


res = {"config": "c4x", "index_vectors": N, "code_bytes": N * 8, "per_nq": {},
       "source": "tools/c4x_pmc.sh: rocprofv3 --pmc (one counter set per pass, kernel trace only) over tools/stream_pmc_driver.py N nq 8 -- every "
                 "k_adc_stream<M, G, false> dispatch is one exhaustive launch over the whole index",
       "correction": "gfx950: FETCH_SIZE x 2 (16-byte-per-lane streaming loads are tallied at half their size); WRITE_SIZE as reported; KB"}
for nq in (1, 2, 4):
    ent = {}
    try:
        f = rows("gpurun_out/%s_c4x_nq%d_FETCH_SIZE_pmc.csv" % (tag, nq))
        w = rows("gpurun_out/%s_c4x_nq%d_WRITE_SIZE_pmc.csv" % (tag, nq))
        sq = rows("gpurun_out/%s_c4x_nq%d_sq_pmc.csv" % (tag, nq))
    except Exception as e:
        res["per_nq"][str(nq)] = {"error": repr(e)}
        continue
    # pmc_summary.py prints one line per (kernel, counter): total over the dispatches and the dispatch count
    fk = sum(r[2] for r in f if r[1] == "FETCH_SIZE")
    wk = sum(r[2] for r in w if r[1] == "WRITE_SIZE")
    nf = max(sum(r[3] for r in f if r[1] == "FETCH_SIZE"), 1)
    nw = max(sum(r[3] for r in w if r[1] == "WRITE_SIZE"), 1)
    ent["kernel"] = f[0][0] if f else None
    ent["launches_in_pass"] = nf
    ent["fetch_kb_per_launch"] = fk / nf
    ent["write_kb_per_launch"] = wk / nw
    ent["hbm_bytes_per_launch"] = (2.0 * fk / nf + wk / nw) * 1024.0
    ent["hbm_bytes_over_code_bytes"] = ent["hbm_bytes_per_launch"] / (N * 8.0)
    per = {}
    for r in sq:
        per[r[1]] = per.get(r[1], 0.0) + r[2] / max(r[3], 1)
    ent["sq_per_launch"] = per
    try:
        ks = list(csv.DictReader(open("gpurun_out/%s_c4x_nq%d_kernel_stats.csv" % (tag, nq))))
        ms = [float(r["AverageNs"]) / 1e6 for r in ks if "false" in r["Name"]]
        if ms:
            ent["avg_launch_ms_rocprof"] = ms[0]
            cyc = ms[0] * 1e-3 * CLK
            ent["hbm_frac_of_8TBs"] = ent["hbm_bytes_per_launch"] / (ms[0] * 1e-3) / 8.0e12
            if "SQ_INSTS_VALU" in per:
                ent["valu_busy_frac"] = per["SQ_INSTS_VALU"] * 4.0 / (SIMDS * cyc)
            if "SQ_LDS_IDX_ACTIVE" in per:
                ent["lds_busy_frac"] = per["SQ_LDS_IDX_ACTIVE"] / (CUS * cyc)
                if per.get("SQ_LDS_IDX_ACTIVE"):
                    ent["lds_conflict_ratio"] = per.get("SQ_LDS_BANK_CONFLICT", 0.0) / per["SQ_LDS_IDX_ACTIVE"]
    except Exception as e:
        ent["kernel_stats_error"] = repr(e)
    res["per_nq"][str(nq)] = ent
res["kernel_sources_sha1"] = kernel_sources_sha1()
json.dump(res, open("gpurun_out/scan_traffic_c4x.json", "w"), indent=1)
print(json.dumps(res, indent=1))
