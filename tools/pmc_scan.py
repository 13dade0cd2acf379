#!/usr/bin/env python3
"""Per-launch PMC figures of the scan kernel of one bench configuration, from the raw rocprofv3 counter_collection CSVs of separate
passes (FETCH_SIZE, WRITE_SIZE, the SQ set): the MEDIAN over the kernel's dispatches -- most dispatches of a bench run are full
8192-query batches, the recall batch (1024 queries) and the small parity batches are the minority, so the median is a full launch
whatever the number of repetitions the run made (round 4 divided totals by a hand-counted number of "full-launch equivalents").
usage: pmc_scan.py <tag> <config> <serial bench line json>   (reads gpurun_out/<tag>_<config>_{FETCH_SIZE,WRITE_SIZE,sq}_raw.csv)
writes gpurun_out/scan_traffic_<config>.json and gpurun_out/scan_binding_<config>.json (copy to profiles/: bench.py reads them)."""
import csv, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_stamp import kernel_sources_sha1
tag, cfg, line_path = sys.argv[1], sys.argv[2], sys.argv[3]
CLK, SIMDS, CUS = 2.4e9, 1024, 256


def medians(path, want):
    per = {}
    kern = None
    for row in csv.DictReader(open(path)):
        k = row.get("Kernel_Name", "")
        if want not in k:
            continue
        kern = k.split("(")[0]
        per.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
        per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {c: statistics.median(v.values()) for c, v in per.items()}, {c: len(v) for c, v in per.items()}, kern


line = json.load(open(line_path))
r = line["roofline"]
want = r["kernel"] + "<"          # e.g. "k_adc_scan4<": the main kernel, not its fall-back launch
ms = r["avg_launch_ms"]           # one batch at a time: the launch alone on the chip
f, nf, kern = medians("gpurun_out/%s_%s_FETCH_SIZE_raw.csv" % (tag, cfg), want)
w, nw, _ = medians("gpurun_out/%s_%s_WRITE_SIZE_raw.csv" % (tag, cfg), want)
sq, nsq, _ = medians("gpurun_out/%s_%s_sq_raw.csv" % (tag, cfg), want)
fk, wk = f.get("FETCH_SIZE", 0.0), w.get("WRITE_SIZE", 0.0)
traffic = {"kernel": kern, "config": cfg, "dispatches_in_pass": nf.get("FETCH_SIZE"),
           "source": "gpurun_out/%s_%s_{FETCH_SIZE,WRITE_SIZE}_raw.csv: rocprofv3 --pmc in separate passes (kernel trace only, one batch at a time), "
                     "MEDIAN over the kernel's dispatches of bench.py --config %s" % (tag, cfg, cfg),
           "fetch_size_kb_per_launch": fk, "write_size_kb_per_launch": wk,
           "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read -> doubled (the scan reads 8-16 B/lane: upper-bound style "
                         "estimate); WRITE_SIZE as reported",
           "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0, "write_bytes_per_launch": wk * 1024.0,
           "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"]}
traffic["kernel_sources_sha1"] = kernel_sources_sha1()
json.dump(traffic, open("gpurun_out/scan_traffic_%s.json" % cfg, "w"), indent=1)
cyc = ms * 1e-3 * CLK
b = {"config": cfg, "kernel": kern, "avg_launch_ms": ms, "dispatches_in_pass": nsq.get("SQ_INSTS_VALU"),
     "source": "gpurun_out/%s_%s_sq_raw.csv (rocprofv3 --pmc, own pass, kernel trace only; median over the kernel's dispatches) and the launch time of "
               "the un-instrumented one-batch-at-a-time line of the same run" % (tag, cfg),
     "per_launch": sq}
if "SQ_INSTS_VALU" in sq:
    b["valu_busy_frac"] = sq["SQ_INSTS_VALU"] * 4.0 / (SIMDS * cyc)
if sq.get("SQ_LDS_IDX_ACTIVE"):
    b["lds_busy_frac"] = sq["SQ_LDS_IDX_ACTIVE"] / (CUS * cyc)
    b["lds_conflict_ratio"] = sq.get("SQ_LDS_BANK_CONFLICT", 0.0) / sq["SQ_LDS_IDX_ACTIVE"]
b["hbm_bytes_per_launch"] = traffic["hbm_bytes_per_launch"]
b["hbm_frac"] = traffic["hbm_bytes_per_launch"] / (ms * 1e-3) / 8.0e12
b["kernel_sources_sha1"] = kernel_sources_sha1()
json.dump(b, open("gpurun_out/scan_binding_%s.json" % cfg, "w"), indent=1)
print(json.dumps({"traffic_MB": traffic["hbm_bytes_per_launch"] / 1e6, "valu_busy": b.get("valu_busy_frac"), "lds_busy": b.get("lds_busy_frac"),
                  "conflicts": b.get("lds_conflict_ratio"), "valu_insts": sq.get("SQ_INSTS_VALU"), "launch_ms": ms, "kernel": kern}))
