#!/usr/bin/env python3
"""Measured binding of the scan kernel from the SQ counter pass of tools/gpu_config_profile.sh (<tag>_<cfg>_sq_pmc.csv) and the
launch time of the un-instrumented bench line of the same run -> gpurun_out/scan_binding_<cfg>.json (copied to profiles/, read by
bench.py as roofline.binding.measured).
usage: scan_binding.py <tag> <config> <full_launch_equivalents>
  valu_busy_frac  = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x launch cycles)   (a wave64 VALU instruction holds its SIMD 4 cycles)
  lds_busy_frac   = SQ_LDS_IDX_ACTIVE / (256 CUs x launch cycles)            (LDS pipe cycles incl. bank-conflict replays)
  lds_conflict_ratio = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  hbm_frac        = PMC traffic of scan_traffic_<cfg>.json / launch time / 8 TB/s (if that file exists)"""
import json, os, sys
tag, cfg, equiv = sys.argv[1], sys.argv[2], float(sys.argv[3])
CLK, SIMDS, CUS = 2.4e9, 1024, 256
tot = {}
kern = None
for line in open("gpurun_out/%s_%s_sq_pmc.csv" % (tag, cfg)).read().splitlines()[1:]:
    parts = line.rsplit(",", 4)
    if len(parts) == 5 and "k_adc_scan" in parts[0]:
        tot[parts[2]] = tot.get(parts[2], 0.0) + float(parts[3])
        if kern is None or "scan4" in parts[0]:
            kern = parts[0]
serial = "gpurun_out/%s_%s_bench_line_serial.json" % (tag, cfg)   # one batch at a time: the launch alone on the chip
line = json.load(open(serial if os.path.exists(serial) else "gpurun_out/%s_%s_bench_line.json" % (tag, cfg)))
ms = line["roofline"].get("isolated_launch_ms") or line["roofline"]["avg_launch_ms"]
cyc = ms * 1e-3 * CLK
per = {k: v / equiv for k, v in tot.items()}
out = {"config": cfg, "kernel": kern, "avg_launch_ms": ms,
       "source": "gpurun_out/%s_%s_sq_pmc.csv (rocprofv3 --pmc, own pass, kernel trace only; all k_adc_scan* dispatches of bench.py --config %s "
                 "--steps 3 --warmup 1 = %.3f full-launch equivalents) and the launch time of the un-instrumented line of the same run" % (tag, cfg, cfg, equiv),
       "per_launch": per}
if "SQ_INSTS_VALU" in per:
    out["valu_busy_frac"] = per["SQ_INSTS_VALU"] * 4.0 / (SIMDS * cyc)
if "SQ_LDS_IDX_ACTIVE" in per:
    out["lds_busy_frac"] = per["SQ_LDS_IDX_ACTIVE"] / (CUS * cyc)
    if "SQ_LDS_BANK_CONFLICT" in per and per["SQ_LDS_IDX_ACTIVE"] > 0:
        out["lds_conflict_ratio"] = per["SQ_LDS_BANK_CONFLICT"] / per["SQ_LDS_IDX_ACTIVE"]
tp = "gpurun_out/scan_traffic_%s.json" % cfg
if os.path.exists(tp):
    out["hbm_bytes_per_launch"] = json.load(open(tp))["hbm_bytes_per_launch"]
    out["hbm_frac"] = out["hbm_bytes_per_launch"] / (ms * 1e-3) / 8.0e12
json.dump(out, open("gpurun_out/scan_binding_%s.json" % cfg, "w"), indent=1)
print(json.dumps(out))
