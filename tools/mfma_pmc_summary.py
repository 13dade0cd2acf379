#!/usr/bin/env python3
"""MFMA pipe utilisation per kernel from a tools/pmc_summary.py table holding SQ_VALU_MFMA_BUSY_CYCLES,
SQ_INSTS_VALU_MFMA_MOPS_F32 and GRBM_GUI_ACTIVE (tools/gpu_pmc_cnn.sh).  GRBM_GUI_ACTIVE is summed over the 8 XCDs, the
MFMA counters over the 1024 SIMDs; one MOPS unit = 512 flops; the f32 MFMA peak is 64 flop/clk/SIMD (157.3 TFLOP/s).
usage: mfma_pmc_summary.py <table.csv>"""
import collections, sys
d = collections.defaultdict(dict)
for line in open(sys.argv[1]).read().splitlines()[1:]:
    parts = line.rsplit(",", 4)  # kernel names contain commas
    if len(parts) < 5:
        continue
    k, n, c, tot, _ = parts
    d[k][c] = float(tot)
    d[k]["n"] = int(n)
busy = act_all = flop = 0.0
for k, v in d.items():
    act = v.get("GRBM_GUI_ACTIVE", 0) / 8
    act_all += act
    if v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) == 0:
        continue
    print("%-44s dispatches %3d  MFMA busy cycles / (active cycles x 1024 SIMDs) = %.3f   MFMA flops / f32 MFMA peak = %.3f" % (
        k[:44], v["n"], v["SQ_VALU_MFMA_BUSY_CYCLES"] / (act * 1024), v["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512 / (act * 65536)))
    busy += v["SQ_VALU_MFMA_BUSY_CYCLES"]
    flop += v["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512
print("all kernels of the run (pooling, layout changes and the generator's torch kernels included): MFMA busy %.3f, MFMA flops / peak %.3f" % (
    busy / (act_all * 1024), flop / (act_all * 65536)))
