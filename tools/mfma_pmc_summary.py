#!/usr/bin/env python3
"""MFMA pipe utilisation of a CNN forward, reconciled between the PMC counters and the bench's wall clock.

Inputs: a tools/pmc_summary.py table holding SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_F32 and GRBM_GUI_ACTIVE (tools/gpu_pmc_cnn.sh),
the number of forwards the profiled command ran, the useful multiply-adds per item, the batch, and the line the same command printed WITHOUT the
profiler ("batch B: T ms ...").  GRBM_GUI_ACTIVE is summed over the 8 XCDs, the MFMA counters over the 1024 SIMDs; one MOPS unit = 512 flops;
the f32 MFMA peak is 64 flop/clk/SIMD (157.3 TFLOP/s at 2.4 GHz).

Three figures per network:
  (a) useful flops / wall time / peak                     -- what tools/bench_*.py and bench.py report
  (b) MFMA flops ISSUED (PMC) / wall time / peak          -- (a) + the zero padding of K and of the tiles; must be >= (a)
  (c) MFMA busy cycles / active cycles (PMC), per kernel  -- the kernel ALONE on the chip (counter collection serialises the launches and
      adds ~7 us of counter start / stop per launch to GRBM_GUI_ACTIVE), so (c) of short kernels is below what they reach in the pipelined run
usage: mfma_pmc_summary.py <table.csv> <forwards> <mac per item> <batch> <wall ms per forward>"""
import collections, sys
path, nfwd, mac, batch, wall_ms = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
PEAK = 157.3e12
d = collections.defaultdict(dict)
for line in open(path).read().splitlines()[1:]:
    parts = line.rsplit(",", 4)  # kernel names contain commas
    if len(parts) < 5:
        continue
    k, n, c, tot, _ = parts
    d[k][c] = float(tot)
    d[k]["n"] = int(n)
busy = act_mfma = flop = 0.0
for k, v in sorted(d.items()):
    if v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) == 0:
        continue
    act = v.get("GRBM_GUI_ACTIVE", 0) / 8
    print("(c) %-44s launches/forward %5.1f  MFMA busy / active = %.3f   us per launch (active cycles at 2.4 GHz) %7.1f" % (
        k[:44], v["n"] / nfwd, v["SQ_VALU_MFMA_BUSY_CYCLES"] / (act * 1024), act / v["n"] / 2400.0))
    busy += v["SQ_VALU_MFMA_BUSY_CYCLES"]; act_mfma += act
    flop += v["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512
useful = 2.0 * mac * batch
issued = flop / nfwd
print("(c) all MFMA kernels of the forward, each alone on the chip: MFMA busy / active = %.3f" % (busy / (act_mfma * 1024)))
print("useful flops per forward  %.4e  (2 x %d multiply-adds x %d items)" % (useful, int(mac), batch))
print("issued MFMA flops (PMC)   %.4e  = %.3f x useful (zero padding of K and of the tiles' edges)" % (issued, issued / useful))
print("(a) useful / wall / peak  %.3f   (wall %.3f ms per forward, unprofiled run of the same command)" % (useful / (wall_ms * 1e-3) / PEAK, wall_ms))
print("(b) issued / wall / peak  %.3f" % (issued / (wall_ms * 1e-3) / PEAK))
