#!/usr/bin/env python3
"""HBM/fabric bytes per full scan launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/gpu_pmc.sh tables).
usage: scan_traffic.py <tag> <config> <full_launch_equivalents> <algorithmic_bytes_per_launch>  -> gpurun_out/scan_traffic_<config>.json
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of a wide coalesced read -> doubled;
WRITE_SIZE as reported; both are in KB."""
import json, sys
tag, config, equiv, algo = sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4])


def total(counter):
    tot, kern = 0.0, None
    for line in open("gpurun_out/%s_%s_%s_pmc.csv" % (tag, config, counter)).read().splitlines()[1:]:
        parts = line.rsplit(",", 4)
        if len(parts) == 5 and "k_adc_scan" in parts[0] and parts[2] == counter:
            tot += float(parts[3])
            kern = parts[0]
    return tot, kern


f, kern = total("FETCH_SIZE")
w, _ = total("WRITE_SIZE")
out = {"kernel": kern, "config": config,
       "source": "gpurun_out/%s_%s_{FETCH_SIZE,WRITE_SIZE}_pmc.csv: rocprofv3 --pmc in separate passes with --kernel-trace only, "
                 "bench.py --config %s --steps 3 --warmup 1 --no-cpu-baseline --no-cnn --no-pcie" % (tag, config, config),
       "fetch_size_kb_total": f, "write_size_kb_total": w, "full_launch_equivalents": equiv,
       "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read -> doubled (the scan reads 8-16 B/lane: "
                     "upper-bound style estimate); WRITE_SIZE as reported",
       "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0 / equiv, "write_bytes_per_launch": w * 1024.0 / equiv,
       "algorithmic_bytes_per_launch": algo}
json.dump(out, open("gpurun_out/scan_traffic_%s.json" % config, "w"), indent=1)
print(json.dumps(out))
