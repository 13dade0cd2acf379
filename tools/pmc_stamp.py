#!/usr/bin/env python3
"""Provenance of the PMC summaries bench.py replays (profiles/scan_{traffic,binding}_*.json).  PMC counters need their own rocprofv3
passes, so the driver's bench line cannot measure them: it prints the committed values -- stamped with WHICH kernel build they were
taken from, so that a stale file is named as such (`roofline.pmc_source.sources_match_this_build`).

    kernel_sources_sha1()            sha1 of every csrc file a scan kernel is compiled from (run on the GPU box: no .git there)
    tools/pmc_stamp.py FILE...       adds "commit" = git HEAD of this checkout to JSON summaries being copied into profiles/
"""
import hashlib
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ("lopq_scan3.hip", "lopq_search.hip", "lopq_stream.hip", "scan_common.h", "common.h")


def kernel_sources_sha1():
    out = {}
    for f in KERNEL_SOURCES:
        p = os.path.join(REPO, "columbiaimagesearch_amd", "csrc", f)
        try:
            out[f] = hashlib.sha1(open(p, "rb").read()).hexdigest()
        except OSError:
            out[f] = None
    return out


if __name__ == "__main__":
    head = subprocess.check_output(["git", "-C", REPO, "rev-parse", "HEAD"]).decode().strip()
    dirty = bool(subprocess.check_output(["git", "-C", REPO, "status", "--porcelain", "--", "columbiaimagesearch_amd/csrc"]).decode().strip())
    for p in sys.argv[1:]:
        d = json.load(open(p))
        d["commit"] = head + ("+uncommitted csrc changes" if dirty else "")
        json.dump(d, open(p, "w"), indent=1)
        print("stamped", p, d["commit"])
