#!/bin/bash
# per-kernel timeline of the last dlib forward (rocprofv3 kernel trace); usage: tools/dlib_timeline.sh [batch]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/dl; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/dl -o r -- python tools/bench_dlib.py $1 2>&1 | grep batch
python - <<PY
import csv, re
rows=list(csv.DictReader(open("/tmp/dl/r_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "normalize" in r["Kernel_Name"] or "conv7x7" in r["Kernel_Name"]]
tot=0
for r in rows[idx[-1]:]:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3; tot+=d
    print("%-44s %8.1f us  blocks %6d x %s x %s" % (re.sub(r"\(.*","",r["Kernel_Name"])[:44], d, int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["Grid_Size_Z"]))
print("total", tot)
PY
