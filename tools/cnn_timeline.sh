#!/bin/bash
# per-kernel timeline of the last DeepSentibank forward (rocprofv3 kernel trace); usage: tools/cnn_timeline.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/cn; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/cn -o r -- python tools/bench_cnn.py $1 2>&1 | grep batch
python - <<PY
import csv, re
rows=list(csv.DictReader(open("/tmp/cn/r_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "nchw3_to_nhwc" in r["Kernel_Name"]]
if not idx: idx=[i for i,r in enumerate(rows) if "k_conv_igemm<1, 3, 4, 1, 3>" in r["Kernel_Name"] or "k_conv_igemm<1, 3, 4, 1, 2>" in r["Kernel_Name"]]
tot=0
for r in rows[idx[-1]:]:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3; tot+=d
    print("%-44s %8.1f us  blocks %6d x %s x %s" % (re.sub(r"\(.*","",r["Kernel_Name"])[:44], d, int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["Grid_Size_Z"]))
print("total", tot)
PY
