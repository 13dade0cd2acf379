cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for g in 512 256 768 1024; do
  echo "== CIS_STREAM_GRID=$g"
  CIS_STREAM_GRID=$g python tools/r06_stream_lib.py 200000000 1 2>&1 | grep "exhaustive"
done
rm -rf /tmp/prof_s
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o r -- python tools/r06_stream_lib.py 200000000 1 > /tmp/s.log 2>&1
python tools/kstats.py /tmp/prof_s/r_kernel_stats.csv "stream"
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/prof_s/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'k_adc_stream<8, 1, false>' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
print("k_adc_stream<8,1,false> dispatches:", " ".join("%.0f"%x for x in d))
PY
tools/probes/stream_probe 200000000 8 2>&1 | grep "library"
