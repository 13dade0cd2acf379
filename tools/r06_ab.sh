cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_stream_route.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
for ch in 0 65536; do
  echo "== CIS_STREAM_CHUNK=$ch"
  CIS_STREAM_CHUNK=$ch python tools/r06_stream_lib.py 200000000 1,2,4 2>&1 | grep -v amdgpu.ids | grep -v "route =="
done
done
