cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for p in 3 4 5 2; do
  echo "== --pipeline $p"
  python bench.py --config c4 --no-cnn --no-pcie --no-cpu-baseline --no-c4x --pipeline $p 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['timing'])"
done
for l in 3 4 5; do
  echo "== CIS_BENCH_PCIE_LANES=$l"
  CIS_BENCH_PCIE_LANES=$l python bench.py --config c4 --no-cnn --no-cpu-baseline --no-c4x 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['pcie_inclusive'])"
done
