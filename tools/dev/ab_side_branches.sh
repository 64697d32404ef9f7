set -e
python -m pytest tests/test_gpu_models.py tests/test_gpu_dist.py -q -x 2>&1 | tail -2
run() { python bench.py "$@" --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-peaks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
for i in 1 2; do
  echo -n "det serial  "; CVPCE_SIDE_BRANCHES=0 run --workload detector
  echo -n "det beside  "; run --workload detector
done
