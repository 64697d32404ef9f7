set -e
run() { python bench.py --workload detector --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-peaks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
for i in 1 2 3; do
  echo -n "default  "; run
  echo -n "any 1x1  "; CVPCE_CONV1X1_ANY=1 run
done
