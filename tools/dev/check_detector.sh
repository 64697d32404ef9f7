set -e
python -m pytest tests/test_gpu_kernels.py -q -x -k "postprocess or nms" 2>&1 | tail -1
python -m pytest tests/test_gpu_models.py -q -x 2>&1 | tail -1
run() { python bench.py --workload detector --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-peaks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
for i in 1 2 3; do echo -n "this "; run; done
