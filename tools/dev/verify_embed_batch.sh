set -e
d() { python bench.py --verify --steps 2 --warmup 1 --no-parity --no-h2d --no-cpu-baseline --no-roofline --no-peaks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['verify']['digest'])"; }
echo -n "256: "; CVPCE_EMBED_BATCH=256 d
echo -n "768: "; d
python -m pytest tests/test_gpu_models.py tests/test_gpu_harness.py -q -x 2>&1 | tail -2
