# A/B of an environment switch on one box: bash scripts/experiments/ab_env.sh VAR=a VAR=b ...
for i in 1 2 3; do
for E in "$@"; do
  env $E python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-concurrent --no-extras --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$E', d['value'], d['roofline']['kernel_ms'], d['config']['efSearch'], d.get('parity'))"
done; done
