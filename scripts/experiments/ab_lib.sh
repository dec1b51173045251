# A/B of library builds on one box: bash scripts/experiments/ab_lib.sh [lib.so ...]   ("" = the shipped library)
for i in 1 2 3; do
for L in "" "$@"; do
  HNSW_MI355X_LIB=${L:+$PWD/$L} python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-concurrent --no-extras --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${L:-shipped}', d['value'], d['roofline']['kernel_ms'], d['config']['efSearch'])"
done; done
