cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2; do for g in 1 0; do for sp in 2 1; do
  env VBX_GEMM5=$g VBX_SAMPLE_SPLIT=$sp timeout 300 python bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('GEMM5=$g SPLIT=$sp', d['ms_per_step'], 'ms')"
done; done; done
