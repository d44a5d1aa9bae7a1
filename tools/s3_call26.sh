cd $GRAFT_REPO_ROOT
for v in 257 96 257 96; do
VBX_BM160_MIN=$v timeout 200 python bench.py --mode sample --steps 2 --warmup 1 --intervals 16 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('min160=$v sample ms',d['ms_per_step'])"
done
