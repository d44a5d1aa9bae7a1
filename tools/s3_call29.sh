cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_host_cpu.py -m gpu -q -k "sampl or golden or wc or text or exports" 2>&1 | tail -3
timeout 300 python tools/sample_concurrent.py 16 2>&1 | tail -3
timeout 300 python bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('sample 64 intervals ms',d['ms_per_step'], d['step_roofline_frac'])"
