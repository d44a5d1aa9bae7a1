cd $GRAFT_REPO_ROOT
export VBX_LIB_PATH=$GRAFT_REPO_ROOT/voicebox-pytorch_amd/lib/libvbx_hip_rsq.so
timeout 300 python -m pytest tests/test_model_gpu.py -q -k "cfg4 or cfg1 or small_golden" 2>&1 | grep -E "passed|failed|cfg4 \(depth|assert " | head
timeout 200 python bench.py --mode sample --steps 2 --warmup 1 --intervals 16 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('rsq+NR sample ms',d['ms_per_step'])"
unset VBX_LIB_PATH
timeout 200 python bench.py --mode sample --steps 2 --warmup 1 --intervals 16 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('IEEE sample ms',d['ms_per_step'])"
