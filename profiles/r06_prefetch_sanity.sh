cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r06/p; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_joint_gpu.py -q -x -k "prefetch or graphed" 2>&1 | tail -3
timeout 600 python bench.py --skip-secondary --skip-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d['config'].get('backbone_prefetch'), d['config'].get('graph_capture_error'))"
