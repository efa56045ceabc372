cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r06/h; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_act3d_gpu.py tests/test_engine_gpu.py -q -s -k "fpn or lateral or golden or bf16 or graphed or conv3x3" 2>&1 | grep -E "parity\] (fpn lateral|fused)|passed|failed|Error|error|FAILED" | head -60 | tee $O/tests.txt
