#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04c; mkdir -p $O
python -m pytest tests/test_joint_gpu.py -q -s -k "joint_iteration" > $O/joint.log 2>&1; grep -E "parity|passed|failed" $O/joint.log | tail -12
python -m pytest tests/test_data_gpu.py -q -s -k "reference or matches" > $O/harness.log 2>&1; grep -E "parity|passed|failed|Error|assert" $O/harness.log | tail -20
