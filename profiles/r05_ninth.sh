#!/bin/bash
# Ninth GPU call of round 5: persistent sampler after the zeroing kernel replaced the captured memset -- probe, tests, bench A/B.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05i; mkdir -p $O
timeout 300 python profiles/cfg3_graph_probe.py > $O/probe.txt 2>&1; grep -E "n_steps=|fault|Error" $O/probe.txt | head -12
timeout 600 python -m pytest tests/test_diffusion_gpu.py -q > $O/t_diff.log 2>&1; echo "diffusion rc=$? $(grep -E 'passed|failed' $O/t_diff.log | tail -1)"; grep -E "^FAILED|^ERROR|fault" $O/t_diff.log | head
for cfg in "0 8" "1 2" "1 4" "1 8"; do set -- $cfg; A3D_DN_PERSIST=$1 A3D_DN_PERSIST_SPLIT=$2 timeout 200 python bench_denoise.py --mode sample > $O/s_$1_$2.json 2> $O/s.err; python - <<P
import json
try:
    d=json.load(open("$O/s_$1_$2.json")); print("persist=$1 split=$2", round(d["value"],1), "traj/s", round(d["ms_per_denoise_step"],4), "ms/step", "graph-eager", d["config"].get("graph_vs_eager_max_abs_diff"))
except Exception as e: print("persist=$1 split=$2 failed", e, open("$O/s.err").read()[-400:])
P
done
