#!/bin/bash
# Eighth GPU call of round 5: persistent sampler -- where the cfg-3 memory fault comes from (probe with progress prints), and the
# sampling bench without acquire fences / with non-temporal K-V loads.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05h; mkdir -p $O
timeout 300 python profiles/cfg3_graph_probe.py > $O/probe.txt 2>&1; grep -E "^--|abort|n_steps=|fault|Error" $O/probe.txt | head -30
for cfg in "1 4" "1 8" "1 2"; do set -- $cfg; A3D_DN_PERSIST=$1 A3D_DN_PERSIST_SPLIT=$2 timeout 200 python bench_denoise.py --mode sample > $O/s.json 2> $O/s.err; python - <<P
import json
try:
    d=json.load(open("$O/s.json")); print("persist=$1 split=$2", round(d["value"],1), "traj/s", round(d["ms_per_denoise_step"],4), "ms/step")
except Exception as e: print("persist=$1 split=$2 failed", e, open("$O/s.err").read()[-400:])
P
done
timeout 300 python -m pytest tests/test_diffusion_gpu.py -q -x -s -k "persistent" 2>&1 | grep -E "passed|failed|persistent vs per-phase" | head -5
