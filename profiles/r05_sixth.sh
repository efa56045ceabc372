#!/bin/bash
# Sixth GPU call of round 5: the persistent sampler (a3d_dn_persist) -- parity against the per-phase launches and the oracle,
# then the cfg-3 sampling bench with it on / off and at 4 / 8 / 16 key splits.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05f; mkdir -p $O
timeout 300 python -m pytest tests/test_diffusion_gpu.py -q -x -s -k "persistent" > $O/t_persist.log 2>&1; echo "persist rc=$? $(grep -E 'passed|failed' $O/t_persist.log | tail -1)"; grep -E "^FAILED|^ERROR|^E  |\[parity\] persistent" $O/t_persist.log | head -20
timeout 600 python -m pytest tests/test_diffusion_gpu.py -q -s > $O/t_diff.log 2>&1; echo "diffusion rc=$? $(grep -E 'passed|failed' $O/t_diff.log | tail -1)"; grep -E "^FAILED|^ERROR" $O/t_diff.log | head
for cfg in "0 8" "1 8" "1 4" "1 16"; do set -- $cfg; A3D_DN_PERSIST=$1 A3D_DN_PERSIST_SPLIT=$2 timeout 200 python bench_denoise.py --mode sample > $O/s.json 2> $O/s.err; python - <<P
import json
try:
    d=json.load(open("$O/s.json")); print("persist=$1 split=$2", round(d["value"],1), "traj/s", round(d["ms_per_denoise_step"],4), "ms/step")
except Exception as e: print("persist=$1 split=$2 failed", e, open("$O/s.err").read()[-400:])
P
done
