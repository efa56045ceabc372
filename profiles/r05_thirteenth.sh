#!/bin/bash
# Thirteenth GPU call of round 5: three-deep K/V prefetch in fixed register sets (stream role + dn_cross_kernel): diffusion tests
# on both sampler paths, sampling bench A/B, phase probe.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05m; mkdir -p $O
timeout 600 python -m pytest tests/test_diffusion_gpu.py -q > $O/t1.log 2>&1; echo "diffusion (persist) rc=$? $(grep -E 'passed|failed' $O/t1.log | tail -1)"; grep -E "^FAILED|^ERROR|fault" $O/t1.log | head
A3D_DN_PERSIST=0 timeout 600 python -m pytest tests/test_diffusion_gpu.py -q -k "not persistent" > $O/t0.log 2>&1; echo "diffusion (per-phase) rc=$? $(grep -E 'passed|failed' $O/t0.log | tail -1)"; grep -E "^FAILED|^ERROR|fault" $O/t0.log | head
for cfg in "0 8" "1 4" "1 8" "1 16"; do set -- $cfg; A3D_DN_PERSIST=$1 A3D_DN_PERSIST_SPLIT=$2 timeout 200 python bench_denoise.py --mode sample > $O/s_$1_$2.json 2> $O/s.err; python - <<P
import json
try:
    d=json.load(open("$O/s_$1_$2.json")); print("persist=$1 split=$2", round(d["value"],1), "traj/s", round(d["ms_per_denoise_step"],4), "ms/step", "graph-eager", d["config"].get("graph_vs_eager_max_abs_diff"))
except Exception as e: print("persist=$1 split=$2 failed", e, open("$O/s.err").read()[-400:])
P
done
A3D_DN_PROF=1 A3D_DN_PERSIST_SPLIT=8 timeout 200 python profiles/dn_persist_phases.py 6 > $O/phases_split8.json 2> $O/phases.err; python - <<P
import json
try:
    d=json.load(open("$O/phases_split8.json")); s=d["sample_0_step_1"]; print("split 8: head", s["head_us"], "tail", s["tail_us"], "step", s["step_us"], "abort", d["abort_word"])
    for i,l in enumerate(s["layers"]):
        if i in (0, 1, 4, 6): print(i, l)
    print("items", d["streamer_0_items"][:5])
except Exception as e: print("phases failed", e, open("$O/phases.err").read()[-600:])
P
