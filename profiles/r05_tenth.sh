#!/bin/bash
# Tenth GPU call of round 5: persistent sampler -- phase probe, the multi-tile (L = 50) test, regression tests, bench.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05j; mkdir -p $O
A3D_DN_PROF=1 A3D_DN_PERSIST_SPLIT=4 timeout 200 python profiles/dn_persist_phases.py 6 > $O/phases_split4.json 2> $O/phases.err; python - <<P
import json
try:
    d=json.load(open("$O/phases_split4.json")); s=d["sample_0_step_1"]; print("head", s["head_us"], "tail", s["tail_us"], "step", s["step_us"], "abort", d["abort_word"])
    for i,l in enumerate(s["layers"]): print(i, l)
    print("items", d["streamer_0_items"][:12])
except Exception as e: print("phases failed", e, open("$O/phases.err").read()[-600:])
P
timeout 600 python -m pytest tests/test_diffusion_gpu.py -q -s -k "persistent or cfg3 or sampling_loop or fused_denoise" > $O/t.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/t.log | tail -1)"; grep -E "^FAILED|^ERROR|^E   |fault|L = 50" $O/t.log | head -20
for cfg in "0 8" "1 4"; do set -- $cfg; A3D_DN_PERSIST=$1 A3D_DN_PERSIST_SPLIT=$2 timeout 200 python bench_denoise.py --mode sample > $O/s_$1_$2.json 2> $O/s.err; python - <<P
import json
try:
    d=json.load(open("$O/s_$1_$2.json")); print("persist=$1 split=$2", round(d["value"],1), "traj/s", round(d["ms_per_denoise_step"],4), "ms/step", "graph-eager", d["config"].get("graph_vs_eager_max_abs_diff"))
except Exception as e: print("persist=$1 split=$2 failed", e, open("$O/s.err").read()[-400:])
P
done
