#!/bin/bash
# Eleventh GPU call of round 5: persistent sampler with finer key splits after the head-staging fix (phase probe + bench).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05k; mkdir -p $O
for sp in 8 16; do
A3D_DN_PROF=1 A3D_DN_PERSIST_SPLIT=$sp timeout 200 python profiles/dn_persist_phases.py 6 > $O/phases_split$sp.json 2> $O/phases.err; python - <<P
import json
try:
    d=json.load(open("$O/phases_split$sp.json")); s=d["sample_0_step_1"]; print("split $sp: head", s["head_us"], "tail", s["tail_us"], "step", s["step_us"], "abort", d["abort_word"])
    for i,l in enumerate(s["layers"][:3]): print(i, l)
    print("items", d["streamer_0_items"][:6])
except Exception as e: print("phases failed", e, open("$O/phases.err").read()[-600:])
P
A3D_DN_PERSIST=1 A3D_DN_PERSIST_SPLIT=$sp timeout 200 python bench_denoise.py --mode sample > $O/s_1_$sp.json 2> $O/s.err; python - <<P
import json
try:
    d=json.load(open("$O/s_1_$sp.json")); print("persist=1 split=$sp", round(d["value"],1), "traj/s", round(d["ms_per_denoise_step"],4), "ms/step", "graph-eager", d["config"].get("graph_vs_eager_max_abs_diff"))
except Exception as e: print("persist=1 split=$sp failed", e, open("$O/s.err").read()[-400:])
P
done
