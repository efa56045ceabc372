#!/bin/bash
# Last GPU call of round 5: the diffusion tests and the sampling bench entries on the final tree (4-way instruction attention in the
# sampler head; streaming body back to one half per update), phase probe for the record, smoke.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05y; mkdir -p $O
timeout 600 python -m pytest tests/test_diffusion_gpu.py -q > $O/t1.log 2>&1; echo "diffusion rc=$? $(grep -E 'passed|failed' $O/t1.log | tail -1)"; grep -E "^FAILED|^ERROR|fault" $O/t1.log | head
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | tail -2
for cfg in "64 16" "24 50"; do set -- $cfg; timeout 300 python bench_denoise.py --mode sample --batch $1 --horizon $2 > $O/s_$1_$2.json 2> $O/s.err; python - <<P
import json
try:
    d=json.load(open("$O/s_$1_$2.json")); print("B=$1 L=$2", round(d["value"],1), "traj/s", round(d["ms_per_denoise_step"],4), "ms/step", d["config"].get("sampler"), "graph-eager", d["config"].get("graph_vs_eager_max_abs_diff"), "roofline frac", round(d["roofline"]["frac"],4), "traffic", d["roofline"].get("traffic"))
except Exception as e: print("B=$1 L=$2 failed", e, open("$O/s.err").read()[-400:])
P
done
A3D_DN_PROF=1 timeout 200 python profiles/dn_persist_phases.py 6 > $O/phases.json 2> $O/phases.err; python - <<P
import json
try:
    d=json.load(open("$O/phases.json")); s=d["sample_0_step_1"]; print("head", s["head_us"], "tail", s["tail_us"], "step", s["step_us"], "abort", d["abort_word"])
    for i,l in enumerate(s["layers"]):
        if i in (0, 1, 4, 6): print(i, l)
except Exception as e: print("phases failed", e, open("$O/phases.err").read()[-600:])
P
