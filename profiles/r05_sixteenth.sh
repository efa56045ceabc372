#!/bin/bash
# Sixteenth GPU call of round 5: three-halves-per-update streaming (ping-pong fragment groups), 4-way instruction attention in the
# sampler head, constant-divisor rope_merge_bwd: tests, sampling bench, phase probe, keypose bench.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05q; mkdir -p $O
timeout 600 python -m pytest tests/test_diffusion_gpu.py -q -k "persistent or cfg3 or sampling_loop or fused_denoise" > $O/t1.log 2>&1; echo "diffusion rc=$? $(grep -E 'passed|failed' $O/t1.log | tail -1)"; grep -E "^FAILED|^ERROR|fault" $O/t1.log | head
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attn_block or rope or merge or query_stream" > $O/t2.log 2>&1; echo "kernels rc=$? $(grep -E 'passed|failed' $O/t2.log | tail -1)"; grep -E "^FAILED|^ERROR" $O/t2.log | head
for cfg in "1 4" "1 8"; do set -- $cfg; A3D_DN_PERSIST=$1 A3D_DN_PERSIST_SPLIT=$2 timeout 200 python bench_denoise.py --mode sample > $O/s_$1_$2.json 2> $O/s.err; python - <<P
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
timeout 600 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp.json 2> $O/bench_kp.err
python - <<P
import json
try:
    d=json.load(open("$O/bench_kp.json")); print("bench_kp", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only"))
except Exception as e: print("bench_kp failed", e)
P
