#!/bin/bash
# The GPU-lease scripts of round 5 (one per gpurun call, in order), kept as one record: each section is what one call ran.

# ======================================================================== r05_first.sh
# First GPU call of round 5 (merged preparation branch): the whole parity suite once (no -x: every failure is attributable in one
# call), the keypose bench line with its A/B switches, the eager kernel traces of the keypose and the diffusion training step (compare
# kernel by kernel with profiles/r04_kernel_trace_*.txt), then the sq_bwd phase probe.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "all rc=$? $(grep -E 'passed|failed' $O/pytest_all.log | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest_all.log | head -20
timeout 600 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp.json 2> $O/bench_kp.err
A3D_CTX_SINK=0 A3D_FOLD_DS_BN=0 timeout 600 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp_nosink_nofold.json 2> /dev/null
for f in bench_kp bench_kp_nosink_nofold; do python - <<P
import json
try:
    d=json.load(open("$O/$f.json")); print("$f", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only"))
except Exception as e: print("$f", "failed", e)
P
done
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace" -o kp -- python "$GRAFT_REPO_ROOT/bench.py" --skip-secondary --skip-cpu-baseline --no-graph --steps 10 --warmup 4 > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1 )
DB=$(find $O/trace -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/kernel_trace_B64.txt 2>&1; rm -rf $O/trace
head -45 $O/kernel_trace_B64.txt | cut -c1-140
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace2" -o dt -- python "$GRAFT_REPO_ROOT/bench_denoise.py" --mode train --no-graph --reps 5 > "$GRAFT_REPO_ROOT/$O/trace_dt.log" 2>&1 )
DB=$(find $O/trace2 -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/kernel_trace_diffusion_train.txt 2>&1; rm -rf $O/trace2
head -30 $O/kernel_trace_diffusion_train.txt | cut -c1-140
timeout 200 python bench_denoise.py --mode train > $O/denoise_train.json 2> $O/denoise_train.err; cat $O/denoise_train.json | cut -c1-400
timeout 200 python bench_denoise.py --mode sample > $O/denoise_sample.json 2> $O/denoise_sample.err; cat $O/denoise_sample.json | cut -c1-400
timeout 120 python profiles/sq_bwd_phases.py > $O/sq_bwd_phases.json 2> $O/sq_phases.err; cat $O/sq_bwd_phases.json

# ======================================================================== r05_second.sh
# Second GPU call of round 5: the whole parity suite with its [parity] lines (-s) after the gate-node gradient sink, the
# cfg-3 graph test three more times (a capture-vs-replay difference was seen once in the first call), the bench line.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest_all.log 2>&1; echo "all rc=$? $(grep -E 'passed|failed' $O/pytest_all.log | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest_all.log | head -20
grep "\[parity\]" $O/pytest_all.log > $O/parity_report.txt; wc -l $O/parity_report.txt
for i in 1 2 3; do timeout 300 python -m pytest tests/test_diffusion_gpu.py -q -k "cfg3_full_shape" 2>&1 | tail -3 | grep -E "passed|failed|max abs diff"; done
timeout 600 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp.json 2> $O/bench_kp.err
python - <<P
import json
try:
    d=json.load(open("$O/bench_kp.json")); print("bench_kp", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only"))
except Exception as e: print("bench_kp failed", e)
P

# ======================================================================== r05_third.sh
# Third GPU call of round 5: the capture-vs-replay probe of the cfg-3 sampler, the wave-local single-query kernels (parity tests,
# micro-benchmark A/B against the round-4 kernels), the keypose tests that use them, the joint-iteration test.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05c; mkdir -p $O
timeout 300 python profiles/cfg3_graph_probe.py > $O/cfg3_probe.txt 2>&1; cat $O/cfg3_probe.txt | tail -12
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "single_query or query_stream or sq_" > $O/t_sq.log 2>&1; echo "sq rc=$? $(grep -E 'passed|failed' $O/t_sq.log | tail -1)"; grep -E "^FAILED|^ERROR|^E  " $O/t_sq.log | head -10
timeout 600 python -m pytest tests/test_act3d_gpu.py tests/test_joint_gpu.py tests/test_engine_gpu.py -q > $O/t_act3d.log 2>&1; echo "act3d rc=$? $(grep -E 'passed|failed' $O/t_act3d.log | tail -1)"; grep -E "^FAILED|^ERROR" $O/t_act3d.log | head -10
for w in 1 0; do A3D_SQ_WAVE=$w timeout 300 python bench.py --kernels-only > $O/kernels_wave$w.json 2> $O/kernels_wave$w.err; python - <<P
import json
try:
    k=json.load(open("$O/kernels_wave$w.json"))["kernels"]; print("A3D_SQ_WAVE=$w", {n: round(k[n]["ms"]*1e3,1) for n in ("sq_fwd","sq_bwd","attn_fwd","attn_bwd","kv_proj_rope") if n in k})
except Exception as e: print("kernels $w failed", e)
P
done
timeout 600 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp.json 2> $O/bench_kp.err
python - <<P
import json
try:
    d=json.load(open("$O/bench_kp.json")); print("bench_kp", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only"))
except Exception as e: print("bench_kp failed", e)
P

# ======================================================================== r05_fourth.sh
# Fourth GPU call of round 5: (1) which tree makes ONE fused denoise step run-to-run non-deterministic: the round-4 tree (.ab/r04),
# the current tree with round 4's sincos helper (.ab/v1), the current tree; (2) wave-local single-query kernels after the LDS /
# split-count fix: micro-benchmark A/B.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05d; mkdir -p $O
for d in .ab/r04 .ab/v1 .; do echo "== probe in $d"; (cd $d && timeout 200 python profiles/cfg3_graph_probe.py 2>&1 | grep -E "context run|n_steps|Error|error" | head -8); done > $O/cfg3_probe_ab.txt 2>&1; cat $O/cfg3_probe_ab.txt
for w in 1 0; do A3D_SQ_WAVE=$w timeout 300 python bench.py --kernels-only > $O/kernels_wave$w.json 2> $O/kernels_wave$w.err; python - <<P
import json
try:
    k=json.load(open("$O/kernels_wave$w.json"))["kernels"]; print("A3D_SQ_WAVE=$w", {n: round(k[n]["ms"]*1e3,1) for n in ("sq_fwd","sq_bwd") if n in k})
except Exception as e: print("kernels $w failed", e)
P
done
A3D_SQ_WGS=1024 timeout 300 python bench.py --kernels-only > $O/kernels_wave1_1024.json 2> /dev/null; python - <<P
import json
try:
    k=json.load(open("$O/kernels_wave1_1024.json"))["kernels"]; print("A3D_SQ_WAVE=1 WGS=1024", {n: round(k[n]["ms"]*1e3,1) for n in ("sq_fwd","sq_bwd") if n in k})
except Exception as e: print("failed", e)
P
A3D_SQ_WGS=256 timeout 300 python bench.py --kernels-only > $O/kernels_wave1_256.json 2> /dev/null; python - <<P
import json
try:
    k=json.load(open("$O/kernels_wave1_256.json"))["kernels"]; print("A3D_SQ_WAVE=1 WGS=256", {n: round(k[n]["ms"]*1e3,1) for n in ("sq_fwd","sq_bwd") if n in k})
except Exception as e: print("failed", e)
P
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "single_query or query_stream or sq_" 2>&1 | tail -2

# ======================================================================== r05_fifth.sh
# Fifth GPU call of round 5: whole parity suite after the call-free sincos (determinism of the fused denoise step, joint test),
# forward-kernel occupancy A/B (2 vs 3 workgroups per CU), bench line.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05e; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest_all.log 2>&1; echo "all rc=$? $(grep -E 'passed|failed' $O/pytest_all.log | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest_all.log | head -20
grep "\[parity\]" $O/pytest_all.log > $O/parity_report.txt; wc -l $O/parity_report.txt
timeout 200 python profiles/cfg3_graph_probe.py 2>&1 | grep -E "n_steps" | head -4
for cfg in "libact3d_hip.so 512" "libact3d_hip_occ3.so 512" "libact3d_hip_occ3.so 768"; do set -- $cfg; A3D_LIB=$1 A3D_SQ_WGS=$2 timeout 300 python bench.py --kernels-only > $O/k.json 2> /dev/null; python - <<P
import json
try:
    k=json.load(open("$O/k.json"))["kernels"]; print("$1 WGS=$2", {n: round(k[n]["ms"]*1e3,1) for n in ("sq_fwd","sq_bwd") if n in k})
except Exception as e: print("failed", e)
P
done
timeout 600 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp.json 2> $O/bench_kp.err
python - <<P
import json
try:
    d=json.load(open("$O/bench_kp.json")); print("bench_kp", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only"))
except Exception as e: print("bench_kp failed", e)
P

# ======================================================================== r05_sixth.sh
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

# ======================================================================== later calls (seventh .. last): see the scripts kept beside this file
#   r05_seventh.sh .. r05_sixteenth.sh  persistent sampler bring-up (fault isolation, fences, prefetch depth, helper workgroups, splits)
#   r05_final.sh                        the evidence run (parity suite, bench line, traces)
#   r05_pmc.sh                          counter passes
#   r05_last.sh, r05_rope_ab.sh         final tree check, rope_merge_bwd A/B
