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

# ======================================================================== r05_seventh.sh
# Seventh GPU call of round 5: (1) the memory fault seen in test_cfg3_full_shape_graph_vs_oracle with the persistent sampler: alone,
# with the per-phase path, in file order; (2) the sampling bench after the parallel key-split combine.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05g; mkdir -p $O
A3D_DN_PERSIST=1 timeout 200 python -m pytest tests/test_diffusion_gpu.py -q -x -k "cfg3_full_shape" > $O/t1.log 2>&1; echo "cfg3 alone persist=1 rc=$? $(grep -E 'passed|failed|fault' $O/t1.log | tail -2)"
A3D_DN_PERSIST=0 timeout 200 python -m pytest tests/test_diffusion_gpu.py -q -x -k "cfg3_full_shape" > $O/t0.log 2>&1; echo "cfg3 alone persist=0 rc=$? $(grep -E 'passed|failed|fault' $O/t0.log | tail -2)"
A3D_DN_PERSIST=1 timeout 300 python -m pytest tests/test_diffusion_gpu.py -q -x -k "sampling_loop or cfg3_full_shape" > $O/t2.log 2>&1; echo "loop+cfg3 persist=1 rc=$? $(grep -E 'passed|failed|fault' $O/t2.log | tail -2)"
A3D_DN_PERSIST=0 timeout 600 python -m pytest tests/test_diffusion_gpu.py -q > $O/t3.log 2>&1; echo "file persist=0 rc=$? $(grep -E 'passed|failed|fault' $O/t3.log | tail -2)"
for cfg in "1 2" "1 4" "1 8"; do set -- $cfg; A3D_DN_PERSIST=$1 A3D_DN_PERSIST_SPLIT=$2 timeout 200 python bench_denoise.py --mode sample > $O/s.json 2> $O/s.err; python - <<P
import json
try:
    d=json.load(open("$O/s.json")); print("persist=$1 split=$2", round(d["value"],1), "traj/s", round(d["ms_per_denoise_step"],4), "ms/step")
except Exception as e: print("persist=$1 split=$2 failed", e, open("$O/s.err").read()[-400:])
P
done

# ======================================================================== r05_eighth.sh
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

# ======================================================================== r05_ninth.sh
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

# ======================================================================== r05_tenth.sh
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

# ======================================================================== r05_eleventh.sh
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

# ======================================================================== r05_twelfth.sh
# Twelfth GPU call of round 5: persistent sampler with primary / helper workgroups (position and rotation stacks in parallel) and
# fence-free coherent stores: tests, phase probes at 4 / 8 / 16 splits, bench.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05l; mkdir -p $O
timeout 600 python -m pytest tests/test_diffusion_gpu.py -q -s -k "persistent or cfg3 or sampling_loop or fused_denoise" > $O/t.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/t.log | tail -1)"; grep -E "^FAILED|^ERROR|^E   |fault|L = 50|persistent vs per-phase" $O/t.log | head -20
for sp in 4 8 16; do
A3D_DN_PROF=1 A3D_DN_PERSIST_SPLIT=$sp timeout 200 python profiles/dn_persist_phases.py 6 > $O/phases_split$sp.json 2> $O/phases.err; python - <<P
import json
try:
    d=json.load(open("$O/phases_split$sp.json")); s=d["sample_0_step_1"]; print("split $sp: head", s["head_us"], "tail", s["tail_us"], "step", s["step_us"], "abort", d["abort_word"])
    for i,l in enumerate(s["layers"]):
        if i in (0, 1, 4, 6): print(i, l)
    print("items", d["streamer_0_items"][:5])
except Exception as e: print("phases failed", e, open("$O/phases.err").read()[-600:])
P
A3D_DN_PERSIST=1 A3D_DN_PERSIST_SPLIT=$sp timeout 200 python bench_denoise.py --mode sample > $O/s_1_$sp.json 2> $O/s.err; python - <<P
import json
try:
    d=json.load(open("$O/s_1_$sp.json")); print("persist=1 split=$sp", round(d["value"],1), "traj/s", round(d["ms_per_denoise_step"],4), "ms/step", "graph-eager", d["config"].get("graph_vs_eager_max_abs_diff"))
except Exception as e: print("persist=1 split=$sp failed", e, open("$O/s.err").read()[-400:])
P
done

# ======================================================================== r05_thirteenth.sh
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

# ======================================================================== r05_sixteenth.sh
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

# ======================================================================== r05_final.sh
# Round-5 evidence run: the whole parity suite with its [parity] lines, the default bench line (secondary entries + CPU baseline),
# eager kernel traces of the keypose step, the diffusion training step and the samplers.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05z; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest_all.log 2>&1; echo "all rc=$? $(grep -E 'passed|failed' $O/pytest_all.log | tail -1)"; grep -E "^FAILED|^ERROR|fault" $O/pytest_all.log | head -20
grep "\[parity\]" $O/pytest_all.log > $O/parity_report.txt; wc -l $O/parity_report.txt
timeout 900 python bench.py > $O/bench_B64.json 2> $O/bench_B64.err; echo "bench rc=$?"
python - <<P
import json
try:
    d=json.load(open("$O/bench_B64.json")); print("bench", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only"), "roofline", d["roofline"].get("kernel"), round(d["roofline"].get("frac",0),4))
    for s in d.get("secondary", []): print("  ", s.get("name"), s.get("value"), s.get("unit"), s.get("ms_per_step") or s.get("ms_per_denoise_step"), s.get("error"), (s.get("config") or {}).get("sampler"))
    print("  cpu", d.get("cpu_baseline"))
    print("  kernels", {k: round(v["ms"]*1e3,1) for k,v in d.get("kernels",{}).items()})
except Exception as e: print("bench parse failed", e, open("$O/bench_B64.err").read()[-500:])
P
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace" -o kp -- python "$GRAFT_REPO_ROOT/bench.py" --skip-secondary --skip-cpu-baseline --no-graph --steps 10 --warmup 4 > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1 )
DB=$(find $O/trace -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/kernel_trace_B64.txt 2>&1; rm -rf $O/trace
head -30 $O/kernel_trace_B64.txt | cut -c1-130
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace2" -o dt -- python "$GRAFT_REPO_ROOT/bench_denoise.py" --mode train --no-graph --reps 5 > "$GRAFT_REPO_ROOT/$O/trace_dt.log" 2>&1 )
DB=$(find $O/trace2 -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/kernel_trace_diffusion_train.txt 2>&1; rm -rf $O/trace2
head -12 $O/kernel_trace_diffusion_train.txt | cut -c1-130
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace3" -o ds -- python "$GRAFT_REPO_ROOT/bench_denoise.py" --mode sample --reps 3 > "$GRAFT_REPO_ROOT/$O/trace_ds.log" 2>&1 )
find $O/trace3 -name '*kernel_stats.csv' | head -1 | xargs -r head -12 > $O/kernel_stats_sampling.txt; rm -rf $O/trace3; cat $O/kernel_stats_sampling.txt | cut -c1-160

# ======================================================================== r05_pmc.sh
# Round-5 counter passes (FETCH_SIZE | WRITE_SIZE | MfmaUtil VALUBusy, one --pmc set per pass, --kernel-trace only): the dominant
# keypose kernels (bench.py --kernels-only) and the samplers (persistent and per-phase).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05p; mkdir -p $O
timeout 600 bash profiles/pmc_json_cmd.sh $O/pmc_B64.json 64 python "$GRAFT_REPO_ROOT/bench.py" --kernels-only > $O/pmc_B64.log 2>&1; tail -3 $O/pmc_B64.log
python - <<P
import json
try:
    d=json.load(open("$O/pmc_B64.json"))["kernels"]
    for k,v in d.items(): print(k, round(v["hbm_bytes"]/1e6,1), "MB", v.get("pmc"))
except Exception as e: print("pmc B64 failed", e)
P
A3D_DN_PERSIST=1 timeout 400 bash profiles/pmc_json_cmd.sh $O/pmc_denoise_persist.json 64 python "$GRAFT_REPO_ROOT/bench_denoise.py" --mode sample --reps 1 > $O/pmc_dn1.log 2>&1; tail -3 $O/pmc_dn1.log
A3D_DN_PERSIST=0 timeout 400 bash profiles/pmc_json_cmd.sh $O/pmc_denoise_perphase.json 64 python "$GRAFT_REPO_ROOT/bench_denoise.py" --mode sample --reps 1 > $O/pmc_dn0.log 2>&1; tail -3 $O/pmc_dn0.log
python - <<P
import json
for f in ("pmc_denoise_persist", "pmc_denoise_perphase"):
    try:
        d=json.load(open("$O/%s.json" % f))["kernels"]
        for k,v in d.items(): print(f, k, round(v["hbm_bytes"]/1e6,1), "MB", v.get("pmc"))
    except Exception as e: print(f, "failed", e)
P

# ======================================================================== r05_last.sh
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

# ======================================================================== r05_rope_ab.sh
# A/B of the constant-divisor rope_merge_bwd kernel inside the eager keypose step (kernel trace, per-kernel averages).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05r; mkdir -p $O
for v in 0 1; do
( cd /tmp && export TMPDIR=/tmp && A3D_ROPE_MERGE_PAIRS=$v timeout 200 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace$v" -o kp -- python "$GRAFT_REPO_ROOT/bench.py" --skip-secondary --skip-cpu-baseline --no-graph --steps 10 --warmup 4 > "$GRAFT_REPO_ROOT/$O/trace$v.log" 2>&1 )
DB=$(find $O/trace$v -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/kernel_trace_pairs$v.txt 2>&1; rm -rf $O/trace$v
echo "pairs=$v"; head -1 $O/kernel_trace_pairs$v.txt; grep -E "rope_merge|attn16_fwd|sqw_bwd" $O/kernel_trace_pairs$v.txt | cut -c1-110
done

# ======================================================================== r05_dropfold.sh
# Dropout launches folded into their producers (a3d_linear_fwd_drop, a3d_add_layernorm_bwd_drop), gradient sums folded into the
# dgrad kernels: whole parity suite on the tree, A/B of the diffusion training step (fold on / off, wgrad two-stage threshold),
# default bench line, smoke.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05s; mkdir -p $O
timeout 420 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "all rc=$? $(grep -E 'passed|failed' $O/pytest_all.log | tail -1)"; grep -E "^FAILED|^ERROR|fault" $O/pytest_all.log | head -20
for e in "A3D_X=0" "A3D_DROPOUT_FOLD=0" "A3D_WGRAD_TWO_STAGE_MIN_ROWS=2048"; do
  env $e timeout 120 python bench_denoise.py --mode train > $O/train_$e.json 2> $O/train.err; python - <<P
import json
try:
    d=json.load(open("$O/train_$e.json")); print("$e", "train", round(d["value"],1), d.get("unit"), round(d.get("ms_per_step",0),3), "ms")
except Exception as ex: print("$e failed", ex, open("$O/train.err").read()[-400:])
P
done
timeout 400 python bench.py > $O/bench_B64.json 2> $O/bench_B64.err; echo "bench rc=$?"
python - <<P
import json
try:
    d=json.load(open("$O/bench_B64.json")); print("bench", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only"))
    for s in d.get("secondary", []): print("  ", s.get("name"), s.get("value"), s.get("unit"), s.get("ms_per_step") or s.get("ms_per_denoise_step"), s.get("error"))
except Exception as ex: print("bench parse failed", ex, open("$O/bench_B64.err").read()[-500:])
P
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | tail -2

