#!/bin/bash
# The GPU-lease scripts of round 4 (one per gpurun call, in order), kept as one record: each section is what one call ran.
# Not meant to be run as a whole; copy the section you need.

# ======================================================================== r04_run1.sh
# round-4 GPU call 1: parity suite, keypose bench line, eager kernel trace of the keypose step
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04a
python -m pytest tests -m gpu -q -s > gpurun_out/r04a/pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r04a/rc.txt
grep -E "passed|failed|error" gpurun_out/r04a/pytest.log | tail -5
python bench.py --skip-secondary --skip-cpu-baseline > gpurun_out/r04a/bench_kp.json 2> gpurun_out/r04a/bench_kp.err; echo "bench rc=$?" >> gpurun_out/r04a/rc.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/r04a/trace" -o kp -- python "$GRAFT_REPO_ROOT/bench.py" --skip-secondary --skip-cpu-baseline --no-graph --steps 10 --warmup 4 > "$GRAFT_REPO_ROOT/gpurun_out/r04a/trace.log" 2>&1
cd "$GRAFT_REPO_ROOT"
DB=$(find gpurun_out/r04a/trace -name '*.db' | head -1)
python profiles/summarize.py "$DB" > gpurun_out/r04a/kernel_trace_B64.txt 2>&1
rm -rf gpurun_out/r04a/trace
cat gpurun_out/r04a/rc.txt; head -c 600 gpurun_out/r04a/bench_kp.json; head -45 gpurun_out/r04a/kernel_trace_B64.txt

# ======================================================================== r04_run2.sh
# round-4 GPU call 2: parity suite, keypose bench + trace, A/B builds (fma_mix, SLP, libm sincos), per-layer convolution probe
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04b; mkdir -p $O
python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "passed|failed" $O/pytest.log | tail -3; grep -E "^FAILED" $O/pytest.log | head
python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp.json 2> $O/bench_kp.err; echo "bench rc=$?" >> $O/rc.txt
python bench.py --kernels-only > $O/kern_default.json 2>$O/kern_default.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace" -o kp -- python "$GRAFT_REPO_ROOT/bench.py" --skip-secondary --skip-cpu-baseline --no-graph --steps 10 --warmup 4 > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1 )
DB=$(find $O/trace -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/kernel_trace_B64.txt 2>&1; rm -rf $O/trace
python profiles/conv_layers_probe.py > $O/conv_layers.txt 2>&1
A3D_HIPCC_FLAGS="-DA3D_NO_FMA_MIX" python act3d-chained-diffuser_amd/build.py --force > /dev/null 2>&1
python bench.py --kernels-only > $O/kern_nomix.json 2>/dev/null
A3D_HIPCC_FLAGS="-fno-slp-vectorize" python act3d-chained-diffuser_amd/build.py --force > /dev/null 2>&1
python bench.py --kernels-only > $O/kern_noslp.json 2>/dev/null
A3D_HIPCC_FLAGS="-DA3D_LIBM_SINCOS" python act3d-chained-diffuser_amd/build.py --force > /dev/null 2>&1
python -m pytest tests/test_diffusion_gpu.py -q -s > $O/pytest_libm_sincos.log 2>&1
grep -E "passed|failed|state before" $O/pytest_libm_sincos.log | tail -8
cat $O/rc.txt; head -c 400 $O/bench_kp.json; echo; head -40 $O/kernel_trace_B64.txt; tail -3 $O/conv_layers.txt | head -c 1500
for f in default nomix noslp; do echo $f; python - <<P
import json
d=json.load(open("$O/kern_$f.json"))
print({k:(round(v.get("ms",0),4), v.get("mfma_util_executed")) for k,v in d.get("kernels",d).items() if isinstance(v,dict)})
P
done

# ======================================================================== r04_run3.sh
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04c; mkdir -p $O
python -m pytest tests/test_joint_gpu.py -q -s -k "joint_iteration" > $O/joint.log 2>&1; grep -E "parity|passed|failed" $O/joint.log | tail -12
python -m pytest tests/test_data_gpu.py -q -s -k "reference or matches" > $O/harness.log 2>&1; grep -E "parity|passed|failed|Error|assert" $O/harness.log | tail -20

# ======================================================================== r04_run4.sh
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04d; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -q -s -k "query_stream" > $O/qs.log 2>&1; grep -E "passed|failed|Error|error" $O/qs.log | tail -5; grep -E "parity.*fused vs|Assertion" $O/qs.log | tail -12
python -m pytest tests/test_act3d_gpu.py tests/test_engine_gpu.py tests/test_joint_gpu.py -q -s > $O/model.log 2>&1; grep -E "passed|failed" $O/model.log | tail -3; grep -E "^FAILED" $O/model.log | head
python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp.json 2> $O/bench_kp.err; echo "bench rc=$?"
A3D_OVERLAP_STREAMS=1 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp_overlap.json 2> $O/bench_kp_overlap.err
A3D_QS_FUSED=0 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp_noqs.json 2> /dev/null
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace" -o kp -- python "$GRAFT_REPO_ROOT/bench.py" --skip-secondary --skip-cpu-baseline --no-graph --steps 10 --warmup 4 > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1 )
DB=$(find $O/trace -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/kernel_trace_B64.txt 2>&1; rm -rf $O/trace
for f in bench_kp bench_kp_overlap bench_kp_noqs; do python - <<P
import json
try:
    d=json.load(open("$O/$f.json")); print("$f", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only"))
except Exception as e: print("$f", "failed", e)
P
done
head -30 $O/kernel_trace_B64.txt

# ======================================================================== r04_run5.sh
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04e; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -q -s -k "knn or query_stream" > $O/k.log 2>&1; grep -E "passed|failed" $O/k.log | tail -3; grep -E "^FAILED|Error" $O/k.log | head
python -m pytest tests/test_diffusion_gpu.py tests/test_act3d_gpu.py -q -s > $O/model.log 2>&1; grep -E "passed|failed" $O/model.log | tail -3; grep -E "^FAILED" $O/model.log | head
python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp.json 2> $O/bench_kp.err; echo "bench rc=$?"
A3D_QS_FUSED=0 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp_noqs.json 2> /dev/null
python bench_denoise.py --mode sample > $O/denoise.json 2> $O/denoise.err; tail -c 600 $O/denoise.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace" -o kp -- python "$GRAFT_REPO_ROOT/bench.py" --skip-secondary --skip-cpu-baseline --no-graph --steps 10 --warmup 4 > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1 )
DB=$(find $O/trace -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/kernel_trace_B64.txt 2>&1; rm -rf $O/trace
for f in bench_kp bench_kp_noqs; do python - <<P
import json
try:
    d=json.load(open("$O/$f.json")); print("$f", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only")); print({k:(round(v["ms"],4), round(v["frac"],4)) for k,v in d.get("kernels",{}).items()}, d.get("kernels_error"))
except Exception as e: print("$f", "failed", e)
P
done
grep -E "qs_|knn|sq_|dispatches" $O/kernel_trace_B64.txt | head -20

# ======================================================================== r04_run6.sh
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04f; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -q -s -k "conv1x1 or backbone or query_stream" > $O/k.log 2>&1; grep -E "passed|failed" $O/k.log | tail -3; grep -E "^FAILED|Error" $O/k.log | head
python profiles/conv1x1_layers_probe.py > $O/conv1x1_layers.txt 2>&1; tail -1 $O/conv1x1_layers.txt | head -c 300; echo
grep -E "^\{'cin'" $O/conv1x1_layers.txt | cut -c1-200
python profiles/conv1x1_probe.py > $O/conv1x1_probe.json 2>&1; tail -1 $O/conv1x1_probe.json
python -m pytest tests/test_diffusion_gpu.py tests/test_dropout_gpu.py -q -s > $O/diff.log 2>&1; grep -E "passed|failed" $O/diff.log | tail -3; grep -E "^FAILED" $O/diff.log | head; grep "script shape" $O/diff.log
python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp.json 2> $O/bench_kp.err; echo "bench rc=$?"
A3D_FUSED_CONV1X1=0 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp_miopen.json 2> /dev/null
python bench_denoise.py --mode sample > $O/denoise.json 2> $O/denoise.err
for f in bench_kp bench_kp_miopen; do python - <<P
import json
try:
    d=json.load(open("$O/$f.json")); print("$f", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only"))
except Exception as e: print("$f", "failed", e)
P
done
python - <<P
import json
d=json.load(open("$O/denoise.json")); print("denoise ms/step", d.get("ms_per_denoise_step"))
P

# ======================================================================== r04_run7.sh
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04g; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -q -s -k "conv1x1 or backbone or query_stream or attn_block" > $O/k.log 2>&1; grep -E "passed|failed" $O/k.log | tail -3; grep -E "^FAILED|Error" $O/k.log | head
python profiles/conv1x1_layers_probe.py > $O/conv1x1_layers.txt 2>&1; tail -1 $O/conv1x1_layers.txt | head -c 200; echo
grep -E "^\{'cin'" $O/conv1x1_layers.txt | grep -v "nan" | cut -c1-200
python profiles/conv1x1_probe.py > $O/conv1x1_probe.json 2>&1; tail -1 $O/conv1x1_probe.json
python -m pytest tests/test_diffusion_gpu.py -q > $O/diff.log 2>&1; grep -E "passed|failed" $O/diff.log | tail -2
python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp.json 2> $O/bench_kp.err; echo "bench rc=$?"
A3D_QS_FUSED=0 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp_noqs.json 2> /dev/null
python bench_denoise.py --mode sample > $O/denoise.json 2> $O/denoise.err
for f in bench_kp bench_kp_noqs; do python - <<P
import json
try:
    d=json.load(open("$O/$f.json")); print("$f", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only"))
except Exception as e: print("$f", "failed", e)
P
done
python - <<P
import json
d=json.load(open("$O/denoise.json")); print("denoise ms/step", d.get("ms_per_denoise_step"))
P

# ======================================================================== r04_run9.sh
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04h; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_act3d_gpu.py -q -s -k "projection or attn_block or act3d or query_stream" > $O/k.log 2>&1; grep -E "passed|failed" $O/k.log | tail -3; grep -E "^FAILED|Error" $O/k.log | head
python bench.py --kernels-only > $O/kern_res.json 2>/dev/null
A3D_PROJ_RES=0 python bench.py --kernels-only > $O/kern_nores.json 2>/dev/null
python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp.json 2> $O/bench_kp.err
A3D_PROJ_RES=0 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp_nores.json 2> /dev/null
for f in bench_kp bench_kp_nores; do python - <<P
import json
try:
    d=json.load(open("$O/$f.json")); print("$f", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only"))
except Exception as e: print("$f", "failed", e)
P
done
for f in res nores; do python - <<P
import json
d=json.load(open("$O/kern_$f.json"))["kernels"]; print("$f", {k:round(v["ms"],4) for k,v in d.items() if isinstance(v,dict)})
P
done

# ======================================================================== r04_run10.sh
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04i; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -s -k "conv3x3 or backbone" > $O/k.log 2>&1; grep -E "passed|failed" $O/k.log | tail -3; grep -E "^FAILED|Error|max err" $O/k.log | head -20
grep "parity. conv3x3\|parity. backbone" $O/k.log | head -30
timeout 600 python profiles/conv3x3_probe.py > $O/conv3x3_probe.json 2> $O/probe.err; tail -3 $O/probe.err; cat $O/conv3x3_probe.json
timeout 600 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp.json 2> $O/bench_kp.err
A3D_FUSED_CONV3X3=0 timeout 600 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp_no3.json 2> /dev/null
for f in bench_kp bench_kp_no3; do python - <<P
import json
try:
    d=json.load(open("$O/$f.json")); print("$f", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only"))
except Exception as e: print("$f", "failed", e)
P
done

# ======================================================================== r04_final.sh
# round-4 evidence: parity report of the full GPU suite, the bench line, eager kernel traces (keypose, diffusion training),
# counter passes (keypose kernels at B = 64; the diffusion training attention micro-benchmark; the diffusion training step)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04z; mkdir -p $O
python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED" $O/pytest.log | head
grep -E "^\.*\[parity\]|^\[parity\]" $O/pytest.log | sed 's/^\.*//' > $O/parity_report.txt
profiles/pmc_json_cmd.sh $O/pmc_B64.json 64 python "$GRAFT_REPO_ROOT/bench.py" --kernels-only --batch 64
profiles/pmc_json_cmd.sh $O/pmc_diffusion_attn_B22_L50.json 22 python "$GRAFT_REPO_ROOT/bench_denoise.py" --mode attn --batch 22 --horizon 50
profiles/pmc_json_cmd.sh $O/pmc_diffusion_attn_B64_L16.json 64 python "$GRAFT_REPO_ROOT/bench_denoise.py" --mode attn --batch 64 --horizon 16
cp $O/pmc_B64.json profiles/r04_pmc_B64.json; cp $O/pmc_diffusion_attn_B22_L50.json profiles/r04_pmc_diffusion_attn_B22_L50.json; cp $O/pmc_diffusion_attn_B64_L16.json profiles/r04_pmc_diffusion_attn_B64_L16.json
python bench.py > $O/bench_B64.json 2> $O/bench_B64.err; echo "bench rc=$?" >> $O/rc.txt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace" -o kp -- python "$GRAFT_REPO_ROOT/bench.py" --skip-secondary --skip-cpu-baseline --no-graph --steps 10 --warmup 4 > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1 )
DB=$(find $O/trace -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/kernel_trace_B64.txt 2>&1; rm -rf $O/trace
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace2" -o dt -- python "$GRAFT_REPO_ROOT/bench_denoise.py" --mode train --no-graph > "$GRAFT_REPO_ROOT/$O/trace2.log" 2>&1 )
DB=$(find $O/trace2 -name '*.db' | head -1); python profiles/summarize.py "$DB" 4 12 > $O/kernel_trace_diffusion_train.txt 2>&1; python profiles/trace_summary.py "$DB" > $O/kernel_totals_diffusion_train.txt 2>&1; rm -rf $O/trace2
cat $O/rc.txt; head -c 500 $O/bench_B64.json; echo; head -12 $O/kernel_trace_B64.txt; head -8 $O/kernel_trace_diffusion_train.txt

# ======================================================================== r04_final2.sh
# round-4 evidence refresh after conv3x3.hip / the hardware bf16 rounding in conv1x1.hip: parity report of the full GPU suite, the
# bench line, the eager kernel trace of the keypose step, the convolution probes (the counter passes of r04_final.sh cover the
# attention / single-query / k-NN / projection kernels, which did not change)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04y; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head
grep -E "^\.*\[parity\]|^\[parity\]" $O/pytest.log | sed 's/^\.*//' > $O/parity_report.txt
timeout 300 python profiles/conv3x3_probe.py > $O/conv3x3_probe.json 2> $O/probe3.err; cat $O/conv3x3_probe.json
timeout 300 python profiles/conv1x1_probe.py > $O/conv1x1_probe.json 2> $O/probe1.err; cat $O/conv1x1_probe.json
timeout 900 python bench.py > $O/bench_B64.json 2> $O/bench_B64.err; echo "bench rc=$?" >> $O/rc.txt
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace" -o kp -- python "$GRAFT_REPO_ROOT/bench.py" --skip-secondary --skip-cpu-baseline --no-graph --steps 10 --warmup 4 > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1 )
DB=$(find $O/trace -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/kernel_trace_B64.txt 2>&1; rm -rf $O/trace
cat $O/rc.txt; head -c 400 $O/bench_B64.json; echo; head -30 $O/kernel_trace_B64.txt | cut -c1-150

# ======================================================================== r04_final3.sh
# after the 64 -> 64 output-channel split and the unrolled 32 -> 64 instance of conv3x3.hip: convolution tests, the probe with and
# without the split, the bench line, the keypose kernel trace
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04x; mkdir -p $O
timeout 200 python -m pytest tests/test_kernels_gpu.py -q -s -k "conv3x3 or backbone or conv1x1" > $O/k.log 2>&1; echo "pytest rc=$?" > $O/rc.txt; grep -E "passed|failed" $O/k.log | tail -2; grep -E "^FAILED|^ERROR|max err" $O/k.log | head
timeout 120 python profiles/conv3x3_probe.py > $O/conv3x3_probe.json 2> $O/probe3.err; cat $O/conv3x3_probe.json
A3D_C3_SPLIT=0 timeout 120 python profiles/conv3x3_probe.py > $O/conv3x3_probe_nosplit.json 2> /dev/null; cat $O/conv3x3_probe_nosplit.json
timeout 400 python bench.py > $O/bench_B64.json 2> $O/bench_B64.err; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt; head -c 300 $O/bench_B64.json; echo
( cd /tmp && export TMPDIR=/tmp && timeout 150 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace" -o kp -- python "$GRAFT_REPO_ROOT/bench.py" --skip-secondary --skip-cpu-baseline --no-graph --steps 10 --warmup 4 > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1 )
DB=$(find $O/trace -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/kernel_trace_B64.txt 2>&1; rm -rf $O/trace
head -12 $O/kernel_trace_B64.txt | cut -c1-150; grep conv3x3 $O/kernel_trace_B64.txt | cut -c1-150
