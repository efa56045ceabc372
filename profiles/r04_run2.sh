#!/bin/bash
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
