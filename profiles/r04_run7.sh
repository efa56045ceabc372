#!/bin/bash
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
