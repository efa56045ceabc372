#!/bin/bash
# deep-layer 1x1 GEMM (conv1x1_deep.hip): kernel tests + per-layer A/B against MIOpen (profiles/conv1x1_layers_probe.py)
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r06/${1:-d}; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv1x1" 2>&1 | tail -4 | tee $O/tests.txt
timeout 600 python profiles/conv1x1_layers_probe.py 2> $O/probe.err | tail -1 > $O/conv1x1_layers.json
python - <<P
import json
d=json.load(open("$O/conv1x1_layers.json"))
print(d["total_ms"])
for l in d["layers"]: print("%5d -> %5d hw %3d x%d %-6s miopen %6.1f miopen+bn %6.1f fused %6.1f" % (l["cin"], l["cout"], l["hw"], l["n"], l["pos"], l["miopen_us"], l["miopen_bn_us"], l["fused_us"]))
P
