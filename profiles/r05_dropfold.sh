#!/bin/bash
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
