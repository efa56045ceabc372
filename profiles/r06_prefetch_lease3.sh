#!/bin/bash
# prefetched backbone: where to fork it (A3D_PREFETCH_FORK=start|tokens -- a knob of the measured tree, removed afterwards: no difference), stream priorities (A3D_PREFETCH_HIPRIO=main|side)
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r06/p; mkdir -p $O
for v in "${@:-tokens}"; do
  A3D_PREFETCH_BACKBONE=1 A3D_PREFETCH_FORK=$v timeout 600 python bench.py --skip-secondary --skip-cpu-baseline --steps 20 --warmup 5 > $O/bench_prefetch1_fork$v.json 2> $O/bench_prefetch1_fork$v.err
  python -c "
import json; d=json.load(open('$O/bench_prefetch1_fork$v.json')); print('A3D_PREFETCH_FORK=$v', round(d['value'],1), round(d['ms_per_step'],3), d['config'].get('backbone_prefetch'), d['config'].get('graph_capture_error'))" || tail -5 $O/bench_prefetch1_fork$v.err
done
