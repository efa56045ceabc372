cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r06/c; mkdir -p $O; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv3x3 or fpn" 2>&1 | tail -15 | tee $O/tests_fpn.txt
timeout 600 python bench.py --skip-secondary --skip-cpu-baseline --steps 20 --warmup 5 > $O/bench_sparse.json 2> $O/bench_sparse.err; python -c "
import json; d=json.load(open('$O/bench_sparse.json')); print('sparse dgrad', d['value'], d['ms_per_step'])"
A3D_FPN_SPARSE_DGRAD=0 timeout 600 python bench.py --skip-secondary --skip-cpu-baseline --steps 20 --warmup 5 > $O/bench_dense.json 2> $O/bench_dense.err; python -c "
import json; d=json.load(open('$O/bench_dense.json')); print('dense dgrad', d['value'], d['ms_per_step'])"
tail -3 $O/bench_sparse.err
