cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r06
timeout 300 python profiles/r06_fpn_diag.py 2>&1 | grep -v Warning | tee gpurun_out/r06/fpn_diag.txt
