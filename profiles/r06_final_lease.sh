#!/bin/bash
# Round-6 evidence on the final tree, one gpurun call: full GPU test suite + parity report, default bench line, counter passes, kernel traces
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r06; mkdir -p $O
bash profiles/r06_campaign.sh parity 2>&1 | tail -8
bash profiles/r06_campaign.sh bench 2>&1 | tail -c 400
bash profiles/r06_campaign.sh pmc > $O/call_pmc.log 2>&1; tail -4 $O/call_pmc.log | cut -c1-600
bash profiles/r06_campaign.sh pmc5 > $O/call_pmc5.log 2>&1; tail -1 $O/call_pmc5.log | cut -c1-400
bash profiles/r06_campaign.sh pmcdt > $O/call_pmcdt.log 2>&1; tail -2 $O/call_pmcdt.log
bash profiles/r06_campaign.sh trace > $O/call_trace.log 2>&1; tail -2 $O/call_trace.log
A3D_LIB=libact3d_hip.so timeout 300 python profiles/attn_ab.py --check > $O/attn_ab_final.json 2>/dev/null; cut -c1-400 $O/attn_ab_final.json
