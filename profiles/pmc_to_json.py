#!/usr/bin/env python3
"""Folds the rocprofv3 --pmc passes of profiles/run_pmc.sh into one JSON keyed by bench.py's kernel names.

usage: python profiles/pmc_to_json.py <pmc-dir> <batch> > profiles/rNN_pmc_B<batch>.json
HBM traffic per launch = FETCH_SIZE * 1024 * 2 (gfx950 under-reports wide coalesced reads by 2x, MI355X_MICROARCH.md
"HBM") + WRITE_SIZE * 1024, summed over the kernels that make up one bench.py "launch" (attn_bwd = prep + dq + dkv)."""
import collections
import csv
import glob
import json
import os
import re
import sys

GROUPS = {
    # (the split-bf16 A/B family has its own groups since round 6: bench.py --kernels-only times both families in one run)
    "attn_fwd": ["attn16_fwd_kernel", "attn16_combine_kernel"],
    "attn_bwd": ["attn16_bwd_prep_kernel", "attn16_bwd_dq_kernel", "attn16_bwd_dkv_kernel"],
    "attn_fwd_bf16x3": ["attn_fwd_kernel", "attn_combine_kernel"],
    "attn_bwd_bf16x3": ["attn_bwd_prep_bf16_kernel", "attn_bwd_dq_bf16_kernel", "attn_bwd_dkv_bf16_kernel"],
    "kv_proj_rope": ["proj_rope_split_kernel"],
    "sq_fwd": ["sq_fwd_kernel", "sqw_fwd_kernel", "sq_combine_kernel"],
    "sq_bwd": ["sq_bwd_kernel", "sqw_bwd_kernel"],
    "dn_cross": ["dn_cross_kernel"],
    "dn_rest": ["dn_rest_loop_kernel"],
    "dn_persist": ["dn_persist_kernel"],
    "knn_topk": ["knn_dist_hist_kernel", "knn_select_sort_kernel"],
    "bn_stats": ["bn_stats_kernel"],
    "attn8_fwd": ["attn8_fwd_kernel", "attn8_amax_kernel", "attn8_pack_kernel"],
    "conv1x1_deep": ["conv1x1_deep_kernel"],
}


def means(pmc_dir, name):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(pmc_dir, name, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                # by base name: template instances of one kernel (attn16_fwd_kernel<false, 2, 3, true> / <false, 2, 2, true>) are averaged
                # together over their dispatches, not added up as if they were different kernels of one launch
                k = re.split(r"[<(]", row.get("Kernel_Name", ""))[0].strip()
                agg[(k, row["Counter_Name"])][0] += 1
                agg[(k, row["Counter_Name"])][1] += float(row["Counter_Value"])
    return {k: v[1] / v[0] for k, v in agg.items()}


def main():
    pmc_dir, batch = sys.argv[1], int(sys.argv[2])
    fetch, write, derived = means(pmc_dir, "fetch"), means(pmc_dir, "write"), means(pmc_dir, "derived")
    out = {"batch": batch, "source": "rocprofv3 --pmc passes of profiles/run_pmc.sh (FETCH_SIZE, WRITE_SIZE, MfmaUtil/VALUBusy), "
                                     "bench.py --kernels-only", "kernels": {}}
    for key, kernels in GROUPS.items():
        fb = wb = 0.0
        util = {}
        found = False
        for kn in kernels:
            for (name, ctr), v in fetch.items():
                if name.endswith(kn) and ctr == "FETCH_SIZE":
                    fb += v * 1024 * 2
                    found = True
            for (name, ctr), v in write.items():
                if name.endswith(kn) and ctr == "WRITE_SIZE":
                    wb += v * 1024
            for (name, ctr), v in derived.items():
                if name.endswith(kn) and ctr in ("MfmaUtil", "VALUBusy"):
                    util.setdefault(kn, {})[ctr] = round(v, 1)
        if found:
            out["kernels"][key] = {"hbm_read_bytes": fb, "hbm_write_bytes": wb, "hbm_bytes": fb + wb, "pmc": util}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
