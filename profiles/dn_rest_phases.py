#!/usr/bin/env python3
"""Phase timeline of one dn_rest workgroup (csrc/denoise.hip, A3D_DN_PROF=1): where the 70 us of the per-sample layer
remainder go.  usage (GPU box): A3D_DN_PROF=1 python profiles/dn_rest_phases.py"""
import ctypes
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("A3D_DN_PROF", "1")
a3d = importlib.import_module("act3d-chained-diffuser_amd")
import bench_denoise as BD  # noqa: E402

dev = torch.device("cuda:0")
r = BD.sampling_bench(a3d, dev, 64, 16, 3, reps=1, graph=False)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 18)()
a3d.lib.call("a3d_dbg_dn_prof", ctypes.cast(buf, ctypes.c_void_p).value)
t = [buf[i] for i in range(18)]
# dn_rest_loop_kernel: marks 0, 1, 2, then one per operation (3 .. 15), 17 at the end (the straight-line kernel of round 3 is gone)
names = ["stage vectors + warm-up of the weights", "load x rows + pads + combine key splits", "linear c_out (120->120)", "add+LN",
         "AdaLN (self)", "linear q|k (120->240)", "linear v (120->120)", "RoPE", "16x16 self-attention", "linear s_out (120->120)",
         "add+LN", "AdaLN (ffn)", "linear ffn1 (120->480, relu)", "linear ffn2 (480->120)", "add+LN", "store"]
marks = list(range(16)) + [17]
out = {"kernel": "dn_rest_loop_kernel", "eager_ms_per_denoise_step": r.get("ms_per_denoise_step"),
       "total_us": (t[17] - t[0]) * 0.01,
       "phases_us": {f"{i:02d} {names[i]}": round((t[marks[i + 1]] - t[marks[i]]) * 0.01, 2) for i in range(len(marks) - 1)}}
print(json.dumps(out, indent=1))
