"""Phase breakdown of sq_bwd_kernel (single-query key pass, backward) at the bench shape (B = 64, S = 4097, E = 60, H = 4):
arms a3d_dbg_sq_prof, runs one a3d_sq_attn_bwd, prints the 100 MHz timestamps of workgroup (0, 0) as microseconds per phase of
its last tile plus the prologue and the whole kernel.  usage (GPU box): python profiles/sq_bwd_phases.py"""
import ctypes
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
a3d = importlib.import_module("act3d-chained-diffuser_amd")
Lb = a3d.ops.L
lib = Lb.load()
dev = torch.device("cuda:0")
B, S, E, H = 64, 4097, 60, 4
g = torch.Generator().manual_seed(0)
x = torch.randn(B, S, E, generator=g).to(dev)
xyz = torch.rand(B, S, 3, generator=g).to(dev)
w = (torch.randn(3 * E, E, generator=g) / E ** 0.5).to(dev)
bb = torch.zeros(3 * E, device=dev)
freq = a3d.ops.rope_freq(E, dev)
qrot = torch.randn(B, H, 1, 16, device=dev)
nsplit = max(1, min((S + 63) // 64, 1024 // B))
f4 = 4
ws = torch.empty((lib.a3d_sq_fwd_ws_floats(B, H, E, nsplit),), device=dev)
xbar, lse = torch.empty((B, H, E), device=dev), torch.empty((B, H), device=dev)
wp, bp = w.data_ptr(), bb.data_ptr()
Lb.call("a3d_sq_attn_fwd", x.data_ptr(), xyz.data_ptr(), wp + E * E * f4, E, bp + E * f4, None, E, None, qrot.data_ptr(), freq.data_ptr(),
        ws.data_ptr(), xbar.data_ptr(), lse.data_ptr(), None, B, S, E, H, nsplit, Lb.stream())
wsb = torch.zeros((lib.a3d_sq_bwd_ws_floats(B, H, E, nsplit),), device=dev)
wsb[:B * H * E] = torch.randn(B * H * E, generator=g).to(dev)          # dxbar (the caller's value-projection backward wrote it)
dX = torch.empty((B, S, E), device=dev)
dqp = torch.empty((nsplit, B, H, 1, 16), device=dev)
gW, gb = torch.zeros(3 * E, E, device=dev), torch.zeros(3 * E, device=dev)


def bwd():
    Lb.call("a3d_sq_attn_bwd", x.data_ptr(), xyz.data_ptr(), wp + E * E * f4, E, bp + E * f4, None, E, qrot.data_ptr(), freq.data_ptr(),
            xbar.data_ptr(), lse.data_ptr(), None, wsb.data_ptr(), dX.data_ptr(), dqp.data_ptr(), gW.data_ptr() + E * E * f4, E,
            gb.data_ptr() + E * f4, None, E, None, B, S, E, H, nsplit, Lb.stream())


for _ in range(3):
    bwd()
torch.cuda.synchronize()
Lb.call("a3d_dbg_sq_prof", 1, None)
bwd()
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 12)()
Lb.call("a3d_dbg_sq_prof", 0, ctypes.cast(buf, ctypes.c_void_p).value)
t = [buf[i] for i in range(12)]
names = ["prologue (weights, query, dxbar; first rows in flight)", "rows -> LDS + barrier", "key projection + RoPE", "scores + dp + barrier",
         "p, ds + barrier", "rotated-query gradient + barrier", "inverse rotation + barrier", "dX GEMM + stores", "dW GEMM + barrier"]
out = {"shape": {"B": B, "S": S, "E": E, "H": H, "nsplit": nsplit}, "tiles_of_workgroup_0": t[11], "kernel_us_workgroup_0": (t[10] - t[0]) * 0.01,
       "prologue_us": (t[1] - t[0]) * 0.01,
       "last_tile_phases_us": {f"{i} {names[i]}": round((t[i + 1] - t[i]) * 0.01, 2) for i in range(1, 9)}}
print(json.dumps(out, indent=1))
