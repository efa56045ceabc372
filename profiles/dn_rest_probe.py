#!/usr/bin/env python3
"""Stage costs of a3d_dn_rest (per-sample remainder of a denoiser layer): times the kernel with the self-attention block
and / or the FFN block switched off (NULL weights), B = 64 samples, L = 16, E = 120."""
import ctypes
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
a3d = importlib.import_module("act3d-chained-diffuser_amd")
Lb = a3d.lib
Lb.load()
dev = torch.device("cuda:0")
B, L, E, H, D, ns = 64, 16, 120, 8, 9, 2
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)
x, traj = r(B, L, E), r(B, L, D)
ws = torch.rand(ns * B * H * 16 * 17, generator=g).to(dev)
W = dict(c_out_w=r(E, E) / 11, c_out_b=r(E), c_ln_g=r(E), c_ln_b=r(E), sem=r(L, E), s_mod=r(2 * E) * 0.1, s_in_w=r(3 * E, E) / 11,
         s_in_b=r(3 * E), s_out_w=r(E, E) / 11, s_out_b=r(E), s_ln_g=r(E), s_ln_b=r(E), freq=a3d.ops.rope_freq(E, dev),
         f_mod=r(2 * E) * 0.1, f_w1=r(4 * E, E) / 11, f_b1=r(4 * E), f_w2=r(E, 4 * E) / 22, f_b2=r(E), f_ln_g=r(E), f_ln_b=r(E))
out = torch.empty_like(x)
for name, drop in (("cross-out + LN only", ("s_in_w", "f_w1")), ("+ self-attention", ("f_w1",)), ("+ FFN (no self-attn)", ("s_in_w",)),
                   ("full", ())):
    kw = {k: (None if k in drop else v.data_ptr()) for k, v in W.items()}
    p = Lb.DnRestParams(kmask=None, F=4 * E, **kw)

    def run():
        Lb.call("a3d_dn_rest", x.data_ptr(), traj.data_ptr(), D, ws.data_ptr(), ctypes.byref(p), out.data_ptr(), B, L, E, H, ns,
                Lb.stream())
    for _ in range(5):
        run()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    st.record()
    for _ in range(50):
        run()
    en.record()
    torch.cuda.synchronize()
    print(f"{name:28s} {st.elapsed_time(en) / 50 * 1e3:8.1f} us")
