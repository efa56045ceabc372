#!/usr/bin/env python3
"""A/B of two builds of libact3d_hip.so on the ghost-attention shape (B = 64, Lq = 333, S = 4097, E = 60, H = 4): per-launch
timings of the training forward (two-part P), the gradient-free forward and the backward, plus SHA-256 digests of O, LSE,
dQ (partials), dK, dV on fixed seeded inputs -- "identical bits" claims between builds are checked by comparing the digests.
    A3D_LIB=libact3d_hip_base.so python profiles/attn_ab.py   (one JSON line per run; the build under test is named in it)"""
import argparse
import hashlib
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def digest(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


def tm(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--lq", type=int, default=333)
    ap.add_argument("--keys", type=int, default=4097)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--gain", type=float, default=1.0)
    ap.add_argument("--nsplit", type=int, default=0, help="key splits of the forward / dQ launches (0: ops.pick_nsplit)")
    ap.add_argument("--check", action="store_true", help="errors of O, dQ, dK, dV against a float64 evaluation (B = 2; mild and sharp logits)")
    args = ap.parse_args()
    a3d = importlib.import_module("act3d-chained-diffuser_amd")
    O = a3d.ops
    dev = torch.device("cuda:0")
    H, E, B, Lq, S = 4, 60, args.batch, args.lq, args.keys
    g = torch.Generator().manual_seed(11)
    q_pre = (torch.randn(B, Lq, E, generator=g) * args.gain).to(dev)
    k_pre = (torch.randn(B, S, E, generator=g) * args.gain).to(dev)
    v_pre = torch.randn(B, S, E, generator=g).to(dev)
    dO = torch.randn(B, Lq, E, generator=g).to(dev)
    dO[:, ::7] *= 1e-4                                   # rows of very different scale: the sorted packs have several exponents
    qc, kc, vc = q_pre.reshape(B * Lq, E), k_pre.reshape(B * S, E), v_pre.reshape(B * S, E)
    Qs, Ks, Vt, Lqp, Sp, scale, freq, extra = O.attn_operands16(qc.data_ptr(), E, kc.data_ptr(), E, vc.data_ptr(), E, None, None,
                                                                B, Lq, S, E, H, dev, need_bwd=True)
    ns = args.nsplit or O.pick_nsplit(B, H, Lqp, Sp)
    Oo, LSE = O.attn_core_fwd(Qs, Ks, Vt, None, B, H, Lq, Lqp, S, Sp, ns, need_bwd=True)
    On, _ = O.attn_core_fwd(Qs, Ks, Vt, None, B, H, Lq, Lqp, S, Sp, ns, nograd=True)
    dQp, dK, dV = O.attn_core_bwd(Qs, Ks, Vt, None, Oo, dO, LSE, B, H, Lq, Lqp, S, Sp, ns, extra=extra)
    torch.cuda.synchronize()
    rec = {"lib": os.environ.get("A3D_LIB", "libact3d_hip.so"), "shape": [B, Lq, S, E, H], "nsplit": ns,
           "sha": {"O": digest(Oo), "O_nograd": digest(On), "LSE": digest(LSE), "dQp": digest(dQp), "dK": digest(dK), "dV": digest(dV)},
           "fwd_train_ms": tm(lambda: O.attn_core_fwd(Qs, Ks, Vt, None, B, H, Lq, Lqp, S, Sp, ns, need_bwd=True), args.iters),
           "fwd_nograd_ms": tm(lambda: O.attn_core_fwd(Qs, Ks, Vt, None, B, H, Lq, Lqp, S, Sp, ns, nograd=True), args.iters),
           "bwd_ms": tm(lambda: O.attn_core_bwd(Qs, Ks, Vt, None, Oo, dO, LSE, B, H, Lq, Lqp, S, Sp, ns, extra=extra), args.iters)}
    if args.check:
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        import attn16_check as AC
        for tag, gain in (("mild", 1.0), ("sharp", 6.0)):
            gg = torch.Generator().manual_seed(7)
            qq = (torch.randn(2, Lq, E, generator=gg) * gain).to(dev)
            kk = (torch.randn(2, S, E, generator=gg) * gain).to(dev)
            vv = torch.randn(2, S, E, generator=gg).to(dev)
            dd = (torch.randn(2, Lq, E, generator=gg) * 1e-3).to(dev)
            ref = AC.reference(qq, kk, vv, dd, H)
            got = AC.run_family(O, "f16", qq, kk, vv, dd, H)
            rec["err_" + tag] = {n: float((g_.double() - r_).abs().max() / r_.abs().max()) for n, g_, r_ in zip(("O", "dQ", "dK", "dV"), got[:4], ref[:4])}
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
