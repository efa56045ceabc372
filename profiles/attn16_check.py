#!/usr/bin/env python3
"""A/B of the two attention-core families (split-fp16 attention16.hip vs split-bf16 attention.hip) on one MI355X:
error of O, dQ, dK, dV against a float64 torch evaluation of the same formulas on the device (at mild and at sharp logits,
|s| ~ 100), and kernel timings at the bench shape (B = 64, Lq = 333, S = 4097, E = 60, H = 4).
    python profiles/attn16_check.py [--batch 64] > gpurun_out/r03/attn16_check.txt"""
import argparse
import importlib
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def reference(q_pre, k_pre, v_pre, dO, H, S_valid=None):
    q = q_pre.double().requires_grad_()
    k = k_pre.double().requires_grad_()
    v = v_pre.double().requires_grad_()
    B, Lq, E = q.shape
    S = k.shape[1]
    d = E // H
    qh = (q * d ** -0.5).view(B, Lq, H, d).transpose(1, 2)
    kh = k.view(B, S, H, d).transpose(1, 2)
    vh = v.view(B, S, H, d).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2)
    p = torch.softmax(s, -1)
    o = (p @ vh).transpose(1, 2).reshape(B, Lq, E)
    o.backward(dO.double())
    return o.detach(), q.grad, k.grad, v.grad, s.detach().abs().max().item()


def run_family(O, fam, q_pre, k_pre, v_pre, dO, H, time_it=False):
    B, Lq, E = q_pre.shape
    S = k_pre.shape[1]
    dev = q_pre.device
    op = O.attn_operands16 if fam == "f16" else O.attn_operands
    qc, kc, vc = q_pre.reshape(B * Lq, E).contiguous(), k_pre.reshape(B * S, E).contiguous(), v_pre.reshape(B * S, E).contiguous()
    Qs, Ks, Vt, Lqp, Sp, scale, freq, extra = op(qc.data_ptr(), E, kc.data_ptr(), E, vc.data_ptr(), E, None, None, B, Lq, S, E, H,
                                                 dev, need_bwd=True)
    ns = O.pick_nsplit(B, H, Lqp, Sp)
    Oo, LSE = O.attn_core_fwd(Qs, Ks, Vt, None, B, H, Lq, Lqp, S, Sp, ns, need_bwd=True)
    dQp, dK, dV = O.attn_core_bwd(Qs, Ks, Vt, None, Oo, dO, LSE, B, H, Lq, Lqp, S, Sp, ns, extra=extra)
    dq = torch.empty(B * Lq, E, device=dev)
    dk = torch.empty(B * S, E, device=dev)
    dv = torch.empty(B * S, E, device=dev)
    O.rope_merge(dQp, ns, None, freq, scale, dq.data_ptr(), E, B, Lq, Lqp, E, H)
    O.rope_merge(dK, 1, None, freq, 1.0, dk.data_ptr(), E, B, S, Sp, E, H)
    O.rope_merge(dV, 1, None, freq, 1.0, dv.data_ptr(), E, B, S, Sp, E, H)
    times = None
    if time_it:
        def tm(fn, iters=20):
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
        times = {"fwd_ms": tm(lambda: O.attn_core_fwd(Qs, Ks, Vt, None, B, H, Lq, Lqp, S, Sp, ns)),
                 "bwd_ms": tm(lambda: O.attn_core_bwd(Qs, Ks, Vt, None, Oo, dO, LSE, B, H, Lq, Lqp, S, Sp, ns, extra=extra)),
                 "nsplit": ns}
    return Oo, dq.view(B, Lq, E), dk.view(B, S, E), dv.view(B, S, E), times


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--time-only", action="store_true")
    ap.add_argument("--family", default=None)
    args = ap.parse_args()
    a3d = importlib.import_module("act3d-chained-diffuser_amd")
    O = a3d.ops
    dev = torch.device("cuda:0")
    H, E = 4, 60
    out = {}
    cases = {} if args.time_only else {"mild": (2, 333, 4097, 1.0), "sharp": (2, 333, 4097, 6.0), "short": (3, 37, 131, 3.0)}
    for tag, (B, Lq, S, gain) in cases.items():
        g = torch.Generator().manual_seed(7)
        q_pre = (torch.randn(B, Lq, E, generator=g) * gain).to(dev)
        k_pre = (torch.randn(B, S, E, generator=g) * gain).to(dev)
        v_pre = torch.randn(B, S, E, generator=g).to(dev)
        dO = (torch.randn(B, Lq, E, generator=g) * 1e-3).to(dev)
        ro, rq, rk, rv, smax = reference(q_pre, k_pre, v_pre, dO, H)
        for fam in ("f16", "bf16x3"):
            o, dq, dk, dv, _ = run_family(O, fam, q_pre, k_pre, v_pre, dO, H)
            rec = {"max|s|": smax}
            for name, got, ref in (("O", o, ro), ("dQ", dq, rq), ("dK", dk, rk), ("dV", dv, rv)):
                err = (got.double() - ref).abs().max().item()
                sc = max(ref.abs().max().item(), 1e-30)
                rec[name] = {"max_abs_err": err, "ref_absmax": sc, "err_over_scale": err / sc,
                             "rel_l2": ((got.double() - ref).norm() / ref.norm()).item()}
            out[f"{tag}/{fam}"] = rec
            print(tag, fam, json.dumps(rec), flush=True)
    B, Lq, S = args.batch, 333, 4097
    g = torch.Generator().manual_seed(1)
    q_pre = torch.randn(B, Lq, E, generator=g).to(dev)
    k_pre = torch.randn(B, S, E, generator=g).to(dev)
    v_pre = torch.randn(B, S, E, generator=g).to(dev)
    dO = torch.randn(B, Lq, E, generator=g).to(dev)
    for fam in ([args.family] if args.family else ["f16", "bf16x3"]):
        *_, times = run_family(O, fam, q_pre, k_pre, v_pre, dO, H, time_it=True)
        f_fwd, f_bwd = 4.0 * Lq * S * E * B, 10.0 * Lq * S * E * B
        times["fwd_frac_of_2.5PF"] = f_fwd / (times["fwd_ms"] * 1e-3) / 2.5e15
        times["bwd_frac_of_2.5PF"] = f_bwd / (times["bwd_ms"] * 1e-3) / 2.5e15
        out[f"time/{fam}"] = times
        print("time", fam, json.dumps(times), flush=True)


if __name__ == "__main__":
    main()
