#!/usr/bin/env python3
"""Localises errors of the fp8 attention forward on exactly representable operands (development aid)."""
import importlib
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
a3d = importlib.import_module("act3d-chained-diffuser_amd")
import test_attn8_gpu as T  # noqa: E402

L = a3d.lib
dev = torch.device("cuda:0")


def run(Qr, Kr, Vp, kmask, B, H, Lq, S, nsplit=1):
    Lqp, Sp = T._pad(Lq, 64), T._pad(S, 64)
    O = torch.full((B, Lq, H * 15), float("nan"), device=dev)
    LSE = torch.full((B, H, Lqp), float("nan"), device=dev)
    ws = torch.empty(nsplit * B * H * Lqp * 18, device=dev) if nsplit > 1 else None
    ops8 = torch.zeros(L.load().a3d_attn8_operand_bytes(B, H, Sp), device=dev, dtype=torch.uint8)
    L.call("a3d_attn8_fwd", Qr.data_ptr(), Kr.data_ptr(), Vp.data_ptr(), ops8.data_ptr(), None if kmask is None else kmask.data_ptr(),
           O.data_ptr(), LSE.data_ptr(), None if ws is None else ws.data_ptr(), B, H, Lq, Lqp, S, Sp, nsplit, L.stream())
    torch.cuda.synchronize()
    return O, LSE, ops8


def report(tag, O, o_ref, LSE, lse_ref, B, H, Lq):
    d = (O.double().cpu() - o_ref).abs().view(B, Lq, H, 15)
    print(f"{tag}: O max err {d.max().item():.3e} (nan: {torch.isnan(O).sum().item()}), LSE err "
          f"{(LSE[:, :, :Lq].double().cpu() - lse_ref).abs().max().item():.3e}")
    print("   per head   :", [f"{d[:, :, h].max().item():.2e}" for h in range(H)])
    print("   per channel:", [f"{d[..., c].max().item():.2e}" for c in range(15)])
    nt = (Lq + 15) // 16
    pt = [d[:, i * 16:(i + 1) * 16].max().item() for i in range(min(nt, 24))]
    print("   per 16-query tile:", [f"{x:.1e}" for x in pt])
    pq = [d[:, i::16].max().item() for i in range(16)]
    print("   per query-in-tile :", [f"{x:.1e}" for x in pq])


B, H, Lq, S = 2, 4, 333, 4097
Qr, Kr, Vp, kmask, o_ref, lse_ref, Lqp, Sp = T._exact_operands(B, H, Lq, S, 5, dev, False)
# A: zero logits -> O = mean v
Qz, Kz = torch.zeros_like(Qr), torch.zeros_like(Kr)
v = Vp[:, :, 0, :15, :S].double().cpu()                      # [B, H, 15, S]
o_mean = v.mean(-1).permute(0, 1, 2).unsqueeze(1).expand(B, Lq, H, 15).reshape(B, Lq, H * 15)
O, LSE, ops8 = run(Qz, Kz, Vp, None, B, H, Lq, S)
report("A zero logits (mean of v)", O, o_mean, LSE, torch.full((B, H, Lq), math.log2(S), dtype=torch.float64), B, H, Lq)
kv = ((B * H * Sp * 16 + 255) // 256) * 256
am = ops8[2 * kv:2 * kv + B * H * 16].view(torch.float32).view(B, H, 4).cpu()
print("   amax words (k, q, v):", am[0].tolist())
K8 = ops8[:B * H * Sp * 16].view(torch.float8_e4m3fn).view(B, H, Sp, 16).float().cpu()
V8 = ops8[kv:kv + B * H * Sp * 16].view(torch.float8_e4m3fn).view(B, H, 16, Sp).float().cpu()
print("   V8[0,0,:,0..3]:", V8[0, 0, :, :4].t().tolist()[0], " ones channel:", V8[0, 0, 15, :4].tolist(), V8[0, 0, 15, S - 2:S + 2].tolist())
print("   v [0,0,:,0]   :", Vp[0, 0, 0, :, 0].float().cpu().tolist())
# B: channel-constant values -> O = the constants whatever the weights
Vc = torch.zeros_like(Vp)
const = (torch.arange(16).float() - 7) / 8.0
Vc[:, :, 0, :, :S] = const.view(1, 1, 16, 1).half().to(dev)
Vc[:, :, 0, 15, :S] = 1.0
o_c = const[:15].double().view(1, 1, 1, 15).expand(B, Lq, H, 15).reshape(B, Lq, H * 15)
O, LSE, ops8 = run(Qr, Kr, Vc, None, B, H, Lq, S)
report("B channel-constant v, real logits", O, o_c, LSE, lse_ref, B, H, Lq)
K8 = ops8[:B * H * Sp * 16].view(torch.float8_e4m3fn).view(B, H, Sp, 16).float().cpu()
kk = Kr[:, :, :, :16].float().cpu()
for h in range(H):
    ratio = (K8[0, h, :S] / kk[0, h, :S]).nan_to_num(0.0)
    nz = ratio[ratio != 0]
    print(f"   head {h}: K8 / k ratios min {nz.min().item()} max {nz.max().item()}")
# C: the full case
O, LSE, ops8 = run(Qr, Kr, Vp, None, B, H, Lq, S)
report("C full exact case", O, o_ref, LSE, lse_ref, B, H, Lq)
# D: single head-0-like scales (no power-of-two shift): H = 1
Qr1, Kr1, Vp1, _, o1, l1, _, _ = T._exact_operands(1, 1, 64, 64, 9, dev, False)
O, LSE, _ = run(Qr1, Kr1, Vp1, None, 1, 1, 64, 64)
report("D one chunk, one head", O, o1, LSE, l1, 1, 1, 64)
Qr1, Kr1, Vp1, _, o1, l1, _, _ = T._exact_operands(1, 1, 64, 256, 9, dev, False)
O, LSE, _ = run(Qr1, Kr1, Vp1, None, 1, 1, 64, 256)
report("E four chunks, one head", O, o1, LSE, l1, 1, 1, 64)
