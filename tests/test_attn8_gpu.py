"""The opt-in fp8 (e4m3) attention forward, csrc/attention8.hip (BASELINE.json configs[4]: "fp8 MFMA attention").

Two kinds of check, both through the C-ABI:
 * mechanics, EXACT: operands chosen so that every quantity the kernel rounds is exactly representable (integer log2-logits
   in [-7, 7], weights that are powers of two, values on e4m3's grid) -- lane layouts, the LDS-DMA ring, key masks and
   tails, the power-of-two scales, the ones channel, key splits and both query-tile variants must then reproduce the
   float64 softmax to fp32 rounding;
 * accuracy at e4m3's tolerance: random projected operands at the logit range of the configs[4] fixture against float64;
 * the amax / pack kernels against the oracle quantiser (oracle/fp8.py), bit for bit.
"""
import importlib
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fp8 as OF  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def a3d():
    return importlib.import_module("act3d-chained-diffuser_amd")


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _pad(n, m):
    return (n + m - 1) // m * m


def _exact_operands(B, H, Lq, S, seed, dev, masked):
    """rows16 / planes16 operands whose fp8 images are exact.  q, k in {-1, 0, 1} (5 non-zero channels per query, so the
    log2-logits are integers in [-5, 5] and every weight 2^(s - m + 5) >= 2^-5 is a normal e4m3 number: with 7 channels the
    weights 2^-7 .. 2^-9 of the smallest keys are e4m3 subnormals, which the hardware path flushes -- measured as 4e-5 errors
    on exactly those queries); per head a power-of-two factor moves magnitude from q to k (exercises the
    balanced scales); v on a grid e4m3 holds exactly after the 2^ev scale; channel 15 of the hi plane = 1."""
    g = torch.Generator().manual_seed(seed)
    Lqp, Sp = _pad(Lq, 64), _pad(S, 64)
    q = torch.zeros(B, H, Lq, 16)
    for b in range(B):
        for h in range(H):
            idx = torch.rand(Lq, 15, generator=g).argsort(-1)[:, :5]
            sign = (torch.randint(0, 2, (Lq, 5), generator=g) * 2 - 1).float()
            q[b, h].scatter_(1, idx, sign)
    k = torch.randint(-1, 2, (B, H, S, 16), generator=g).float()
    k[..., 15] = 0
    v = (torch.randint(-14, 15, (B, H, S, 16), generator=g).float()) / 8.0
    v[..., 15] = 1.0
    shift = torch.tensor([0, 2, -3, 1, 4, -1, 0, 3])[:H].float()
    fq = (2.0 ** -shift).view(1, H, 1, 1)
    fk = (2.0 ** shift).view(1, H, 1, 1)
    kmask = None
    if masked:
        kmask = torch.rand(B, S, generator=g) < 0.3
        kmask[:, 0] = False
    Qr = torch.zeros(B, H, Lqp, 32, dtype=torch.float16)
    Kr = torch.zeros(B, H, Sp, 32, dtype=torch.float16)
    Vp = torch.zeros(B, H, 2, 16, Sp, dtype=torch.float16)
    Qr[:, :, :Lq, :16] = (q * fq).half()
    Kr[:, :, :S, :16] = (k * fk).half()
    Vp[:, :, 0, :, :S] = v.transpose(-1, -2).half()
    s = torch.einsum("bhqc,bhkc->bhqk", q.double(), k.double())           # log2 units
    if kmask is not None:
        s = s.masked_fill(kmask.view(B, 1, 1, S), -math.inf)
    m = s.max(-1, keepdim=True).values
    p = torch.exp2(s - m)
    den = p.sum(-1, keepdim=True)
    o = (p @ v[..., :15].double()) / den                                  # [B, H, Lq, 15]
    lse2 = (m + torch.log2(den)).squeeze(-1)
    return (Qr.to(dev), Kr.to(dev), Vp.to(dev), None if kmask is None else kmask.to(torch.uint8).to(dev),
            o.permute(0, 2, 1, 3).reshape(B, Lq, H * 15), lse2, Lqp, Sp)


@pytest.mark.parametrize("B,H,Lq,S,nsplit,masked", [
    (2, 4, 333, 4097, 1, False),      # the ghost attention of configs[1]: two-tile waves, unmasked body + masked tail chunk
    (1, 4, 2500, 3073, 1, False),     # configs[4]: 2500 ghost points per level
    (3, 2, 37, 131, 1, True),         # one-tile waves, key-padding mask, ragged tails on both axes
    (2, 8, 50, 3074, 4, True),        # key splits + combine (the diffusion shapes: 8 heads)
    (2, 4, 333, 4097, 3, False),      # two-tile waves with key splits
])
def test_attn8_forward_is_exact_on_representable_operands(a3d, dev, B, H, Lq, S, nsplit, masked):
    L = a3d.lib
    Qr, Kr, Vp, kmask, o_ref, lse_ref, Lqp, Sp = _exact_operands(B, H, Lq, S, 11 + Lq, dev, masked)
    O = torch.full((B, Lq, H * 15), float("nan"), device=dev)
    LSE = torch.full((B, H, Lqp), float("nan"), device=dev)
    ws = torch.empty(nsplit * B * H * Lqp * 18, device=dev) if nsplit > 1 else None
    ops8 = torch.empty(L.load().a3d_attn8_operand_bytes(B, H, Sp), device=dev, dtype=torch.uint8)
    L.call("a3d_attn8_fwd", Qr.data_ptr(), Kr.data_ptr(), Vp.data_ptr(), ops8.data_ptr(),
           None if kmask is None else kmask.data_ptr(), O.data_ptr(), LSE.data_ptr(), None if ws is None else ws.data_ptr(),
           B, H, Lq, Lqp, S, Sp, nsplit, L.stream())
    torch.cuda.synchronize()
    err_o = (O.double().cpu() - o_ref).abs().max().item()
    err_l = (LSE[:, :, :Lq].double().cpu() - lse_ref).abs().max().item()
    print(f"[parity] attn8 exact B={B} H={H} Lq={Lq} S={S} nsplit={nsplit}: O max_abs_err={err_o:.3e} (|O|max {o_ref.abs().max():.3f}) "
          f"LSE2 max_abs_err={err_l:.3e}")
    assert torch.isfinite(O).all() and torch.isfinite(LSE[:, :, :Lq]).all()
    assert err_o <= 1e-5 * max(1.0, o_ref.abs().max().item()), err_o       # fp32 accumulation over up to 4097 keys only
    assert err_l <= 2e-5, err_l


def fp8_tolerance(max_log2_logit):
    """Stated tolerance of the fp8 forward, as a function of the logit range.  An e4m3 rounding is 2^-5-class relative (2^-4
    worst case).  v and the softmax weights are rounded once each; k is rounded once and its error reaches the weights through
    the exponent: d(weight) / weight = ln2 * d(s2), d(s2) ~ 2^-5 * spread of the query's log2-logits, and the spread is about a
    quarter of the largest |log2-logit|.  Relative L2 of an attention output (itself an average over keys, so the
    independent errors do not average out relative to it): 2^-5 * (1.5 + L / 8), which is twice what is measured on random
    operands at L = 3 .. 100 (profiles/r03_attn8_check.txt); max-abs: 2.5 x that of the output's largest element."""
    return 2.0 ** -5 * (1.5 + max_log2_logit / 8.0)


def _projected_case(a3d, dev, B, Lq, S, gain, seed, H=4):
    """Random pre-projection rows through the product's own operand builder (no RoPE), and the float64 softmax."""
    O_ = a3d.ops
    E = H * 15
    g = torch.Generator().manual_seed(seed)
    q_pre = (torch.randn(B, Lq, E, generator=g) * gain).to(dev)
    k_pre = (torch.randn(B, S, E, generator=g) * gain).to(dev)
    v_pre = torch.randn(B, S, E, generator=g).to(dev)
    qh = (q_pre.double() * 15 ** -0.5).view(B, Lq, H, 15).transpose(1, 2)
    kh = k_pre.double().view(B, S, H, 15).transpose(1, 2)
    vh = v_pre.double().view(B, S, H, 15).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2)
    o = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Lq, E)
    qc, kc, vc = q_pre.reshape(B * Lq, E).contiguous(), k_pre.reshape(B * S, E).contiguous(), v_pre.reshape(B * S, E).contiguous()
    old = O_.ATTN_MODE
    O_.ATTN_MODE = "fp8"              # the fp8 mode's operand set: value PLANES (attention8.hip packs from them; the default set is rows-only)
    try:
        Qs, Ks, Vt, Lqp, Sp, scale, freq, extra = O_.attn_operands16(qc.data_ptr(), E, kc.data_ptr(), E, vc.data_ptr(), E, None, None,
                                                                     B, Lq, S, E, H, dev, need_bwd=False)
    finally:
        O_.ATTN_MODE = old
    return Qs, Ks, Vt, Lqp, Sp, o, (s.abs().max().item() * math.log2(math.e))


@pytest.mark.parametrize("gain", [0.5, 1.0, 2.0])
def test_attn8_accuracy_at_the_configs4_shapes(a3d, dev, gain):
    """Lq = 2500 ghost points against S = 3073 scene tokens (3 cameras) against float64, at three logit ranges, within
    fp8_tolerance(range); the split-fp16 kernel on the same operands is held to 1e-4 beside it."""
    O_ = a3d.ops
    B, H, Lq, S = 2, 4, 2500, 3073
    Qs, Ks, Vt, Lqp, Sp, o_ref, smax2 = _projected_case(a3d, dev, B, Lq, S, gain, 3)
    old = O_.ATTN_MODE
    try:
        O_.ATTN_MODE = "fp8"
        o8, _ = O_.attn_core_fwd(Qs, Ks, Vt, None, B, H, Lq, Lqp, S, Sp, 1)
        O_.ATTN_MODE = "f16"
        o16, _ = O_.attn_core_fwd(Qs, Ks, Vt, None, B, H, Lq, Lqp, S, Sp, 1)
    finally:
        O_.ATTN_MODE = old
    tol = fp8_tolerance(smax2)
    omax = o_ref.abs().max().item()
    e8 = (o8.double() - o_ref).abs().max().item()
    l8 = ((o8.double() - o_ref).norm() / o_ref.norm()).item()
    e16 = (o16.double() - o_ref).abs().max().item()
    print(f"[parity] attn8 accuracy gain={gain} max|log2-logit|={smax2:.1f} tolerance={tol:.3f}: fp8 max_abs_err={e8:.3e} "
          f"(|O|max {omax:.3f}) rel_l2={l8:.2e}   (split-fp16 kernel, same operands: max_abs_err={e16:.3e})")
    assert torch.isfinite(o8).all()
    assert l8 <= tol, (l8, tol)
    assert e8 <= 2.5 * tol * omax, (e8, tol, omax)
    assert e16 <= 1e-4 * max(1.0, omax)


def test_attn8_pack_kernels_equal_the_oracle_quantiser_bit_for_bit(a3d, dev):
    """amax reduction + pack kernel against oracle/fp8.py (itself pinned to torch's float8_e4m3fn on CPU): the per-(sample, head)
    maxima are the exact fp32 maxima of the fp16 hi parts, the scales follow from them, and every K8 / V8 byte is the
    round-to-nearest-even e4m3 code of (hi + lo) * 2^scale.  (A value exactly midway between two codes may take either
    neighbour -- counted and printed; everything else must match bit for bit.)"""
    L = a3d.lib
    B, H, Lq, S = 2, 4, 300, 1000
    Qs, Ks, Vt, Lqp, Sp, _, _ = _projected_case(a3d, dev, B, Lq, S, 1.5, 11)
    O = torch.empty((B, Lq, H * 15), device=dev)
    LSE = torch.empty((B, H, Lqp), device=dev)
    nbytes = L.load().a3d_attn8_operand_bytes(B, H, Sp)
    ops8 = torch.zeros(nbytes, device=dev, dtype=torch.uint8)
    L.call("a3d_attn8_fwd", Qs.data_ptr(), Ks.data_ptr(), Vt.data_ptr(), ops8.data_ptr(), None, O.data_ptr(), LSE.data_ptr(), None,
           B, H, Lq, Lqp, S, Sp, 1, L.stream())
    torch.cuda.synchronize()
    kv = (B * H * Sp * 16 + 255) // 256 * 256
    raw = ops8.cpu().numpy()
    K8 = raw[:B * H * Sp * 16].reshape(B, H, Sp, 16)
    V8 = raw[kv:kv + B * H * Sp * 16].reshape(B, H, 16, Sp)
    amax = raw[2 * kv:2 * kv + B * H * 16].view(np.float32).reshape(B, H, 4)
    q16 = Qs.float().cpu().numpy().reshape(B, H, Lqp, 32)
    k16 = Ks.float().cpu().numpy().reshape(B, H, Sp, 32)
    v16 = Vt.float().cpu().numpy().reshape(B, H, 2, 16, Sp)
    ref_amax = np.stack([np.abs(k16[:, :, :S, :16]).max((2, 3)), np.abs(q16[:, :, :Lq, :16]).max((2, 3)),
                         np.abs(v16[:, :, 0, :15]).max((2, 3))], -1)
    assert np.array_equal(amax[..., :3], ref_amax), (amax[..., :3], ref_amax)
    ek, ev = OF.attention_scales(ref_amax[..., 0], ref_amax[..., 1], ref_amax[..., 2])
    kx = (k16[..., :16] + k16[..., 16:]) * np.exp2(ek.astype(np.float32))[..., None, None]
    vx = v16[:, :, 0] + v16[:, :, 1]
    vx[:, :, :15] *= np.exp2(ev.astype(np.float32))[..., None, None]          # channel 15 (the ones column) is not scaled
    ties = 0
    for name, got, x in (("K8", K8, kx), ("V8", V8, vx)):
        ref = OF.e4m3_bytes(x)
        bad = np.nonzero(got != ref)
        if bad[0].size:
            g, r_, xv = OF.e4m3_values(got[bad]).astype(np.float64), OF.e4m3_values(ref[bad]).astype(np.float64), x[bad].astype(np.float64)
            is_tie = np.abs(np.abs(xv - g) - np.abs(xv - r_)) == 0.0
            assert is_tie.all(), f"{name}: {int((~is_tie).sum())} bytes differ from the oracle, e.g. x={xv[~is_tie][:4]} got={g[~is_tie][:4]} ref={r_[~is_tie][:4]}"
            ties += int(is_tie.sum())
    print(f"[parity] attn8 pack: amax words, {K8.size} K8 bytes and {V8.size} V8 bytes equal the oracle; {ties} exact ties resolved differently")
    assert torch.isfinite(O).all()
