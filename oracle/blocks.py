"""ORACLE (test infrastructure only) -- CPU restatement of the reference's attention building blocks.

Plain PyTorch-CPU fp32, functional, batch-first.  Nothing under oracle/ is imported by the product package
(`act3d-chained-diffuser_amd/`); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
checker / the timed CPU baseline.  Every function cites the reference lines (relative to /root/reference) it restates.
Parity of these functions with the imported reference is pinned by tests/golden/*.pt (tests/test_oracle_golden.py).
"""
import math

import torch
import torch.nn.functional as F

# Test hook: when set to a list, every ReLU (and planner_loss's L1) appends (site, min |argument|) -- the distance of the
# evaluation from the nearest kink.  Gradients of a ReLU / L1 network are only comparable between two implementations when
# no argument is within forward rounding of zero; tests/golden/make_dropout_case.py searches a case with a stated margin.
KINKS = None


# Test aid (tests/test_diffusion_gpu.py, cfg-4 gradients): RELU_MASKS = {site: [bool mask per call, in call order]} makes relu(x)
# return x * mask -- the branch of the piecewise-linear network ANOTHER implementation took (the device's ReLU masks) -- and
# RELU_DIFFS collects, per call, how many units that differs from this run's own sign pattern on and the largest |pre-activation|
# among them.  A unit whose pre-activation is within rounding of zero is a kink: two correct fp32 implementations may land on
# different sides, and their gradients then differ by a discrete step however exact their arithmetic is.
RELU_MASKS = None
RELU_DIFFS = None


def relu(x, site):
    if KINKS is not None:
        KINKS.append((site, x.detach().abs().min().item()))
    if RELU_MASKS is not None and RELU_MASKS.get(site):
        mask = RELU_MASKS[site].pop(0).reshape(x.shape)
        if RELU_DIFFS is not None:
            diff = (x.detach() > 0) != mask
            n = int(diff.sum())
            RELU_DIFFS.append((site, n, x.detach().abs()[diff].max().item() if n else 0.0, x.detach().abs().max().item()))
        return x * mask.to(x.dtype)
    return F.relu(x)


# Test aid (tests/test_diffusion_gpu.py, cfg-4 gradients): LIFT = torch.float64 makes the oracle evaluate THE SAME FUNCTION in double
# precision -- the constants the reference computes in fp32 under no_grad (RoPE codes, sinusoidal embeddings, DDPM tables) are still
# computed in fp32, then cast up -- so that the fp32 oracle's own rounding error can be measured next to the device's.  None: no cast.
LIFT = None


def _lift(t):
    return t if LIFT is None else t.to(LIFT)


def rope3d_code(xyz, E):
    """cos, sin tables (B, N, E) for points xyz (B, N, 3).

    model/utils/position_encodings.py:64-97: per axis third, frequencies exp(arange(0, E/3, 2) * -ln(1e4)/(E/3)),
    every value duplicated for the (2i, 2i+1) channel pair, axes concatenated x|y|z.
    """
    third = E // 3
    div = torch.exp(torch.arange(0, third, 2, dtype=torch.float32, device=xyz.device)
                    * (-math.log(10000.0) / third))                       # (E/6,)
    ang = xyz.to(torch.float32).unsqueeze(-1) * div                       # (B, N, 3, E/6)
    cos = torch.cos(ang).repeat_interleave(2, dim=-1).flatten(-2)         # (B, N, E): x third | y third | z third
    sin = torch.sin(ang).repeat_interleave(2, dim=-1).flatten(-2)
    return _lift(cos), _lift(sin)


def rotary_apply(x, cos, sin):
    """position_encodings.py:31-34: y = x*cos + rot(x)*sin with rot(x)[2i] = -x[2i+1], rot(x)[2i+1] = x[2i]."""
    xr = torch.empty_like(x)
    xr[..., 0::2] = -x[..., 1::2]
    xr[..., 1::2] = x[..., 0::2]
    return x * cos + xr * sin


def sinusoidal(x, E):
    """position_encodings.py:13-20: [sin(x f_j) | cos(x f_j)], f_j = exp(-j ln(1e4)/(E/2 - 1))."""
    half = E // 2
    f = torch.exp(torch.arange(half, dtype=torch.float32, device=x.device) * -(math.log(10000) / (half - 1)))
    a = x.to(torch.float32)[:, None] * f[None, :]
    return _lift(torch.cat((a.sin(), a.cos()), dim=-1))


def mha(q_in, k_in, v_in, in_w, in_b, out_w, out_b, H, q_xyz=None, k_xyz=None, key_padding_mask=None,
        return_weights=False, drop=None, site=0):
    """multihead_custom_attention.py:157-462 restricted to the paths the hot path takes.

    Batch-first: q_in (B, Lq, E), k_in/v_in (B, S, E).  q is scaled by d^-1/2 BEFORE the rotation (:325), the
    rotation acts on the full E vector before the head split (:348-359), padded keys get -inf (:398-404).
    """
    B, Lq, E = q_in.shape
    S = k_in.shape[1]
    d = E // H
    q = F.linear(q_in, in_w[:E], in_b[:E]) * (float(d) ** -0.5)
    k = F.linear(k_in, in_w[E:2 * E], in_b[E:2 * E])
    v = F.linear(v_in, in_w[2 * E:], in_b[2 * E:])
    if q_xyz is not None:
        qc, qs = rope3d_code(q_xyz, E)
        kc, ks = rope3d_code(k_xyz, E)
        q = rotary_apply(q, qc, qs)
        k = rotary_apply(k, kc, ks)
    qh = q.view(B, Lq, H, d).transpose(1, 2)              # (B, H, Lq, d)
    kh = k.view(B, S, H, d).transpose(1, 2)
    vh = v.view(B, S, H, d).transpose(1, 2)
    w = qh @ kh.transpose(-1, -2)                         # (B, H, Lq, S)
    if key_padding_mask is not None:
        w = w.masked_fill(key_padding_mask[:, None, None, :].bool(), float("-inf"))
    w = torch.softmax(w, dim=-1)
    if drop is not None:                                  # F.dropout on the attention weights (:413), training mode
        w = w * torch.from_numpy(drop.attn(site, B, H, Lq, S))
    o = (w @ vh).transpose(1, 2).reshape(B, Lq, E)
    o = F.linear(o, out_w, out_b)
    return (o, w) if return_weights else o


def layer_norm(x, g, b):
    return F.layer_norm(x, (x.shape[-1],), g, b, 1e-5)


def rel_cross_attn_module(P, prefix, n_layers, query, ctx, H, q_xyz=None, ctx_xyz=None):
    """layers.py:293-351 RelativeCrossAttentionModule: per layer x = LN(x + MHA(x, ctx, ctx)); x = LN(x + FFN(x)),
    FFN hidden = E, ReLU.  Returns the list of per-layer outputs.  P maps reference state-dict names to tensors."""
    outs = []
    x = query
    for i in range(n_layers):
        a = f"{prefix}.attn_layers.{i}."
        o = mha(x, ctx, ctx, P[a + "multihead_attn.in_proj_weight"], P[a + "multihead_attn.in_proj_bias"],
                P[a + "multihead_attn.out_proj.weight"], P[a + "multihead_attn.out_proj.bias"], H, q_xyz, ctx_xyz)
        x = layer_norm(x + o, P[a + "norm.weight"], P[a + "norm.bias"])
        f = f"{prefix}.ffw_layers.{i}."
        hdn = relu(F.linear(x, P[f + "linear1.weight"], P[f + "linear1.bias"]), f + "linear1")
        x = layer_norm(x + F.linear(hdn, P[f + "linear2.weight"], P[f + "linear2.bias"]), P[f + "norm.weight"],
                       P[f + "norm.bias"])
        outs.append(x)
    return outs


def adaln(x, t, w, b):
    """layers.py:273-290: (scale, shift) = Linear(SiLU(t)).chunk(2); x * (1 + scale) + shift."""
    mod = F.linear(F.silu(t), w, b)
    scale, shift = mod.chunk(2, dim=-1)
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def _drop(x, drop, site):
    """nn.Dropout in training mode with the device's mask (oracle.sampling.DropoutTwin); identity when drop is None"""
    return x if drop is None else x * torch.from_numpy(drop.flat(site, tuple(x.shape)))


def parallel_attention_layer(P, prefix, seq1, seq1_mask, seq2, H, seq1_xyz=None, seq2_xyz=None, seq1_sem=None,
                             ada=None, self_attn=True, apply_ffn=True, use_adaln=True, drop=None, name_root="prediction_head."):
    """layers.py:115-218 ParallelAttentionLayer as configured on the hot path (pre_norm=False, only the seq1
    stream is updated, cross_attention1=True, self_attention1=self_attn).  drop: None (eval / p = 0) or a
    DropoutTwin -- training-mode dropout (layers.py:10,34,58,82-84) on the attention weights (site + 0 / + 2), the
    residual branches (+ 1 / + 3) and the FFN (+ 4 hidden, + 5 output), sites named after the module path below
    `name_root`.

    (1) cross: q = AdaLN12(seq1 + sem); key = value = seq2;  seq1 = LN12(seq1 + MHA)
    (2) self : q = k = AdaLN1(seq1 + sem), v = AdaLN1(seq1), key_padding_mask = seq1_mask; seq1 = LN1(seq1 + MHA)
    (3) ffn  : y = AdaLNff(seq1); seq1 = LN122(y + FFN(y))
    With rotary_pe=False (vl_attention / traj_lang_attention) no xyz is passed and positions are not added (they
    are None in the reference calls, diffusion_head.py:306-335).
    """
    def ada_or_id(x, name):
        if use_adaln and ada is not None:
            return adaln(x, ada, P[f"{prefix}.{name}.modulation.1.weight"], P[f"{prefix}.{name}.modulation.1.bias"])
        return x

    sb = 0
    if drop is not None:
        sb = drop.site_id(prefix[len(name_root):] if prefix.startswith(name_root) else prefix)
    q1 = seq1 if seq1_sem is None else seq1 + seq1_sem
    o = mha(ada_or_id(q1, "adaln_12"), seq2, seq2, P[prefix + ".cross_12.in_proj_weight"],
            P[prefix + ".cross_12.in_proj_bias"], P[prefix + ".cross_12.out_proj.weight"],
            P[prefix + ".cross_12.out_proj.bias"], H, seq1_xyz, seq2_xyz, drop=drop, site=sb)
    seq1 = layer_norm(seq1 + _drop(o, drop, sb + 1), P[prefix + ".norm_12.weight"], P[prefix + ".norm_12.bias"])
    if self_attn:
        q1 = seq1 if seq1_sem is None else seq1 + seq1_sem
        qk = ada_or_id(q1, "adaln_1")
        vv = ada_or_id(seq1, "adaln_1")
        o = mha(qk, qk, vv, P[prefix + ".sa1.in_proj_weight"], P[prefix + ".sa1.in_proj_bias"],
                P[prefix + ".sa1.out_proj.weight"], P[prefix + ".sa1.out_proj.bias"], H, seq1_xyz, seq1_xyz,
                key_padding_mask=seq1_mask, drop=drop, site=sb + 2)
        seq1 = layer_norm(seq1 + _drop(o, drop, sb + 3), P[prefix + ".norm_1.weight"], P[prefix + ".norm_1.bias"])
    if apply_ffn:
        y = ada_or_id(seq1, "adaln_ff1")
        hdn = _drop(relu(F.linear(y, P[prefix + ".ffn_12.0.weight"], P[prefix + ".ffn_12.0.bias"]), prefix + ".ffn_12.0"), drop, sb + 4)
        ffn = _drop(F.linear(hdn, P[prefix + ".ffn_12.3.weight"], P[prefix + ".ffn_12.3.bias"]), drop, sb + 5)
        seq1 = layer_norm(y + ffn, P[prefix + ".norm_122.weight"], P[prefix + ".norm_122.bias"])
    return seq1


def parallel_attention(P, prefix, n_layers, seq1, seq1_mask, seq2, H, **kw):
    """layers.py:221-270: stack of the above, seq1 updated layer after layer."""
    for i in range(n_layers):
        seq1 = parallel_attention_layer(P, f"{prefix}.layers.{i}", seq1, seq1_mask, seq2, H, **kw)
    return seq1


def mlp2(x, P, prefix, i0="0", i1="2", drop=None, name_root="prediction_head."):
    """Linear-ReLU-Linear heads (act3d.py:162-166; diffusion_head.py:41-46 uses indices 0 and 3, with an nn.Dropout(0.1)
    between the ReLU and the second Linear: `drop` applies it, site = (module path, 4))."""
    hdn = relu(F.linear(x, P[f"{prefix}.{i0}.weight"], P[f"{prefix}.{i0}.bias"]), f"{prefix}.{i0}")
    if drop is not None:
        hdn = _drop(hdn, drop, drop.site_id(prefix[len(name_root):] if prefix.startswith(name_root) else prefix, 4))
    return F.linear(hdn, P[f"{prefix}.{i1}.weight"], P[f"{prefix}.{i1}.bias"])
