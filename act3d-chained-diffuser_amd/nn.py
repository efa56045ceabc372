"""Parameter containers that mirror the reference's module tree (names, shapes, initialisation) so that state
dicts are interchangeable, with forward passes that launch the HIP operators of ops.py.

Reference classes mirrored: MultiheadCustomAttention (multihead_custom_attention.py:14-94), RelativeCrossAttentionLayer /
FeedforwardLayer / RelativeCrossAttentionModule (layers.py:293-351), ParallelAttentionLayer / ParallelAttention / AdaLN
(layers.py:7-290), torchvision's FeaturePyramidNetwork (third-party; restated) and a synthetic CLIP-RN50-shaped backbone.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops as O


class MultiheadCustomAttention(nn.Module):
    """Holds in_proj_weight (3E,E), in_proj_bias (3E), out_proj (Linear E->E); init as multihead_custom_attention.py:80-94."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        assert embed_dim % num_heads == 0 and embed_dim // num_heads == 15, \
            "the HIP attention kernels are specialised for head_dim 15 (60/4, 120/8), as in both reference models"
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)


class RelativeCrossAttentionLayer(nn.Module):
    def __init__(self, embedding_dim, num_heads, dropout=0.0):
        super().__init__()
        self.multihead_attn = MultiheadCustomAttention(embedding_dim, num_heads, dropout=dropout)
        self.norm = nn.LayerNorm(embedding_dim)
        self.num_heads = num_heads

    def forward(self, query, value, query_xyz=None, value_xyz=None, pad_mask=None, sink=None):
        """query (B, Lq, E), value (B, S, E) batch-first; xyz instead of materialised rotary codes.  sink: the GradSink of
        `value` when its consumers share one gradient buffer (ops.GradSink)."""
        return O.attn_block(query, value, value, query, query_xyz, value_xyz, pad_mask, self.multihead_attn, self.norm,
                            self.num_heads, sink=sink)


class FeedforwardLayer(nn.Module):
    def __init__(self, embedding_dim, hidden_dim, dropout=0.0):
        super().__init__()
        self.linear1 = nn.Linear(embedding_dim, hidden_dim)
        self.linear2 = nn.Linear(hidden_dim, embedding_dim)
        self.norm = nn.LayerNorm(embedding_dim)
        for p in self.parameters():                   # layers.py:323-326
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, x):
        return O.mlp(x, self.linear1, self.linear2, self.norm)


class RelativeCrossAttentionModule(nn.Module):
    def __init__(self, embedding_dim, num_attn_heads, num_layers):
        super().__init__()
        self.attn_layers = nn.ModuleList(
            [RelativeCrossAttentionLayer(embedding_dim, num_attn_heads) for _ in range(num_layers)])
        self.ffw_layers = nn.ModuleList([FeedforwardLayer(embedding_dim, embedding_dim) for _ in range(num_layers)])

    def forward(self, query, value, query_xyz=None, value_xyz=None):
        """Returns the list of per-layer outputs (layers.py:345-351), batch-first."""
        output = []
        sink = getattr(value, "_a3d_sink", None)      # the context's shared gradient buffer (act3d.py attaches it), or None
        for attn, ffw in zip(self.attn_layers, self.ffw_layers):
            mha = attn.multihead_attn
            if O.query_layer_applicable(query, value, mha.embed_dim, attn.num_heads, ffw.linear1.out_features):
                # the one-query stream: attention block + FFN of a layer as the fused launches of csrc/query_stream.hip
                query = O.QueryLayerFn.apply(query, value, query_xyz, value_xyz, mha.in_proj_weight, mha.in_proj_bias,
                                             mha.out_proj.weight, mha.out_proj.bias, attn.norm.weight, attn.norm.bias,
                                             ffw.linear1.weight, ffw.linear1.bias, ffw.linear2.weight, ffw.linear2.bias,
                                             ffw.norm.weight, ffw.norm.bias, attn.num_heads, sink)
            else:
                query = ffw(attn(query, value, query_xyz, value_xyz, sink=sink))
            output.append(query)
        return output


class AdaLN(nn.Module):
    def __init__(self, embedding_dim):
        super().__init__()
        self.modulation = nn.Sequential(nn.SiLU(), nn.Linear(embedding_dim, 2 * embedding_dim, bias=True))
        nn.init.constant_(self.modulation[-1].weight, 0)
        nn.init.constant_(self.modulation[-1].bias, 0)

    def forward(self, x, silu_t):
        """x (B, N, C); silu_t = SiLU(t) (B, C), computed once per forward and shared by every AdaLN."""
        return O.AdaLNFn.apply(x, O.linear(silu_t, self.modulation[1]))


class ParallelAttentionLayer(nn.Module):
    """The seq1-only configuration the hot path uses (cross_attention1 [+ self_attention1] [+ ffn]); layers.py:7-218."""

    def __init__(self, d_model=256, dropout=0.1, n_heads=8, pre_norm=False, self_attention1=True,
                 self_attention2=False, cross_attention1=True, cross_attention2=False, apply_ffn=True,
                 slot_attention12=False, slot_attention21=False, rotary_pe=False, use_adaln=False):
        super().__init__()
        if pre_norm or self_attention2 or cross_attention2 or slot_attention12 or slot_attention21 or not cross_attention1:
            raise NotImplementedError("only the seq1 / post-norm configuration of ParallelAttentionLayer is on the hot "
                                      "path (SURVEY §8a-12); other options are not implemented")
        self.self_attention1, self.apply_ffn, self.rotary_pe = self_attention1, apply_ffn, rotary_pe
        self.n_heads, self.dropout_p = n_heads, dropout
        self.site_base = O.site_id("parallel_attention_layer")     # re-assigned from the module's name by its owner
        if self_attention1:
            self.adaln_1 = AdaLN(d_model) if use_adaln else None
            self.sa1 = MultiheadCustomAttention(d_model, n_heads, dropout=dropout)
            self.norm_1 = nn.LayerNorm(d_model)
        self.adaln_12 = AdaLN(d_model) if use_adaln else None
        self.cross_12 = MultiheadCustomAttention(d_model, n_heads, dropout=dropout)
        self.norm_12 = nn.LayerNorm(d_model)
        # the reference builds FFN-1 whenever seq1 is updated, even if apply_ffn=False leaves it unused
        self.adaln_ff1 = AdaLN(d_model) if use_adaln else None
        self.ffn_12 = nn.Sequential(nn.Linear(d_model, 4 * d_model), nn.ReLU(), nn.Dropout(dropout),
                                    nn.Linear(4 * d_model, d_model), nn.Dropout(dropout))
        self.norm_122 = nn.LayerNorm(d_model)

    def forward(self, seq1, seq1_key_padding_mask, seq2, seq1_xyz=None, seq2_xyz=None, seq1_sem_pos=None, silu_t=None,
                drop=None):
        """drop: ops.DropCtx of this forward pass (None: no dropout).  Training with dropout_p > 0 needs one -- the owner
        (DiffusionHead.begin_dropout) creates it; there is no silent un-regularised path."""
        if self.training and self.dropout_p > 0 and drop is None:
            raise RuntimeError("ParallelAttentionLayer in training mode with dropout=%g needs the forward pass's DropCtx "
                               "(DiffusionHead.begin_dropout())" % self.dropout_p)
        if drop is not None and (not self.training or self.dropout_p <= 0):
            drop = None
        sb = self.site_base
        rope = self.rotary_pe
        q1 = seq1 if seq1_sem_pos is None else O.AddRowsFn.apply(seq1, seq1_sem_pos)
        qa = self.adaln_12(q1, silu_t) if (self.adaln_12 is not None and silu_t is not None) else q1
        # (seq2 may carry a shared gradient sink -- DiffusionHead.forward_tokens attaches one to the context all its cross-attention
        # layers read: their k | v projection gradients are then parked and run as ONE input- and ONE weight-gradient GEMM)
        seq1 = O.attn_block(qa, seq2, seq2, seq1, seq1_xyz if rope else None, seq2_xyz if rope else None, None,
                            self.cross_12, self.norm_12, self.n_heads, drop=drop, site=sb, sink=getattr(seq2, "_a3d_sink", None))
        if self.self_attention1:
            q1 = seq1 if seq1_sem_pos is None else O.AddRowsFn.apply(seq1, seq1_sem_pos)
            if self.adaln_1 is not None and silu_t is not None:
                qk, vv = self.adaln_1(q1, silu_t), self.adaln_1(seq1, silu_t)
            else:
                qk, vv = q1, seq1
            seq1 = O.attn_block(qk, qk, vv, seq1, seq1_xyz if rope else None, seq1_xyz if rope else None,
                                seq1_key_padding_mask, self.sa1, self.norm_1, self.n_heads, drop=drop, site=sb + 2)
        if self.apply_ffn:
            y = self.adaln_ff1(seq1, silu_t) if (self.adaln_ff1 is not None and silu_t is not None) else seq1
            seq1 = O.mlp(y, self.ffn_12[0], self.ffn_12[3], self.norm_122, drop=drop, site_hidden=sb + 4, site_out=sb + 5)
        return seq1


class ParallelAttention(nn.Module):
    def __init__(self, num_layers=1, **kw):
        super().__init__()
        self.layers = nn.ModuleList([ParallelAttentionLayer(**kw) for _ in range(num_layers)])

    def forward(self, seq1, seq1_key_padding_mask, seq2, **kw):
        for layer in self.layers:
            seq1 = layer(seq1, seq1_key_padding_mask, seq2, **kw)
        return seq1


# ------------------------------------------------------------------------------------------------ adjacent: vision
class _Upsample2AddFn(torch.autograd.Function):
    """lat + bias + F.interpolate(top, 2x, nearest) for bf16 channels_last maps in one kernel; backward: dtop = 2x2 sums of
    dy and, in the same pass, the column sums of dy = the bias gradient, accumulated straight into bias.grad (as every wgrad
    kernel of ops.py does).  `bias`: the lateral convolution's fp32 Parameter (<= C entries; the convolution itself runs
    bias-free) or None; `top` None at the pyramid's top level."""

    @staticmethod
    def forward(ctx, lat, top, bias):
        N, C, H, W = lat.shape
        y = torch.empty_like(lat)
        O.L.call("a3d_upsample2_add_fwd", lat.data_ptr(), None if top is None else top.data_ptr(),
                 None if bias is None else bias.data_ptr(), 0 if bias is None else bias.numel(), y.data_ptr(), N, H, W, C,
                 O.L.stream())
        ctx.bias = bias
        ctx.has_top = top is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        dy, dtop = _top_down_backward(dy, ctx.bias, ctx.has_top and ctx.needs_input_grad[1])
        return (dy if ctx.needs_input_grad[0] else None), dtop, None


def _top_down_backward(dy, bias, want_top):
    """backward of y = lat + bias + up2(top) from dy (N, C, H, W) bf16: d top = the 2x2 sums of dy, d bias accumulated straight into
    bias.grad (the lateral branch's gradient is dy itself).  Returns (dy channels_last, d top or None)."""
    N, C, H, W = dy.shape
    want_bias = bias is not None and bias.requires_grad
    dtop = None
    if want_top or want_bias:
        dy = dy.contiguous(memory_format=torch.channels_last)
        if want_top:
            dtop = torch.empty((N, C, H // 2, W // 2), device=dy.device, dtype=dy.dtype, memory_format=torch.channels_last)
        ws = gb = None
        if want_bias:
            ws = torch.empty((O.L.load().a3d_upsample2_add_bwd_ws_floats(N, H, W, C),), device=dy.device, dtype=torch.float32)
            gb = O.grad_buf(bias)
        O.L.call("a3d_upsample2_add_bwd", dy.data_ptr(), None if dtop is None else dtop.data_ptr(),
                 None if gb is None else gb.data_ptr(), 0 if gb is None else gb.numel(), None if ws is None else ws.data_ptr(),
                 N, H, W, C, O.L.stream())
    return dy, dtop


# The FPN's lateral 1x1 convolutions of the fine levels (K <= 256 input channels: res1 / res2 of the CLIP ResNet) as ONE launch with the
# bias and the top-down add in the epilogue (a3d_conv1x1_topdown_fwd) instead of library convolution + a3d_upsample2_add_fwd: at the
# bench shape 332 + 375 us -> one ~1.2 GB pass for the 128 x 128 level (profiles/r06_fpn_fwd_probe.json).  A3D_FPN_LATERAL=0: A/B.
FUSED_FPN_LATERAL = os.environ.get("A3D_FPN_LATERAL", "1") not in ("0", "", "off")
# The FPN's 3x3 output convolutions (64 padded channels) through the backbone's implicit-GEMM stream kernel (a3d_conv3x3_bn_fwd without
# its BatchNorm folds) instead of the library: 911 -> 405 us for the 128 x 128 level of 256 images.  A3D_FPN_OUT3X3=0: A/B.
FUSED_FPN_OUT3X3 = os.environ.get("A3D_FPN_OUT3X3", "1") not in ("0", "", "off")


class _LateralTopDownFn(torch.autograd.Function):
    """bf16(conv1x1(x, w) + bias + up2(top)) in one launch (torchvision FPN: inner_blocks[i] + the top-down add).  x bf16 channels_last
    (the frozen backbone's map), w the fp32 weight (Co, K, 1, 1) (already zero-padded to the map width), bias the fp32 Parameter
    (<= Co entries), top bf16 channels_last at half the resolution.  Backward: (d top, d bias) as _Upsample2AddFn, d w from the library's
    weight-gradient kernel (the same one the unfused convolution's autograd runs), d x only when asked for."""

    @staticmethod
    def forward(ctx, x, w, bias, top):
        N, K, H, W = x.shape
        Co = w.shape[0]
        w16 = w.detach().to(torch.bfloat16).reshape(Co, K).contiguous()
        y = torch.empty((N, Co, H, W), device=x.device, dtype=torch.bfloat16, memory_format=torch.channels_last)
        O.L.call("a3d_conv1x1_topdown_fwd", x.data_ptr(), w16.data_ptr(), None if bias is None else bias.data_ptr(),
                 0 if bias is None else bias.numel(), None if top is None else top.data_ptr(), y.data_ptr(), N, H, W, K, Co, O.L.stream())
        ctx.save_for_backward(x, w16)
        ctx.bias = bias
        ctx.has_top = top is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w16 = ctx.saved_tensors
        dy, dtop = _top_down_backward(dy, ctx.bias, ctx.has_top and ctx.needs_input_grad[3])
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = dw = None
        if need_x or need_w:
            dy = dy.contiguous(memory_format=torch.channels_last)
            w4 = w16.view(w16.shape[0], w16.shape[1], 1, 1)
            dx, dw, _ = torch.ops.aten.convolution_backward(dy, x, w4, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, (need_x, need_w, False))
        return dx, (None if dw is None else dw.float()), None, dtop


def _fused_lateral_ok(x, Co, last):
    cl = torch.channels_last
    return bool(FUSED_FPN_LATERAL and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.is_contiguous(memory_format=cl)
                and O.L.load().a3d_conv1x1_topdown_serves(x.shape[1], Co)
                and (last is None or (last.dtype == torch.bfloat16 and last.is_contiguous(memory_format=cl) and last.shape[1] == Co
                                      and x.shape[-2] == 2 * last.shape[-2] and x.shape[-1] == 2 * last.shape[-1])))


def _fused_top_down_ok(lat, last, with_bias=False):
    cl = torch.channels_last
    ok = (lat.is_cuda and lat.dtype == torch.bfloat16 and lat.shape[1] % 4 == 0 and lat.shape[-2] % 2 == 0 and lat.shape[-1] % 2 == 0
          and lat.is_contiguous(memory_format=cl))
    if last is not None:
        ok = ok and (last.dtype == torch.bfloat16 and lat.shape[-2] == 2 * last.shape[-2] and lat.shape[-1] == 2 * last.shape[-1]
                     and last.is_contiguous(memory_format=cl))
    if with_bias:
        ok = ok and 256 % (lat.shape[1] // 4) == 0
    return ok


def fpn_top_down(lat, last, bias=None):
    """inner_lateral (+ its convolution's bias, when the convolution ran bias-free) + nearest-upsampled last_inner
    (torchvision FPN); fused HIP kernel for the exact-2x bf16 NHWC case."""
    if _fused_top_down_ok(lat, last, bias is not None):
        return _Upsample2AddFn.apply(lat, last, bias)
    if bias is not None:
        lat = lat + F.pad(bias, (0, lat.shape[1] - bias.numel())).to(lat.dtype).view(1, -1, 1, 1)
    return lat if last is None else lat + F.interpolate(last, size=lat.shape[-2:], mode="nearest")


class _LayerConv3x3Fn(torch.autograd.Function):
    """The FPN's 3x3 output convolution (bias-free, stride 1, padding 1, bf16 channels_last, 64 channels) with a SPLIT backward:
    the input gradient is the library's (MIOpen), the weight gradient is taken from the map's gather consumers when they have
    computed it token-sparsely (ops.SparseConvCtx / csrc/fpn_sparse.hip) -- the dense weight-gradient kernel (0.72 ms per step at
    the bench shape, over a gradient map that is non-zero on 6 - 12 % of its pixels) then does not run at all."""

    @staticmethod
    def forward(ctx, x, w, cc):
        w16 = w.to(x.dtype)
        if (FUSED_FPN_OUT3X3 and x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last)
                and O.L.load().a3d_conv3x3_serves(x.shape[1], w16.shape[0], x.shape[2], x.shape[3])):
            # the backbone's implicit-GEMM stream kernel without its BatchNorm folds (a3d_conv3x3_bn_fwd; w [Cout][3][3][Cin] in memory)
            w16 = w16.contiguous(memory_format=torch.channels_last)
            y = torch.empty((x.shape[0], w16.shape[0], x.shape[2], x.shape[3]), device=x.device, dtype=torch.bfloat16,
                            memory_format=torch.channels_last)
            O.L.call("a3d_conv3x3_bn_fwd", x.data_ptr(), w16.data_ptr(), None, None, 0, y.data_ptr(), None, x.shape[0], x.shape[2], x.shape[3],
                     x.shape[1], w16.shape[0], O.L.stream())
        else:
            y = F.conv2d(x, w16, None, padding=1)
        ctx.save_for_backward(x, w16)
        ctx.cc = cc
        cc.x, cc.dw, cc.dense = x, None, False
        cc.mask, cc.ntok = None, 0
        if O.SPARSE_FPN_DGRAD and ctx.needs_input_grad[0] and O.L.load().a3d_conv3x3_tile_count(x.shape[0], x.shape[2], x.shape[3]) > 0 \
                and x.shape[2] // 8 <= 255 and x.shape[3] // 32 <= 255:
            # tile marks of this step's backward (a fill KERNEL: captured memsets replay wrong on this stack)
            cc.mask = torch.zeros((O.L.load().a3d_conv3x3_tile_count(x.shape[0], x.shape[2], x.shape[3]),), device=x.device, dtype=torch.uint8)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w16 = ctx.saved_tensors
        cc = ctx.cc
        sparse = cc.dw is not None and not cc.dense
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = dw = None
        # token-sparse input gradient: every consumer of the map was a gather that marked its tiles, and together they hold at most
        # SPARSE_FPN_DGRAD_MAX of the map's pixels (a static bound: the marked-tile count lives on the device)
        N, _, H, W = x.shape
        sparse_x = (need_x and sparse and cc.mask is not None and 0 < cc.ntok <= O.SPARSE_FPN_DGRAD_MAX * N * H * W and dy.dtype == torch.bfloat16)
        if sparse_x:
            dy = dy.contiguous(memory_format=torch.channels_last)
            wt = w16.flip(2, 3).permute(1, 2, 3, 0).contiguous()            # wt[ci][kh][kw][co] = w[co][ci][2 - kh][2 - kw]
            dx = torch.empty_like(x, memory_format=torch.channels_last)
            ws = torch.empty((O.L.load().a3d_conv3x3_dgrad_tiles_ws_ints(N, H, W),), device=x.device, dtype=torch.int32)
            O.L.call("a3d_conv3x3_dgrad_tiles", dy.data_ptr(), wt.data_ptr(), cc.mask.data_ptr(), ws.data_ptr(), dx.data_ptr(), N, H, W, O.L.stream())
            need_x = False
        if need_x or (need_w and not sparse):
            dy = dy.contiguous(memory_format=torch.channels_last)
            dxl, dw, _ = torch.ops.aten.convolution_backward(dy, x, w16, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                             (need_x, need_w and not sparse, False))
            if need_x:
                dx = dxl
        if need_w and sparse:
            dw = cc.dw.permute(2, 3, 0, 1).contiguous()        # [kh][kw][co][ci] -> the weight's [co][ci][kh][kw]
        cc.dw, cc.mask = None, None
        return dx, (None if dw is None else dw.float()), None


class FeaturePyramidNetwork(nn.Module):
    """torchvision.ops.FeaturePyramidNetwork (0.14 naming: inner_blocks.i.0 / layer_blocks.i.0), restated: 1x1 lateral
    convs, nearest top-down pathway, 3x3 output convs.  Runs on PyTorch-ROCm/MIOpen (adjacent to the hot path,
    SURVEY §2a-8 / §8f-1).  `needed` restricts the 3x3 output convs to the maps the policy actually reads."""

    def __init__(self, in_channels_list, out_channels):
        super().__init__()
        self.inner_blocks = nn.ModuleList([nn.Sequential(nn.Conv2d(c, out_channels, 1)) for c in in_channels_list])
        self.layer_blocks = nn.ModuleList(
            [nn.Sequential(nn.Conv2d(out_channels, out_channels, 3, padding=1)) for _ in in_channels_list])
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1)
                nn.init.constant_(m.bias, 0)

    def forward(self, feats, needed=None, pad_to=None, defer_output_bias=None, sparse_ncam=None):
        """pad_to: run every convolution with its output (and, for the 3x3 layer blocks, input) channel count zero-padded
        to this width and return the padded maps (pad channels are exact zeros).  MIOpen's bf16 NHWC kernels for C = 60
        are ~2x slower than for C = 64 (measured on MI355X, N = 256 at 128 x 128: forward + backward 5.2 ms vs 2.6 ms);
        the parameters keep the reference's shapes, the padding is a per-step F.pad that autograd slices back.

        Biases (bf16 CUDA maps): the lateral 1x1 convolutions run bias-free and their bias is added by the top-down kernel
        that reads the lateral map anyway (forward) / reduced by the kernel that reads its gradient anyway (backward).
        defer_output_bias=True also runs the 3x3 output convolutions bias-free: the consumer adds the bias to the token rows
        it gathers (ops.BuildContextFn) -- a level reads 6 % of the fine map.  Whenever the keyword is passed (True OR False)
        the return value is the pair (maps, {name: bias Parameter owed}); without it, the plain dict of maps.
        sparse_ncam (cameras per sample; only with defer_output_bias=True): the caller reads the output maps through
        ops.BuildContextFn gathers and passes each map's ops.SparseConvCtx on to them -- the return value is then the triple
        (maps, biases, {name: SparseConvCtx}) and the output convolutions' weight gradients come from the gathers (_LayerConv3x3Fn)."""
        names = list(feats.keys())
        xs = list(feats.values())
        C = self.inner_blocks[0][0].out_channels
        pad = 0 if pad_to is None else max(0, pad_to - C)

        def conv(m, x, wpad, with_bias, **kw):
            w = F.pad(m.weight, wpad) if pad else m.weight
            b = None
            if with_bias:
                b = F.pad(m.bias, (0, pad)) if pad else m.bias
            return F.conv2d(x, w, b, **kw)

        def inner(i, x, last):
            m = self.inner_blocks[i][0]
            wpad = (0, 0, 0, 0, 0, 0, 0, pad)
            Co = C + pad
            bf16 = x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16)
            fuse = (x.is_cuda and bf16 and Co % 4 == 0 and 256 % (Co // 4) == 0 and x.shape[-1] % 2 == 0 and x.shape[-2] % 2 == 0
                    and (last is None or (x.shape[-2] == 2 * last.shape[-2] and x.shape[-1] == 2 * last.shape[-1])))
            if not fuse:
                lat = conv(m, x, wpad, True)
                return lat if last is None else fpn_top_down(lat, last)
            if _fused_lateral_ok(x, Co, last):                              # convolution + bias + top-down add in one launch
                return _LateralTopDownFn.apply(x, F.pad(m.weight, wpad) if pad else m.weight, m.bias, last)
            return fpn_top_down(conv(m, x, wpad, False), last, m.bias)      # bias-free convolution, bias in the top-down kernel

        out_bias, out_ctx = {}, {}

        def layer(i, x):
            m = self.layer_blocks[i][0]
            defer = defer_output_bias and x.is_cuda
            if defer:
                out_bias[names[i]] = m.bias
            if (defer and sparse_ncam and O.SPARSE_FPN_WGRAD and C + pad == 64 and x.dtype == torch.bfloat16 and x.shape[1] == 64
                    and x.is_contiguous(memory_format=torch.channels_last) and x.shape[0] % sparse_ncam == 0):
                cc = O.SparseConvCtx(sparse_ncam)
                out_ctx[names[i]] = cc
                w = F.pad(m.weight, (0, 0, 0, 0, 0, pad, 0, pad)) if pad else m.weight
                return _LayerConv3x3Fn.apply(x, w, cc)
            return conv(m, x, (0, 0, 0, 0, 0, pad, 0, pad), not defer, padding=1)

        last = inner(len(xs) - 1, xs[-1], None)
        out = {}
        lowest = min(names.index(n) for n in needed) if needed is not None else 0
        if needed is None or names[-1] in needed:
            out[names[-1]] = layer(len(xs) - 1, last)
        for i in range(len(xs) - 2, lowest - 1, -1):
            last = inner(i, xs[i], last)
            if needed is None or names[i] in needed:
                out[names[i]] = layer(i, last)
        if defer_output_bias is None:
            return out
        return (out, out_bias, out_ctx) if sparse_ncam else (out, out_bias)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.avgpool = nn.AvgPool2d(stride) if stride > 1 else nn.Identity()
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if stride > 1 or inplanes != planes * 4:
            self.downsample = nn.Sequential(nn.AvgPool2d(stride), nn.Conv2d(inplanes, planes * 4, 1, bias=False),
                                            nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(self.avgpool(out)))
        return F.relu(out + (x if self.downsample is None else self.downsample(x)))


class SyntheticCLIPResNet50(nn.Module):
    """Random-init network with the layer structure of CLIP's ModifiedResNet RN50 (3-conv stem, avg-pool
    anti-aliased bottlenecks [3,4,6,3], width 64), returning res1..res5 with channels 64/256/512/1024/2048 at strides
    2/4/8/16/32 like model/utils/clip.py:22-43.  The real weights (openai/CLIP) are not available offline; the
    arithmetic cost and map shapes are the real ones.  Frozen; runs on MIOpen."""

    def __init__(self, width=64, layers=(3, 4, 6, 3)):
        super().__init__()
        self.conv1 = nn.Conv2d(3, width // 2, 3, stride=2, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(width // 2)
        self.conv2 = nn.Conv2d(width // 2, width // 2, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width // 2)
        self.conv3 = nn.Conv2d(width // 2, width, 3, padding=1, bias=False)
        self.bn3 = nn.BatchNorm2d(width)
        self.avgpool = nn.AvgPool2d(2)
        self._inplanes = width
        self.layer1 = self._make_layer(width, layers[0])
        self.layer2 = self._make_layer(width * 2, layers[1], stride=2)
        self.layer3 = self._make_layer(width * 4, layers[2], stride=2)
        self.layer4 = self._make_layer(width * 8, layers[3], stride=2)

    def _make_layer(self, planes, blocks, stride=1):
        layers = [_Bottleneck(self._inplanes, planes, stride)]
        self._inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(_Bottleneck(self._inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.relu(self.bn2(self.conv2(x)))
        x0 = F.relu(self.bn3(self.conv3(x)))
        x1 = self.layer1(self.avgpool(x0))
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        x4 = self.layer4(x3)
        return {"res1": x0, "res2": x1, "res3": x2, "res4": x3, "res5": x4}


class ClipNormalize(nn.Module):
    """CLIP's preprocessing normalisation (clip_transforms.transforms[-1], model/utils/clip.py:19)."""

    def __init__(self):
        super().__init__()
        self.register_buffer("mean", torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(1, 3, 1, 1), persistent=False)
        self.register_buffer("std", torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(1, 3, 1, 1), persistent=False)

    def forward(self, x):
        return (x - self.mean) / self.std


def load_synthetic_clip():
    return SyntheticCLIPResNet50(), ClipNormalize()


def bn_scale_shift(x, bn, partial=None):
    """(2, C) fp32 [scale | shift] of BatchNorm2d `bn` for the bf16 channels_last activation x: batch statistics when
    bn.training (one read of x -- or none, when the producing a3d_conv1x1_bn_fwd already left its `partial` sums), running
    statistics otherwise; the finalize kernel also applies the running-stat update."""
    N, C, H, W = x.shape
    rows = N * H * W
    dev = x.device
    st = O.L.stream()
    scale = torch.empty((2, C), device=dev, dtype=torch.float32)
    train = 1 if bn.training else 0
    nslab = 1
    if train and partial is None:
        nslab = O.L.load().a3d_bn_nslab(rows, C)
        partial = torch.empty((nslab, 2, C), device=dev, dtype=torch.float32)
        O.L.call("a3d_bn_stats", x.data_ptr(), partial.data_ptr(), rows, C, nslab, st)
    elif train:
        nslab = partial.shape[0]
    O.L.call("a3d_bn_finalize", None if not train else partial.data_ptr(), nslab, rows, C, float(bn.eps),
             float(bn.momentum if bn.momentum is not None else 0.1), bn.weight.data_ptr(), bn.bias.data_ptr(),
             bn.running_mean.data_ptr(), bn.running_var.data_ptr(), scale[0].data_ptr(), scale[1].data_ptr(), train, st)
    return scale


def conv1x1_bn(x, conv, in_scale=None, in_relu=False, want_stats=True):
    """1x1 stride-1 convolution of a bf16 channels_last activation as the fused GEMM of csrc/conv1x1.hip: optional
    BatchNorm-apply (+ ReLU) of the producer on the input (`in_scale` = bn_scale_shift of that layer), and the partial
    statistics of the output for the BatchNorm that follows.  Returns (y, partial or None)."""
    N, K, H, W = x.shape
    Cout = conv.weight.shape[0]
    assert x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last)
    assert conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.bias is None and conv.weight.dtype == torch.bfloat16
    M = N * H * W
    w2 = conv.weight.reshape(Cout, K)
    assert w2.is_contiguous()
    y = torch.empty((N, Cout, H, W), device=x.device, dtype=torch.bfloat16, memory_format=torch.channels_last)
    partial = None
    if want_stats:
        partial = torch.empty((O.L.load().a3d_conv1x1_nslab(M, K, Cout), 2, Cout), device=x.device, dtype=torch.float32)
    O.L.call("a3d_conv1x1_bn_fwd", x.data_ptr(), w2.data_ptr(), None if in_scale is None else in_scale[0].data_ptr(),
             None if in_scale is None else in_scale[1].data_ptr(), 1 if in_relu else 0, y.data_ptr(),
             None if partial is None else partial.data_ptr(), M, K, Cout, O.L.stream())
    return y, partial


def conv3x3_serves(x, conv):
    """the 3x3 convolution `conv` on the bf16 channels_last map x is one a3d_conv3x3_bn_fwd serves (stride 1, padding 1, no bias,
    32 -> 32 / 32 -> 64 / 64 -> 64 channels, H % 8 == 0, W % 32 == 0)"""
    return bool(conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1) and
                conv.groups == 1 and conv.bias is None and conv.weight.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and
                O.L.load().a3d_conv3x3_serves(x.shape[1], conv.weight.shape[0], x.shape[2], x.shape[3]))


def conv3x3_bn(x, conv, in_scale=None, in_relu=False, want_stats=True):
    """3x3 stride-1 padding-1 convolution of a bf16 channels_last activation as the implicit GEMM of csrc/conv3x3.hip: optional
    BatchNorm-apply (+ ReLU) of the producer on the input (`in_scale` = bn_scale_shift of that layer; the zero padding is applied
    after it, as F.conv2d pads the normalised map), and the partial statistics of the output for the BatchNorm that follows.
    Returns (y, partial or None)."""
    N, Cin, H, W = x.shape
    Cout = conv.weight.shape[0]
    assert x.is_contiguous(memory_format=torch.channels_last) and conv3x3_serves(x, conv)
    wt = conv.weight
    assert wt.is_contiguous(memory_format=torch.channels_last)        # [Cout][3][3][Cin] in memory
    y = torch.empty((N, Cout, H, W), device=x.device, dtype=torch.bfloat16, memory_format=torch.channels_last)
    partial = None
    if want_stats:
        partial = torch.empty((O.L.load().a3d_conv3x3_nslab(N, H, W, Cin, Cout), 2, Cout), device=x.device, dtype=torch.float32)
    O.L.call("a3d_conv3x3_bn_fwd", x.data_ptr(), wt.data_ptr(), None if in_scale is None else in_scale[0].data_ptr(),
             None if in_scale is None else in_scale[1].data_ptr(), 1 if in_relu else 0, y.data_ptr(),
             None if partial is None else partial.data_ptr(), N, H, W, Cin, Cout, O.L.stream())
    return y, partial


FUSED_STEM = os.environ.get("A3D_FUSED_STEM", "1") == "1"


def stem_serves(x, conv, normalize):
    """conv1 of the CLIP stem on the raw fp32 images x with ClipNormalize in front is one a3d_stem_conv_bn_fwd serves"""
    return bool(FUSED_STEM and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3 and x.is_contiguous() and
                isinstance(normalize, ClipNormalize) and isinstance(conv, nn.Conv2d) and conv.in_channels == 3 and conv.out_channels == 32 and
                conv.kernel_size == (3, 3) and conv.stride == (2, 2) and conv.padding == (1, 1) and conv.bias is None and
                conv.weight.dtype == torch.bfloat16 and O.L.load().a3d_stem_conv_nslab(x.shape[0], x.shape[2], x.shape[3]) > 0)


def stem_conv_bn(x, conv, normalize, want_stats=True):
    """CLIP normalisation + the stem's strided conv1 + the partial statistics of its output in one launch (csrc/stem.hip): the
    normalised bf16 image is never written, the 268 MB output never re-read for its BatchNorm.  Returns (y, partial or None)."""
    N, _, H, W = x.shape
    w2 = conv.weight.reshape(32, 27)                 # [co][ci][kh][kw] whatever the memory format of the 4-d weight
    if not w2.is_contiguous():
        w2 = conv.weight.contiguous().reshape(32, 27)
    y = torch.empty((N, 32, H // 2, W // 2), device=x.device, dtype=torch.bfloat16, memory_format=torch.channels_last)
    partial = None
    if want_stats:
        partial = torch.empty((O.L.load().a3d_stem_conv_nslab(N, H, W), 2, 32), device=x.device, dtype=torch.float32)
    O.L.call("a3d_stem_conv_bn_fwd", x.data_ptr(), normalize.mean.reshape(-1).contiguous().data_ptr(),
             normalize.std.reshape(-1).contiguous().data_ptr(), w2.data_ptr(), y.data_ptr(), None if partial is None else partial.data_ptr(),
             N, H, W, O.L.stream())
    return y, partial


def bn_act(x, bn, relu=True, residual=None, pool=False, keep_full=True, partial=None, residual_scale=None, out=None):
    """Fused BatchNorm2d (batch statistics when bn.training, running-stat update) + optional residual add + ReLU on a
    bf16 channels_last activation (vision.hip).  Three launches: stats (skipped when the producer left `partial` sums),
    finalize, apply.  pool=True also applies the nn.AvgPool2d(2) that follows in the CLIP ResNet inside the apply kernel and
    returns (full, pooled); full is None when keep_full=False.  bn=None: no normalisation (plain 2x2 average pool of x).
    out: a preallocated tensor for the full-resolution result (the backbone's returned maps, run_frozen_backbone(out=...))."""
    N, C, H, W = x.shape
    if out is not None:
        assert out.shape == x.shape and out.dtype == x.dtype and out.is_contiguous(memory_format=torch.channels_last)
    assert x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last)
    if residual is not None:
        assert residual.shape == x.shape and residual.dtype == x.dtype and \
            residual.is_contiguous(memory_format=torch.channels_last)
    rows = N * H * W
    dev = x.device
    st = O.L.stream()
    sc_ptr = sh_ptr = None
    if bn is not None:
        scale = bn_scale_shift(x, bn, partial)
        sc_ptr, sh_ptr = scale[0].data_ptr(), scale[1].data_ptr()
    res_ptr = None if residual is None else residual.data_ptr()
    if not pool:
        y = out if out is not None else torch.empty_like(x)
        O.L.call("a3d_bn_apply", x.data_ptr(), res_ptr, None if residual_scale is None else residual_scale[0].data_ptr(),
                 None if residual_scale is None else residual_scale[1].data_ptr(), sc_ptr, sh_ptr, y.data_ptr(), rows, C,
                 1 if relu else 0, st)
        return y
    assert residual_scale is None, "the pooled apply takes a materialised residual"
    y = (out if out is not None else torch.empty_like(x)) if keep_full else None
    yp = torch.empty((N, C, H // 2, W // 2), device=dev, dtype=x.dtype, memory_format=torch.channels_last)
    O.L.call("a3d_bn_apply_pool2", x.data_ptr(), res_ptr, sc_ptr, sh_ptr, None if y is None else y.data_ptr(), yp.data_ptr(),
             N, H, W, C, 1 if relu else 0, st)
    return y, yp


def _pool2_ok(m, t):
    """m is the nn.AvgPool2d(2) of a stride-2 CLIP bottleneck and t has even spatial size (else: torch's pool)."""
    return isinstance(m, nn.AvgPool2d) and m.kernel_size in (2, (2, 2)) and t.shape[-1] % 2 == 0 and t.shape[-2] % 2 == 0


def fused_frozen_backbone_forward(bb, x, stem=None, out=None):
    """SyntheticCLIPResNet50.forward with MIOpen bf16 NHWC convolutions and the fused BatchNorm of vision.hip.
    Same dataflow as the module's own forward (CLIP ModifiedResNet, model/utils/clip.py:28-43); every AvgPool2d(2) is
    folded into the BatchNorm-apply kernel that produces its input (the block output also feeds the next block's
    downsample branch pooled, so that kernel emits both).  out: {res1..res5} preallocated maps the five returned maps are written
    into by the kernels that produce them (no copy)."""
    conv = lambda m, t: F.conv2d(t, m.weight, None, m.stride, m.padding)
    fuse3 = FUSED_CONV3X3
    o_ = (lambda k: None) if out is None else (lambda k, _maps=out: _maps[k])        # (`out` is rebound as a local further down)

    def conv3(m, t, bn_in, p_in, want_stats):
        """3x3 convolution m of relu(bn_in(t)) (t = the raw output of the previous convolution, p_in its partial statistics or
        None) + the partial statistics of its own output: one a3d_conv3x3_bn_fwd where it serves the shape -- relu(bn_in(t)) is
        then never materialised -- else BatchNorm-apply + MIOpen"""
        if fuse3 and conv3x3_serves(t, m):
            return conv3x3_bn(t, m, in_scale=bn_scale_shift(t, bn_in, p_in), in_relu=True, want_stats=want_stats)
        return conv(m, bn_act(t, bn_in, partial=p_in)), None

    # stem: (raw output of conv1, partial statistics) from a3d_stem_conv_bn_fwd (normalisation + convolution + statistics in one
    # launch, stem_conv_bn), else the library's convolution of the normalised bf16 image x
    c1, p1 = stem if stem is not None else (conv(bb.conv1, x), None)
    c2, p2 = conv3(bb.conv2, c1, bb.bn1, p1, bb.bn2.training)
    c3, p3 = conv3(bb.conv3, c2, bb.bn2, p2, bb.bn3.training)
    if _pool2_ok(bb.avgpool, c3):
        x0, x = bn_act(c3, bb.bn3, pool=True, partial=p3, out=o_("res1"))
    else:
        x0 = bn_act(c3, bb.bn3, partial=p3, out=o_("res1"))
        x = bb.avgpool(x0)
    outs = [x0]
    bns = [bb.bn1, bb.bn2, bb.bn3]
    blocks = [blk for layer in (bb.layer1, bb.layer2, bb.layer3, bb.layer4) for blk in layer]
    last_of_layer = {id(layer[-1]) for layer in (bb.layer1, bb.layer2, bb.layer3, bb.layer4)}
    x_pooled = None                                    # AvgPool2d(2)(x), when the producer of x already emitted it
    fuse = FUSED_CONV1X1

    def fused_ok(m, t):
        """the 1x1 convolution m on input t goes through a3d_conv1x1_bn_fwd (the shapes its streaming kernel serves)"""
        return bool(fuse and m.kernel_size == (1, 1) and m.stride == (1, 1) and
                    O.L.load().a3d_conv1x1_streams(t.shape[1], m.weight.shape[0]))

    def conv1(m, t, **kw):
        """1x1 convolution + the partial statistics of its output (None on the MIOpen path)"""
        if fused_ok(m, t):
            return conv1x1_bn(t, m, **kw)
        return conv(m, t), None

    for bi, blk in enumerate(blocks):
        c1, p1 = conv1(blk.conv1, x, want_stats=blk.bn1.training)
        c2, p2 = conv3(blk.conv2, c1, blk.bn1, p1, blk.bn2.training)
        no_pool = isinstance(blk.avgpool, nn.Identity) or (isinstance(blk.avgpool, nn.AvgPool2d) and blk.avgpool.kernel_size in (1, (1, 1)))
        if no_pool and fused_ok(blk.conv3, c2):
            # BatchNorm-apply + ReLU of bn2 ride on conv3's operand load: c2 is never rewritten
            o3, p3 = conv1x1_bn(c2, blk.conv3, in_scale=bn_scale_shift(c2, blk.bn2, p2), in_relu=True, want_stats=blk.bn3.training)
        else:
            if _pool2_ok(blk.avgpool, c2):
                out = bn_act(c2, blk.bn2, pool=True, keep_full=False, partial=p2)[1]
            else:
                out = blk.avgpool(bn_act(c2, blk.bn2, partial=p2))
            o3, p3 = conv1(blk.conv3, out, want_stats=blk.bn3.training)
        if blk.downsample is not None:
            dpool = blk.downsample[0]
            if x_pooled is not None:
                xin = x_pooled
            elif _pool2_ok(dpool, x):
                xin = bn_act(x, None, relu=False, pool=True, keep_full=False)[1]
            elif isinstance(dpool, nn.AvgPool2d) and dpool.kernel_size in (1, (1, 1)):
                xin = x                                # AvgPool2d(1) is the identity
            else:
                xin = dpool(x)
            cd, pd = conv1(blk.downsample[1], xin, want_stats=blk.downsample[2].training)
            bns.append(blk.downsample[2])
        nxt = blocks[bi + 1] if bi + 1 < len(blocks) else None
        want_pooled = nxt is not None and nxt.downsample is not None and _pool2_ok(nxt.downsample[0], o3)
        idn_scale = None
        if blk.downsample is None:
            idn = x
        elif FOLD_DOWNSAMPLE_BN and not want_pooled:
            # the branch's BatchNorm rides in the block's final apply: bn_d(cd) is never materialised
            idn, idn_scale = cd, bn_scale_shift(cd, blk.downsample[2], pd)
        else:
            idn = bn_act(cd, blk.downsample[2], relu=False, partial=pd)
        o_x = o_("res%d" % (len(outs) + 1)) if id(blk) in last_of_layer else None      # a layer's last block writes the returned map
        if want_pooled:
            x, x_pooled = bn_act(o3, blk.bn3, relu=True, residual=idn, pool=True, partial=p3, out=o_x)
        else:
            x, x_pooled = bn_act(o3, blk.bn3, relu=True, residual=idn, partial=p3, residual_scale=idn_scale, out=o_x), None
        bns += [blk.bn1, blk.bn2, blk.bn3]
        if id(blk) in last_of_layer:
            outs.append(x)
    if bb.training:
        torch._foreach_add_([b.num_batches_tracked for b in bns], 1)
    return dict(zip(["res1", "res2", "res3", "res4", "res5"], outs))


FUSED_BN = os.environ.get("A3D_FUSED_BN", "1") == "1"
# BatchNorm of the bottleneck's downsample branch applied inside the block's final BatchNorm-apply + add + ReLU kernel (a second
# scale / shift pair on the residual operand) instead of its own pass.  A3D_FOLD_DS_BN=0: the branch's map is materialised (A/B).
FOLD_DOWNSAMPLE_BN = os.environ.get("A3D_FOLD_DS_BN", "1") not in ("0", "", "off")
# The backbone's 1x1 convolutions of the HBM-bound layers (1 and 2: K <= 256, the shapes a3d_conv1x1_streams accepts) through
# a3d_conv1x1_bn_fwd, with BatchNorm-apply of the producer and the statistics of the consumer folded into the GEMM; the deep,
# compute-bound layers stay on MIOpen (profiles/r04_conv1x1_layers.json).  A3D_FUSED_CONV1X1=0: MIOpen everywhere (A/B).
FUSED_CONV1X1 = os.environ.get("A3D_FUSED_CONV1X1", "1") not in ("0", "", "off")
# The 3x3 convolutions with <= 64 channels (the stem's conv2 / conv3, layer1's conv2) through a3d_conv3x3_bn_fwd, with BatchNorm-apply
# of the producer and the statistics of the consumer folded in; wider layers stay on MIOpen.  A3D_FUSED_CONV3X3=0: MIOpen (A/B).
FUSED_CONV3X3 = os.environ.get("A3D_FUSED_CONV3X3", "1") not in ("0", "", "off")


def normalize_to_nhwc_bf16(x, normalize):
    """ClipNormalize + channels-last + bf16 cast of the raw images (N, 3, H, W) fp32 in one kernel (vision.hip)."""
    x = x.float().contiguous()
    N, _, H, W = x.shape
    y = torch.empty((N, 3, H, W), device=x.device, dtype=torch.bfloat16, memory_format=torch.channels_last)
    O.L.call("a3d_rgb_normalize_nhwc_bf16", x.data_ptr(), normalize.mean.data_ptr(), normalize.std.data_ptr(), y.data_ptr(), N, H, W,
             O.L.stream())
    return y


def run_frozen_backbone(backbone, x, dtype, keep_dtype=False, normalize=None, out=None):
    """Forward of the frozen backbone under no_grad.  `normalize`: the ClipNormalize module when `x` are the RAW images
    (None: already normalised).  For a reduced dtype the convolution weights are converted ONCE
    (the backbone is frozen, so there is no master copy to keep) instead of being re-cast by autocast at every step;
    BatchNorm keeps fp32 parameters / running statistics and, as in the reference's train() mode, batch statistics.
    On the GPU with bf16 the BatchNorm + ReLU + residual chain runs as the fused HIP kernels of vision.hip.
    out: {res1..res5} preallocated maps (as a previous call returned them) to write the result into -- the fused path's kernels
    write them directly, the other paths copy."""
    def deliver(feats):
        if out is None:
            return feats
        for k, v in feats.items():
            if v.data_ptr() != out[k].data_ptr():
                out[k].copy_(v)
        return out
    fused = FUSED_BN and x.is_cuda and dtype == torch.bfloat16 and isinstance(backbone, SyntheticCLIPResNet50)
    if normalize is not None and not (fused and isinstance(normalize, ClipNormalize) and (x.shape[-1] * x.shape[-2]) % 4 == 0):
        x = normalize(x).contiguous(memory_format=torch.channels_last)
        normalize = None
    if dtype == torch.float32:
        return deliver(backbone(x))
    if getattr(backbone, "_conv_dtype", None) != dtype:
        for m in backbone.modules():
            if isinstance(m, nn.Conv2d):
                m.to(dtype)
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
        backbone._conv_dtype = dtype
    if fused:
        if normalize is not None and stem_serves(x, backbone.conv1, normalize):
            feats = fused_frozen_backbone_forward(backbone, None, stem=stem_conv_bn(x, backbone.conv1, normalize, want_stats=backbone.bn1.training),
                                                  out=out if keep_dtype else None)
        else:
            xb = normalize_to_nhwc_bf16(x, normalize) if normalize is not None else x.to(dtype).contiguous(memory_format=torch.channels_last)
            feats = fused_frozen_backbone_forward(backbone, xb, out=out if keep_dtype else None)
    else:
        feats = backbone(x.to(dtype))
    return deliver(feats if keep_dtype else {k: v.float() for k, v in feats.items()})
