"""ChainedDiffuser trajectory DDPM on the MI355X hot path.

Drop-in for `model.DiffusionPlanner` (model/trajectory_optimization/diffusion_model.py:15-324) and its
`DiffusionHead` (diffusion_head.py:10-363, on top of model/utils/encoder.py): same constructor keywords, same
`forward(gt_trajectory, trajectory_mask, rgb_obs, pcd_obs, instruction, curr_gripper, goal_gripper, run_inference)`
and `compute_trajectory(...)`, same parameter names (checkpoints interchange).

What is restructured, with identical results:
  * everything that does not depend on the denoising step -- image encoding, instruction encoding, vision->language
    attention, gripper tokens and the K/V projections (+RoPE) of all cross-attention layers -- is computed ONCE per
    trajectory batch (`encode_context`, `build_kv_cache`) instead of once per step;
  * one denoise step is a fixed sequence of stream-ordered launches (no host sync: the boolean-mask scatter becomes a
    masked select in the fused DDPM-step kernel, timestep tables live on the device), so the whole 100-step loop is
    captured in a hipGraph and replayed (`compute_trajectory(..., use_graph=True)`).
The DDPM schedules restate diffusers' DDPMScheduler (third-party, un-pinned -> parity unpinned).
Additive keyword arguments: `noise`, `timesteps` (training) and `init_noise`, `step_noise` (sampling) inject the random
draws; `visual_tokens` bypasses the backbone + FPN.  Training-mode dropout (p = 0.1 in every ParallelAttentionLayer --
attention weights, residual branches, FFN -- and in the traj_encoder / regressor MLPs: layers.py:10,
diffusion_head.py:46,183,193) runs on a device-resident Philox stream (csrc/dropout.hip, attention kernels): same
distribution as the reference's torch generator, not the same draws.  The additive constructor keyword `dropout`
(default 0.1 = the reference) sets that probability; 0.0 disables it.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops as O
from .act3d import broadcast_row
from .nn import FeaturePyramidNetwork, ParallelAttention, load_synthetic_clip, run_frozen_backbone


# ------------------------------------------------------------------------------------------------ DDPM tables
class DDPMTables:
    """alphas_cumprod and per-step posterior coefficients of the two schedulers the reference builds
    (diffusion_model.py:51-60), as device tables.  Follows Ho et al. 2020 eq. 6-7 with diffusers' defaults
    (beta_start 1e-4, beta_end 0.02, variance "fixed_small", clip_sample, prediction_type "sample")."""

    def __init__(self, T, device):
        self.T = T
        betas_pos = torch.linspace(0.0001 ** 0.5, 0.02 ** 0.5, T, dtype=torch.float32) ** 2

        def alpha_bar(s):
            return math.cos((s + 0.008) / 1.008 * math.pi / 2) ** 2

        betas_rot = torch.tensor([min(1 - alpha_bar((i + 1) / T) / alpha_bar(i / T), 0.999) for i in range(T)],
                                 dtype=torch.float32)
        acp_pos, acp_rot = torch.cumprod(1.0 - betas_pos, 0), torch.cumprod(1.0 - betas_rot, 0)
        self.acp_pos, self.acp_rot = acp_pos.to(device), acp_rot.to(device)
        self.coef_pos, self.coef_rot = self._coef(acp_pos).to(device), self._coef(acp_rot).to(device)

    def _coef(self, acp):
        one = torch.tensor(1.0)
        rows = []
        for t in range(self.T):
            a_t, a_prev = acp[t], (acp[t - 1] if t > 0 else one)
            b_t, b_prev = 1 - a_t, 1 - a_prev
            cur_a = a_t / a_prev
            cur_b = 1 - cur_a
            var = torch.clamp(b_prev / b_t * cur_b, min=1e-20)
            rows.append(torch.stack([(a_prev ** 0.5 * cur_b) / b_t, cur_a ** 0.5 * b_prev / b_t,
                                     var ** 0.5 if t > 0 else torch.tensor(0.0)]))
        return torch.stack(rows).float().contiguous()


# ------------------------------------------------------------------------------------------------ pose <-> signal
def pose_to_signal(pose, bounds=None):
    """[xyz | quaternion (w, x, y, z) | extra] -> [normalised xyz | 6D rotation | extra] in one launch
    (diffusion_model.py:187-212; csrc/diffusion.hip).  bounds: (2, 3) device tensor or None (xyz untouched).  No grad."""
    x = O._c(pose.detach().float())
    n, D = x.numel() // x.shape[-1], x.shape[-1]
    out = torch.empty(x.shape[:-1] + (D + 2,), device=x.device, dtype=torch.float32)
    O.L.call("a3d_pose_to_signal", x.data_ptr(), None if bounds is None else bounds.data_ptr(), out.data_ptr(), n, D - 7,
             O.L.stream())
    return out


def signal_to_pose(signal, bounds=None):
    """inverse map (diffusion_model.py:192-195,214-230)"""
    x = O._c(signal.detach().float())
    n, D = x.numel() // x.shape[-1], x.shape[-1]
    out = torch.empty(x.shape[:-1] + (D - 2,), device=x.device, dtype=torch.float32)
    O.L.call("a3d_signal_to_pose", x.data_ptr(), None if bounds is None else bounds.data_ptr(), out.data_ptr(), n, D - 9,
             O.L.stream())
    return out


# ------------------------------------------------------------------------------------------------ prediction head
class DiffusionHead(nn.Module):

    def __init__(self, backbone="clip", image_size=(256, 256), embedding_dim=60, output_dim=7, num_attn_heads=8,
                 num_vis_ins_attn_layers=2, num_query_cross_attn_layers=6, use_instruction=False, use_goal=False,
                 use_sigma=False, feat_scales_to_use=1, attn_rounds=1, weight_tying=False,
                 rotation_parametrization='quat', dropout=0.1, dropout_seed=0):
        super().__init__()
        if use_sigma:
            # DiffusionPlanner never forwards use_sigma (diffusion_model.py:37-50): unreachable through the model API
            raise NotImplementedError("use_sigma=True (a learned time embedding, encoder.py:68-76) is not implemented")
        assert feat_scales_to_use in (1, 2, 3, 4) and attn_rounds >= 1
        self.attn_rounds, self.feat_scales = attn_rounds, feat_scales_to_use
        R = attn_rounds * feat_scales_to_use        # one module set per (round, scale) iteration, diffusion_head.py:53-199
        self.image_size = tuple(image_size)
        self.use_instruction, self.use_goal = use_instruction, use_goal
        self.rotation_parametrization = rotation_parametrization
        self.num_attn_heads = num_attn_heads
        self.dropout_p = dropout
        if rotation_parametrization == '6D':
            output_dim += 2
        E = embedding_dim
        # --- Encoder base (model/utils/encoder.py:14-73)
        self.backbone, self.normalize = load_synthetic_clip()
        for p in self.backbone.parameters():
            p.requires_grad = False
        self.backbone_dtype = torch.float32
        self.fpn_dtype = torch.float32           # set to torch.bfloat16 to run the FPN convolutions under autocast (as Act3D.fpn_dtype)
        self.feature_pyramid = FeaturePyramidNetwork([64, 256, 512, 1024, 2048], E)
        self.feature_map_pyramid = ['res3', 'res1', 'res1', 'res1'] if self.image_size == (256, 256) else ['res2', 'res1', 'res1', 'res1']
        self.downscaling_factor_pyramid = [8, 2, 2, 2] if self.image_size == (256, 256) else [4, 2, 2, 2]
        self.curr_gripper_embed = nn.Embedding(1, E)
        self.goal_gripper_embed = nn.Embedding(1, E)
        self.instruction_encoder = nn.Linear(512, E)
        # --- DiffusionHead (diffusion_head.py:41-199)
        self.traj_encoder = nn.Sequential(nn.Linear(9, E), nn.ReLU(), nn.Dropout(dropout), nn.Linear(E, E))
        self.curr_gripper_encoder = nn.Linear(output_dim, E)
        if use_goal:
            self.goal_gripper_encoder = nn.Linear(output_dim, E)
        common = dict(d_model=E, n_heads=num_attn_heads, dropout=dropout, self_attention2=False, cross_attention1=True,
                      cross_attention2=False)
        def stack(make):                      # the reference shares ONE module across all iterations iff weight_tying
            if weight_tying:
                m = make()
                return nn.ModuleList([m for _ in range(R)])
            return nn.ModuleList([make() for _ in range(R)])

        if use_instruction:
            self.vl_attention = stack(lambda: ParallelAttention(num_layers=num_vis_ins_attn_layers, self_attention1=False, **common))
        self.traj_lang_attention = stack(lambda: ParallelAttention(num_layers=1, self_attention1=False, rotary_pe=False,
                                                                   apply_ffn=False, **common))
        self.traj_attention = stack(lambda: ParallelAttention(num_layers=num_query_cross_attn_layers - 2, self_attention1=True,
                                                              rotary_pe=True, use_adaln=True, **common))
        self.pos_attention = stack(lambda: ParallelAttention(num_layers=2, self_attention1=True, rotary_pe=True,
                                                             use_adaln=True, **common))
        self.rot_attention = stack(lambda: ParallelAttention(num_layers=2, self_attention1=True, rotary_pe=True,
                                                             use_adaln=True, **common))
        self.pos_regressor = nn.ModuleList([nn.Sequential(nn.Linear(E, E), nn.ReLU(), nn.Dropout(dropout), nn.Linear(E, 3))
                                            for _ in range(R)])
        self.rot_regressor = nn.ModuleList([nn.Sequential(nn.Linear(E, E), nn.ReLU(), nn.Dropout(dropout),
                                                          nn.Linear(E, output_dim - 3)) for _ in range(R)])
        self._sem_cache = {}
        # dropout sites are named after the modules (ops.site_id); generator state {seed, forward-pass counter} on the device
        for name, mod in self.named_modules():
            if hasattr(mod, "site_base"):
                mod.site_base = O.site_id(name)
        self._mlp_sites = {n: O.site_id(n, 4) for n in ["traj_encoder"] + [f"{k}_regressor.{l}" for k in ("pos", "rot")
                                                                          for l in range(R)]}
        self.register_buffer("_drop_state", torch.tensor([dropout_seed, 0], dtype=torch.int64), persistent=False)

    def begin_dropout(self):
        """DropCtx of one training forward pass (None in eval mode or with dropout=0): snapshots the device generator
        state and advances it by one, both as stream-ordered kernels (capturable)."""
        if not self.training or self.dropout_p <= 0:
            return None
        snap = self._drop_state.clone()
        O.L.call("a3d_rng_advance", self._drop_state.data_ptr(), 1, O.L.stream())
        return O.DropCtx(snap, self.dropout_p)

    # ---- vision (adjacent): one scale
    def encode_images(self, rgb, pcd_norm, maps=None):
        """encoder.py:115-167: FPN tokens (B, ncam*h*w, E) of the scales the head uses -- one tensor for
        feat_scales_to_use = 1 (the res3 map at 1/8), a list [res3 @ 1/8, res1 @ 1/2, ...] otherwise."""
        B, ncam = rgb.shape[:2]
        return self.tokens_from_backbone_maps(maps if maps is not None else self.backbone_maps(rgb), B, ncam)

    def backbone_maps(self, rgb, out=None):
        """The frozen half of encode_images (normalize -> backbone under no_grad): {res1..res5} of the B * ncam views; `out`:
        preallocated maps to write into.  No gradient flows through it and its weights never change, so a training loop may compute
        the maps of batch k + 1 while step k runs (engine.GraphedStep(prefetch=...)) and pass them to encode_images(maps=...)."""
        x = rgb.flatten(0, 1)
        low = self.fpn_dtype != torch.float32 and x.is_cuda
        with torch.no_grad():
            return run_frozen_backbone(self.backbone, x, self.backbone_dtype, keep_dtype=low, normalize=self.normalize, out=out)

    def tokens_from_backbone_maps(self, feats, B, ncam):
        """The FPN + token layout half of encode_images on the backbone's maps {res1..res5} of the B * ncam views (bf16 maps when
        fpn_dtype is bf16, fp32 maps otherwise)."""
        low = self.fpn_dtype != torch.float32 and next(iter(feats.values())).is_cuda
        names = self.feature_map_pyramid[:self.feat_scales]
        E_ = self.curr_gripper_embed.weight.shape[1]
        if low:
            # round 6: the FPN in bf16 on the backbone's bf16 maps (channels padded to a multiple of 64 for MIOpen, the lateral biases
            # folded into the top-down kernel) -- the fp32 path converted all five backbone maps to fp32 (277 MB for res1 alone at the
            # script shape) and ran the FPN's convolutions, forward and backward, on MIOpen's fp32 kernels; only the map(s) the head
            # reads are converted now
            with torch.autocast("cuda", dtype=self.fpn_dtype):
                pyr = self.feature_pyramid(feats, needed=sorted(set(names)), pad_to=(E_ + 63) // 64 * 64)
        else:
            pyr = self.feature_pyramid(feats, needed=sorted(set(names)))
        toks = {}
        for name in set(names):
            fm = pyr[name]
            n, E, h, w = fm.shape
            tk = fm.permute(0, 2, 3, 1).reshape(B, ncam * h * w, E)
            toks[name] = tk[..., :E_].float() if low else tk
        out = [toks[n] for n in names]
        return out[0] if self.feat_scales == 1 else out

    def _sem(self, Ln, E, device):
        key = (Ln, E, str(device))
        if key not in self._sem_cache:
            self._sem_cache[key] = O.sinusoidal_emb(torch.arange(Ln, device=device, dtype=torch.float32), E)
        return self._sem_cache[key]

    def encode_context(self, visual_tokens, ctx_xyz, instruction, curr_gripper, goal_gripper, drop=None):
        """Step-invariant context (diffusion_head.py:222-247, 290-323): returns (ctx (B,S,E), ctx_xyz (B,S,3), instr)."""
        B = visual_tokens.shape[0]
        instr = O.linear(instruction.float(), self.instruction_encoder) if self.use_instruction else None
        ctx = visual_tokens
        if self.use_instruction:
            ctx = self.vl_attention[0](ctx, None, instr, drop=drop)
        cg = O.linear(curr_gripper, self.curr_gripper_encoder)[:, None] + broadcast_row(self.curr_gripper_embed.weight, B, 1)
        extra, extra_xyz = [cg], [curr_gripper[:, None, :3]]
        if self.use_goal:
            gg = O.linear(goal_gripper, self.goal_gripper_encoder)[:, None] + broadcast_row(self.goal_gripper_embed.weight, B, 1)
            extra.append(gg)
            extra_xyz.append(goal_gripper[:, None, :3])
        ctx = O.BuildContextFn.apply(ctx, None, torch.cat(extra, dim=1))
        ctx_xyz = torch.cat([ctx_xyz] + extra_xyz, dim=1).contiguous()
        return ctx, ctx_xyz, instr

    def forward_tokens(self, trajectory, trajectory_mask, timestep, ctx, ctx_xyz, instr, drop=None):
        """The step-dependent part of DiffusionHead.forward (diffusion_head.py:214-219, 325-363) with autograd."""
        B, Ln, _ = trajectory.shape
        E = ctx.shape[-1]
        trajectory = trajectory.contiguous()
        ms = self._mlp_sites
        traj_feats = O.mlp(trajectory, self.traj_encoder[0], self.traj_encoder[3], drop=drop, site_hidden=ms["traj_encoder"])
        traj_xyz = trajectory[..., :3].contiguous()
        time_feats = O.sinusoidal_emb(timestep.float(), E)
        silu_t = O.SiLUFn.apply(time_feats)
        sem = self._sem(Ln, E, trajectory.device)
        if self.use_instruction:
            traj_feats = self.traj_lang_attention[0](traj_feats, trajectory_mask, instr, seq1_sem_pos=sem, drop=drop)
        kw = dict(seq1_xyz=traj_xyz, seq2_xyz=ctx_xyz, seq1_sem_pos=sem, silu_t=silu_t, drop=drop)
        # the context feeds the k | v projections of all 8 cross-attention layers: one shared gradient sink (ops.GradSink: the
        # projections' input and weight gradients run as ONE GEMM each when the last layer's backward has parked its rows)
        self._ctx_sinks = []
        ctx = O.attach_grad_sink(ctx, self._ctx_sinks)
        traj_feats = self.traj_attention[0](traj_feats, trajectory_mask, ctx, **kw)
        pos_feats = self.pos_attention[0](traj_feats, trajectory_mask, ctx, **kw)
        rot_feats = self.rot_attention[0](traj_feats, trajectory_mask, ctx, **kw)
        upd = torch.cat([O.mlp(pos_feats, self.pos_regressor[0][0], self.pos_regressor[0][3], drop=drop,
                               site_hidden=ms["pos_regressor.0"]),
                         O.mlp(rot_feats, self.rot_regressor[0][0], self.rot_regressor[0][3], drop=drop,
                               site_hidden=ms["rot_regressor.0"])], dim=-1)
        return O.TrajUpdateFn.apply(trajectory, upd)

    def forward_multi(self, trajectory, trajectory_mask, timestep, tokens, xyzs, instruction, curr_gripper, goal_gripper,
                      new_drop=None):
        """DiffusionHead.forward for attn_rounds x feat_scales_to_use > 1 (diffusion_head.py:249-275).  Iteration
        l = round * feat_scales + scale runs module set l on the SAME trajectory encoding / positions (the reference never
        carries traj_feats over, :286-288); only the prediction chains (xyz accumulates, rotation is replaced).  With a goal,
        scales > 0 attend to the (64 | 16) * L fine tokens nearest to the previous prediction (find_traj_nn ->
        a3d_traj_nn_topk).  tokens / xyzs: per-scale visual tokens (B, N_s, E) and normalised coordinates (B, N_s, 3).
        new_drop: callable returning a fresh DropCtx per iteration (None: no dropout).  Returns the list of predictions."""
        B, Ln, _ = trajectory.shape
        E = self.curr_gripper_embed.weight.shape[1]
        trajectory = trajectory.contiguous()
        ms = self._mlp_sites
        drop = new_drop() if new_drop is not None else None
        traj_feats0 = O.mlp(trajectory, self.traj_encoder[0], self.traj_encoder[3], drop=drop, site_hidden=ms["traj_encoder"])
        traj_xyz = trajectory[..., :3].contiguous()
        silu_t = O.SiLUFn.apply(O.sinusoidal_emb(timestep.float(), E))
        sem = self._sem(Ln, E, trajectory.device)
        instr = O.linear(instruction.float(), self.instruction_encoder) if self.use_instruction else None
        cg = O.linear(curr_gripper, self.curr_gripper_encoder)[:, None] + broadcast_row(self.curr_gripper_embed.weight, B, 1)
        extra, extra_xyz = [cg], [curr_gripper[:, None, :3]]
        if self.use_goal:
            gg = O.linear(goal_gripper, self.goal_gripper_encoder)[:, None] + broadcast_row(self.goal_gripper_embed.weight, B, 1)
            extra.append(gg)
            extra_xyz.append(goal_gripper[:, None, :3])
        extra, extra_xyz = torch.cat(extra, dim=1), torch.cat(extra_xyz, dim=1).contiguous()
        none = torch.empty((B, 0, E), device=trajectory.device, dtype=torch.float32)
        outs, prev = [], trajectory
        for rnd in range(self.attn_rounds):
            for scale in range(self.feat_scales):
                l = rnd * self.feat_scales + scale
                if l > 0 and new_drop is not None:
                    drop = new_drop()
                feats, xyz = tokens[scale], xyzs[scale]
                idx = None
                if self.use_goal and scale > 0:
                    idx = O.traj_nn_topk(outs[-1][..., :3], xyz, (64 if scale == 1 else 16) * Ln)
                if self.use_instruction:
                    ctx = O.BuildContextFn.apply(feats, idx, none) if idx is not None else feats
                    ctx = self.vl_attention[l](ctx, None, instr, drop=drop)
                    ctx = O.BuildContextFn.apply(ctx, None, extra)
                else:
                    ctx = O.BuildContextFn.apply(feats, idx, extra)
                ctx_xyz = O.gather_rows(xyz, idx, extra_xyz)
                tf = traj_feats0
                if self.use_instruction:
                    tf = self.traj_lang_attention[l](tf, trajectory_mask, instr, seq1_sem_pos=sem, drop=drop)
                kw = dict(seq1_xyz=traj_xyz, seq2_xyz=ctx_xyz, seq1_sem_pos=sem, silu_t=silu_t, drop=drop)
                tf = self.traj_attention[l](tf, trajectory_mask, ctx, **kw)
                pf = self.pos_attention[l](tf, trajectory_mask, ctx, **kw)
                rf = self.rot_attention[l](tf, trajectory_mask, ctx, **kw)
                upd = torch.cat([O.mlp(pf, self.pos_regressor[l][0], self.pos_regressor[l][3], drop=drop,
                                       site_hidden=ms[f"pos_regressor.{l}"]),
                                 O.mlp(rf, self.rot_regressor[l][0], self.rot_regressor[l][3], drop=drop,
                                       site_hidden=ms[f"rot_regressor.{l}"])], dim=-1)
                prev = O.TrajUpdateFn.apply(prev, upd)
                outs.append(prev)
        return outs

    # ---- inference with cached K/V
    def _cross_layers(self):
        out = []
        for stack in (self.traj_attention[0], self.pos_attention[0], self.rot_attention[0]):
            out.extend(stack.layers)
        return out

    @torch.no_grad()
    def build_kv_cache(self, ctx, ctx_xyz, instr):
        cache = {"ctx": [O.kv_cache_build(ctx, ctx_xyz, lay.cross_12, self.num_attn_heads) for lay in self._cross_layers()]}
        if self.use_instruction:
            cache["lang"] = O.kv_cache_build(instr, None, self.traj_lang_attention[0].layers[0].cross_12, self.num_attn_heads)
        return cache

    @torch.no_grad()
    def denoise_tokens_cached(self, trajectory, trajectory_mask, t, cache, time_tables):
        """One network evaluation at integer step t against the prebuilt K/V cache (no autograd, no host sync)."""
        B, Ln, _ = trajectory.shape
        H = self.num_attn_heads
        E = self.curr_gripper_embed.weight.shape[1]
        traj_feats = O.mlp(trajectory, self.traj_encoder[0], self.traj_encoder[3])
        traj_xyz = trajectory[..., :3].contiguous()
        silu_t = time_tables["silu"][t:t + 1].expand(B, E).contiguous()
        sem = self._sem(Ln, E, trajectory.device)
        if self.use_instruction:
            lay = self.traj_lang_attention[0].layers[0]
            q1 = O.AddRowsFn.apply(traj_feats, sem)
            traj_feats = O.attn_block_cached(q1, traj_feats, None, cache["lang"], lay.cross_12, lay.norm_12, H)
        li = 0

        def run_stack(x, stack):
            nonlocal li
            for lay in stack.layers:
                q1 = O.AddRowsFn.apply(x, sem)
                x = O.attn_block_cached(lay.adaln_12(q1, silu_t), x, traj_xyz, cache["ctx"][li], lay.cross_12, lay.norm_12, H)
                li += 1
                q1 = O.AddRowsFn.apply(x, sem)
                qk, vv = lay.adaln_1(q1, silu_t), lay.adaln_1(x, silu_t)
                x = O.attn_block(qk, qk, vv, x, traj_xyz, traj_xyz, trajectory_mask, lay.sa1, lay.norm_1, H)
                y = lay.adaln_ff1(x, silu_t)
                x = O.mlp(y, lay.ffn_12[0], lay.ffn_12[3], lay.norm_122)
            return x

        traj_feats = run_stack(traj_feats, self.traj_attention[0])
        keep = li
        pos_feats = run_stack(traj_feats, self.pos_attention[0])
        rot_feats = run_stack(traj_feats, self.rot_attention[0])
        upd = torch.cat([O.mlp(pos_feats, self.pos_regressor[0][0], self.pos_regressor[0][3]),
                         O.mlp(rot_feats, self.rot_regressor[0][0], self.rot_regressor[0][3])], dim=-1)
        return O.TrajUpdateFn.apply(trajectory, upd)


    # ---- inference, fused: 18 launches per network evaluation (csrc/denoise.hip)
    @torch.no_grad()
    def build_fused(self, ctx, ctx_xyz, instr, kmask, time_sin, Ln, with_persist=True):
        """Step-invariant state of the fused sampling path for one trajectory batch: the context K (fp16 hi | lo rows) / V (fp16
        hi / lo planes) of every cross-attention layer, the instruction tokens through traj_lang_attention's k | v projection, and
        the AdaLN modulation of every layer at every timestep (Linear(SiLU(sinusoidal(t))), layers.py:273-290).  Returns
        {"tensors": [...]} -- the list is what a captured graph must refresh in place -- plus per-layer pointer tables."""
        B, S, E = ctx.shape
        H = self.num_attn_heads
        dev = ctx.device
        Sp = O.ceil_to(S, 64)
        f4 = 4
        freq = O.rope_freq(E, dev)
        silu = F.silu(time_sin)                                         # (T, E)
        st = {"S": S, "Sp": Sp, "layers": [], "tensors": [], "freq": freq, "sem": self._sem(Ln, E, dev), "kmask": kmask}
        ctx = O._c(ctx)
        xyz = O._c(ctx_xyz.float())
        for lay in self._cross_layers():
            mha = lay.cross_12
            # the split-fp16 operand formats (csrc/denoise.hip header): K rows16, V planes16 with the ones channel -- projection, RoPE
            # and formatting in ONE launch per layer (rounds 2 - 5: a [B S, 2E] fp32 projection in HBM + two formatting passes)
            Kf = torch.empty((B, H, Sp, 32), device=dev, dtype=torch.float16)
            Vt = torch.empty((B, H, 2, 16, Sp), device=dev, dtype=torch.float16)
            O.L.call("a3d_proj_rope_split16", ctx.data_ptr(), E, mha.in_proj_weight.data_ptr() + E * E * f4, E,
                     mha.in_proj_bias.data_ptr() + E * f4, E, xyz.data_ptr(), 1.0, Kf.data_ptr(), None, 2,
                     None, 1.0, None, Vt.data_ptr(), 2 | 4, freq.data_ptr(), B, S, Sp, E, H, O.L.stream())
            mods = [O.linear2d(silu, a.modulation[1].weight, a.modulation[1].bias) for a in (lay.adaln_12, lay.adaln_1, lay.adaln_ff1)]
            st["layers"].append({"lay": lay, "Kf": Kf, "Vt": Vt, "mods": mods})
            st["tensors"] += [Kf, Vt] + mods
        st["lang_kv"] = None
        if self.use_instruction:
            mha = self.traj_lang_attention[0].layers[0].cross_12
            instr = O._c(instr)
            st["S_lang"] = instr.shape[1]
            st["lang_kv"] = O.linear_raw(instr.data_ptr(), E, mha.in_proj_weight.data_ptr() + E * E * f4, E,
                                         mha.in_proj_bias.data_ptr() + E * f4, B * instr.shape[1], 2 * E, E, dev)
            st["tensors"].append(st["lang_kv"])
        st["nsplit"] = max(1, min(8, Sp // 128, -(-DN_TARGET_WGS // (B * H))))
        nws = O.L.load().a3d_dn_cross_ws_floats(B, H, st["nsplit"])
        st["ws"] = torch.empty((nws,), device=dev, dtype=torch.float32)
        st["ws_side"] = torch.empty((nws,), device=dev, dtype=torch.float32)      # rotation branch, concurrent
        # with_persist=False: the caller is about to replay a captured graph that owns its persistent-sampler state (tables, exchange
        # buffers, synchronisation words) -- only the refreshed K / V / modulation tensors of this call are needed
        st["persist"] = self._build_persist(st, B, Ln, H, E, Sp, time_sin.shape[0], dev) if (DN_PERSIST and with_persist) else None
        return st

    def _build_persist(self, st, B, Ln, H, E, Sp, T, dev):
        """State of the persistent sampler (a3d_dn_persist: the whole denoise loop as one launch, csrc/denoise.hip): the device
        table of per-layer parameter blocks (AdaLN tables by their base: the kernel indexes them with the step), the query /
        partial exchange buffers and the synchronisation words.  None when the batch leaves too few CUs for the streaming role."""
        import ctypes
        Lb = O.L
        lib = Lb.load()
        NT = -(-Ln // 16)                                   # 16-step row tiles per trajectory: one sample-role workgroup each
        # two sample-role workgroups per (trajectory, tile) -- primary + rotation-stack helper -- and >= 16 streamers, all co-resident
        if 2 * B * NT + 16 > torch.cuda.get_device_properties(dev).multi_processor_count or H > 8 or NT > 4:
            return None
        recs = st["layers"]
        table = (Lb.DnLayerParams * len(recs))()
        for i, rec in enumerate(recs):
            lay, mods, ff = rec["lay"], rec["mods"], rec["lay"].ffn_12
            table[i].cross = Lb.DnCrossParams(
                sem=st["sem"].data_ptr(), mod=mods[0].data_ptr(), q_w=lay.cross_12.in_proj_weight.data_ptr(),
                q_b=lay.cross_12.in_proj_bias.data_ptr(), freq=st["freq"].data_ptr(), Kf=rec["Kf"].data_ptr(), Vt=rec["Vt"].data_ptr())
            table[i].rest = Lb.DnRestParams(
                c_out_w=lay.cross_12.out_proj.weight.data_ptr(), c_out_b=lay.cross_12.out_proj.bias.data_ptr(),
                c_ln_g=lay.norm_12.weight.data_ptr(), c_ln_b=lay.norm_12.bias.data_ptr(), sem=st["sem"].data_ptr(),
                s_mod=mods[1].data_ptr(), s_in_w=lay.sa1.in_proj_weight.data_ptr(), s_in_b=lay.sa1.in_proj_bias.data_ptr(),
                s_out_w=lay.sa1.out_proj.weight.data_ptr(), s_out_b=lay.sa1.out_proj.bias.data_ptr(),
                s_ln_g=lay.norm_1.weight.data_ptr(), s_ln_b=lay.norm_1.bias.data_ptr(), freq=st["freq"].data_ptr(),
                kmask=None if st["kmask"] is None else st["kmask"].data_ptr(), f_mod=mods[2].data_ptr(), f_w1=ff[0].weight.data_ptr(),
                f_b1=ff[0].bias.data_ptr(), f_w2=ff[3].weight.data_ptr(), f_b2=ff[3].bias.data_ptr(),
                f_ln_g=lay.norm_122.weight.data_ptr(), f_ln_b=lay.norm_122.bias.data_ptr(), F=ff[0].weight.shape[0])
        raw = bytes(memoryview(table))
        nsplit = max(1, min(DN_PERSIST_SPLIT, Sp // 64, 16 // lib.a3d_dn_persist_splits(H, 1)))      # at most 16 partials per (sample, head)
        nse = lib.a3d_dn_persist_splits(H, nsplit)
        n_layers = len(recs)
        return {
            "table": torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev),
            "nsplit": nsplit,
            "qbuf": torch.zeros((2 * B * NT * 16 * 128,), device=dev, dtype=torch.float32),
            "part": torch.empty((lib.a3d_dn_cross_ws_floats(2 * B * NT, H, nse),), device=dev, dtype=torch.float32),
            "xbuf": torch.zeros((lib.a3d_dn_persist_xbuf_floats(B, Ln),), device=dev, dtype=torch.float32),
            "kvx": torch.empty((lib.a3d_dn_persist_kvx_floats(B, Ln, E),), device=dev, dtype=torch.float32) if NT > 1 else None,
            "sync": torch.zeros((lib.a3d_dn_persist_sync_ints(B, Ln, n_layers, T),), device=dev, dtype=torch.int32),
            "stacks": (len(self.traj_attention[0].layers), len(self.pos_attention[0].layers), len(self.rot_attention[0].layers)),
        }

    @torch.no_grad()
    def fused_persist(self, st, traj, t_first, nsteps, step_noise, cond_data, cond_mask_u8, tb):
        """nsteps consecutive denoise steps t_first, t_first - 1, ... (network evaluation + DDPM reverse step each) as ONE launch of
        the persistent sampler; returns the trajectory after the last of them (a new tensor).  step_noise: the (T, B, L, D) table."""
        Lb = O.L
        ps = st["persist"]
        B, Ln, D = traj.shape
        H = self.num_attn_heads
        E = self.curr_gripper_embed.weight.shape[1]
        hp = Lb.DnHeadParams(enc_w0=self.traj_encoder[0].weight.data_ptr(), enc_b0=self.traj_encoder[0].bias.data_ptr(),
                             enc_w1=self.traj_encoder[3].weight.data_ptr(), enc_b1=self.traj_encoder[3].bias.data_ptr(),
                             sem=st["sem"].data_ptr(), lang_kv=None if st["lang_kv"] is None else st["lang_kv"].data_ptr(),
                             S_lang=st.get("S_lang", 0))
        if st["lang_kv"] is not None:
            ll = self.traj_lang_attention[0].layers[0]
            hp.q_w, hp.q_b = ll.cross_12.in_proj_weight.data_ptr(), ll.cross_12.in_proj_bias.data_ptr()
            hp.out_w, hp.out_b = ll.cross_12.out_proj.weight.data_ptr(), ll.cross_12.out_proj.bias.data_ptr()
            hp.ln_g, hp.ln_b = ll.norm_12.weight.data_ptr(), ll.norm_12.bias.data_ptr()
        pr, rr = self.pos_regressor[0], self.rot_regressor[0]
        tp = Lb.DnTailParams(pos_w0=pr[0].weight.data_ptr(), pos_b0=pr[0].bias.data_ptr(), pos_w1=pr[3].weight.data_ptr(),
                             pos_b1=pr[3].bias.data_ptr(), rot_w0=rr[0].weight.data_ptr(), rot_b0=rr[0].bias.data_ptr(),
                             rot_w1=rr[3].weight.data_ptr(), rot_b1=rr[3].bias.data_ptr(), noise=step_noise.data_ptr(),
                             cond_data=cond_data.data_ptr(), cond_mask=cond_mask_u8.data_ptr(), coef_pos=tb.coef_pos.data_ptr(),
                             coef_rot=tb.coef_rot.data_ptr())
        out = traj.clone()
        self._last_persist = ps                      # tests read the abort word (sync[2]) after synchronising
        nt, npos, nrot = ps["stacks"]
        Lb.call("a3d_dn_persist", ps["table"].data_ptr(), nt, npos, nrot, C_byref(hp), C_byref(tp), out.data_ptr(), ps["qbuf"].data_ptr(),
                ps["part"].data_ptr(), None if ps["kvx"] is None else ps["kvx"].data_ptr(), ps["xbuf"].data_ptr(), ps["sync"].data_ptr(),
                B, Ln, D, E, H, st["S"], st["Sp"], ps["nsplit"], int(t_first), int(nsteps),
                Lb.stream())
        if DN_PERSIST_CHECK:
            if int(ps["sync"][2].item()) != 0:
                raise RuntimeError("a3d_dn_persist gave up waiting (sync[2] != 0): the trajectory is invalid")
        return out

    @torch.no_grad()
    def fused_step(self, st, traj, t, noise, cond_data, cond_mask_u8, tb):
        """One denoise step: network evaluation at timestep t + DDPM reverse step -> the next trajectory (B, L, D)."""
        Lb = O.L
        out = torch.empty_like(traj)
        B, Ln, D = traj.shape
        H = self.num_attn_heads
        E = self.curr_gripper_embed.weight.shape[1]
        dev = traj.device
        f4 = 4
        nz = lambda x: None if x is None else x.data_ptr()
        new = lambda: torch.empty((B, Ln, E), device=dev, dtype=torch.float32)
        hp = Lb.DnHeadParams(enc_w0=self.traj_encoder[0].weight.data_ptr(), enc_b0=self.traj_encoder[0].bias.data_ptr(),
                             enc_w1=self.traj_encoder[3].weight.data_ptr(), enc_b1=self.traj_encoder[3].bias.data_ptr(),
                             sem=st["sem"].data_ptr(), lang_kv=None if st["lang_kv"] is None else st["lang_kv"].data_ptr(),
                             S_lang=st.get("S_lang", 0))
        if st["lang_kv"] is not None:
            ll = self.traj_lang_attention[0].layers[0]
            hp.q_w, hp.q_b = ll.cross_12.in_proj_weight.data_ptr(), ll.cross_12.in_proj_bias.data_ptr()
            hp.out_w, hp.out_b = ll.cross_12.out_proj.weight.data_ptr(), ll.cross_12.out_proj.bias.data_ptr()
            hp.ln_g, hp.ln_b = ll.norm_12.weight.data_ptr(), ll.norm_12.bias.data_ptr()
        x = new()
        Lb.call("a3d_dn_head", traj.data_ptr(), D, C_byref(hp), x.data_ptr(), B, Ln, E, H, Lb.stream())

        def run_layer(xin, rec, ws):
            lay = rec["lay"]
            stream = Lb.stream()
            mo = [m.data_ptr() + t * 2 * E * f4 for m in rec["mods"]]
            cp = Lb.DnCrossParams(sem=st["sem"].data_ptr(), mod=mo[0], q_w=lay.cross_12.in_proj_weight.data_ptr(),
                                  q_b=lay.cross_12.in_proj_bias.data_ptr(), freq=st["freq"].data_ptr(), Kf=rec["Kf"].data_ptr(),
                                  Vt=rec["Vt"].data_ptr())
            Lb.call("a3d_dn_cross", xin.data_ptr(), traj.data_ptr(), D, C_byref(cp), ws.data_ptr(), B, Ln, E, H, st["S"],
                    st["Sp"], st["nsplit"], stream)
            ff = lay.ffn_12
            rp = Lb.DnRestParams(
                c_out_w=lay.cross_12.out_proj.weight.data_ptr(), c_out_b=lay.cross_12.out_proj.bias.data_ptr(),
                c_ln_g=lay.norm_12.weight.data_ptr(), c_ln_b=lay.norm_12.bias.data_ptr(), sem=st["sem"].data_ptr(),
                s_mod=mo[1], s_in_w=lay.sa1.in_proj_weight.data_ptr(), s_in_b=lay.sa1.in_proj_bias.data_ptr(),
                s_out_w=lay.sa1.out_proj.weight.data_ptr(), s_out_b=lay.sa1.out_proj.bias.data_ptr(),
                s_ln_g=lay.norm_1.weight.data_ptr(), s_ln_b=lay.norm_1.bias.data_ptr(), freq=st["freq"].data_ptr(),
                kmask=None if st["kmask"] is None else st["kmask"].data_ptr(), f_mod=mo[2], f_w1=ff[0].weight.data_ptr(), f_b1=ff[0].bias.data_ptr(),
                f_w2=ff[3].weight.data_ptr(), f_b2=ff[3].bias.data_ptr(), f_ln_g=lay.norm_122.weight.data_ptr(),
                f_ln_b=lay.norm_122.bias.data_ptr(), F=ff[0].weight.shape[0])
            xout = new()
            Lb.call("a3d_dn_rest", xin.data_ptr(), traj.data_ptr(), D, ws.data_ptr(), C_byref(rp), xout.data_ptr(), B, Ln,
                    E, H, st["nsplit"], stream)
            return xout

        recs = st["layers"]
        n_traj = len(self.traj_attention[0].layers)
        n_pos = len(self.pos_attention[0].layers)
        for rec in recs[:n_traj]:
            x = run_layer(x, rec, st["ws"])
        # the position and rotation stacks (diffusion_head.py:343-357) both start from x and are independent: the rotation
        # stack runs on a side stream (fork / join on events: capturable), its 64-workgroup per-sample kernels filling CUs
        # the position stack leaves idle
        cur = torch.cuda.current_stream(dev)
        side = _dn_side_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            rf = x
            for rec in recs[n_traj + n_pos:]:
                rf = run_layer(rf, rec, st["ws_side"])
        pf = x
        for rec in recs[n_traj:n_traj + n_pos]:
            pf = run_layer(pf, rec, st["ws"])
        cur.wait_stream(side)
        rf.record_stream(cur)
        stream = Lb.stream()
        pr, rr = self.pos_regressor[0], self.rot_regressor[0]
        tp = Lb.DnTailParams(pos_w0=pr[0].weight.data_ptr(), pos_b0=pr[0].bias.data_ptr(), pos_w1=pr[3].weight.data_ptr(),
                             pos_b1=pr[3].bias.data_ptr(), rot_w0=rr[0].weight.data_ptr(), rot_b0=rr[0].bias.data_ptr(),
                             rot_w1=rr[3].weight.data_ptr(), rot_b1=rr[3].bias.data_ptr(), noise=nz(noise),
                             cond_data=cond_data.data_ptr(), cond_mask=cond_mask_u8.data_ptr(), coef_pos=tb.coef_pos.data_ptr(),
                             coef_rot=tb.coef_rot.data_ptr())
        Lb.call("a3d_dn_tail", pf.data_ptr(), rf.data_ptr(), traj.data_ptr(), D, C_byref(tp), out.data_ptr(), B, Ln, E,
                int(t), stream)
        return out


# workgroups the cached cross-attention kernel aims for (key splits x samples x heads).  Every split repeats the query
# projection and adds a combine term, so splits only pay while the grid is smaller than the chip: measured at cfg-3
# (B x H = 512) 1 split 0.924 ms per denoise step, 2 splits 0.955, 4 splits 1.04, 8 splits 1.20
DN_TARGET_WGS = int(os.environ.get("A3D_DN_TARGET_WGS", "512"))
FUSED_DENOISE = os.environ.get("A3D_DN_FUSED", "1") == "1"
# the sampling loop as ONE launch of the persistent two-role kernel (a3d_dn_persist); A3D_DN_PERSIST=0: one launch per phase
# (head, per layer cross + rest, tail: 18 per step).  A3D_DN_PERSIST_SPLIT: key splits per (sample, layer) = items of the ready queue.
DN_PERSIST = os.environ.get("A3D_DN_PERSIST", "1") == "1"
# (round 6: 6 -- 0.730 ms per denoise step at cfg-3 against 0.755 with 8 and 0.749 with 4, profiles/r06_sampler_split.json: fewer,
# longer items amortise the ~9 us of per-item overhead until the sample role starts to wait for its last item)
DN_PERSIST_SPLIT = int(os.environ.get("A3D_DN_PERSIST_SPLIT", "6"))
# An aborted persistent launch (a co-resident workgroup never arrived) poisons the whole trajectory batch with NaN inside the launch
# sequence itself (dn_persist_poison_kernel), so the failure is visible in the result without a host synchronisation, eager or
# replayed.  A3D_DN_PERSIST_CHECK=1 additionally synchronises after every launch and raises on the abort word.
DN_PERSIST_CHECK = os.environ.get("A3D_DN_PERSIST_CHECK", "0") == "1"
_DN_SIDE = {}


def _dn_side_stream(dev):
    key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    if key not in _DN_SIDE:
        _DN_SIDE[key] = torch.cuda.Stream(device=dev)
    return _DN_SIDE[key]


def C_byref(struct):
    import ctypes
    return ctypes.byref(struct)


class DiffusionPlanner(nn.Module):

    def __init__(self, backbone="clip", image_size=(256, 256), embedding_dim=60, output_dim=7,
                 num_vis_ins_attn_layers=2, num_query_cross_attn_layers=8, use_instruction=False, use_goal=False,
                 use_goal_at_test=True, feat_scales_to_use=1, attn_rounds=1, weight_tying=False,
                 gripper_loc_bounds=None, rotation_parametrization='quat', diffusion_timesteps=100, num_attn_heads=8,
                 dropout=0.1, dropout_seed=0):
        super().__init__()
        if rotation_parametrization != '6D':
            raise NotImplementedError("only rotation_parametrization='6D' (scripts/train_trajectory.sh) is implemented")
        self._use_goal, self._use_goal_at_test = use_goal, use_goal_at_test
        self._rotation_parametrization = rotation_parametrization
        self.prediction_head = DiffusionHead(
            backbone=backbone, image_size=image_size, embedding_dim=embedding_dim, output_dim=output_dim,
            num_attn_heads=num_attn_heads, num_vis_ins_attn_layers=num_vis_ins_attn_layers,
            num_query_cross_attn_layers=num_query_cross_attn_layers, use_instruction=use_instruction, use_goal=use_goal,
            feat_scales_to_use=feat_scales_to_use, attn_rounds=attn_rounds, weight_tying=weight_tying,
            rotation_parametrization=rotation_parametrization, dropout=dropout, dropout_seed=dropout_seed)
        self.n_steps = diffusion_timesteps
        self.register_buffer("gripper_loc_bounds", torch.tensor(gripper_loc_bounds, dtype=torch.float32), persistent=False)
        self._tables = None
        self._graph = None

    # ---- helpers (diffusion_model.py:187-230)
    def tables(self, device):
        if self._tables is None or self._tables.acp_pos.device != device:
            self._tables = DDPMTables(self.n_steps, device)
            E = self.prediction_head.curr_gripper_embed.weight.shape[1]
            sin = O.sinusoidal_emb(torch.arange(self.n_steps, device=device, dtype=torch.float32), E)
            self._time_tables = {"sin": sin, "silu": F.silu(sin)}
        return self._tables

    def normalize_pos(self, pos):
        lo, hi = self.gripper_loc_bounds[0], self.gripper_loc_bounds[1]
        return (pos - lo) / (hi - lo) * 2.0 - 1.0

    def unnormalize_pos(self, pos):
        lo, hi = self.gripper_loc_bounds[0], self.gripper_loc_bounds[1]
        return (pos + 1.0) / 2.0 * (hi - lo) + lo

    def convert_rot(self, signal):
        return pose_to_signal(signal)

    def unconvert_rot(self, signal):
        return signal_to_pose(signal)

    def _prepare(self, rgb_obs, pcd_obs, curr_gripper, goal_gripper, visual_tokens):
        """Normalised, converted conditioning + visual tokens and their (normalised, down-sampled) coordinates
        (one tensor each, or one per scale for a multi-scale head)."""
        head = self.prediction_head
        with torch.no_grad():
            pcd_n = self.normalize_pos(pcd_obs.float().permute(0, 1, 3, 4, 2)).permute(0, 1, 4, 2, 3).contiguous()
            by_factor = {}
            for f in head.downscaling_factor_pyramid[:head.feat_scales]:
                if f not in by_factor:
                    by_factor[f] = O.pcd_downsample(pcd_n, f)
            ctx_xyz = [by_factor[f] for f in head.downscaling_factor_pyramid[:head.feat_scales]]
            cg = pose_to_signal(curr_gripper, self.gripper_loc_bounds)
            gg = pose_to_signal(goal_gripper, self.gripper_loc_bounds)
        tokens = visual_tokens if visual_tokens is not None else head.encode_images(rgb_obs, pcd_n)
        if head.feat_scales == 1:
            ctx_xyz = ctx_xyz[0]
        return tokens, ctx_xyz, cg, gg

    # ---- training (diffusion_model.py:253-324)
    def forward(self, gt_trajectory, trajectory_mask, rgb_obs, pcd_obs, instruction, curr_gripper, goal_gripper,
                run_inference=False, *, noise=None, timesteps=None, visual_tokens=None, return_pred=False, **sample_kw):
        if run_inference:
            return self.compute_trajectory(trajectory_mask, rgb_obs, pcd_obs, instruction, curr_gripper, goal_gripper,
                                           visual_tokens=visual_tokens, **sample_kw)
        head = self.prediction_head
        dev = pcd_obs.device
        tb = self.tables(dev)
        tokens, ctx_xyz, cg, gg = self._prepare(rgb_obs, pcd_obs, curr_gripper, goal_gripper, visual_tokens)
        with torch.no_grad():
            gt = pose_to_signal(gt_trajectory, self.gripper_loc_bounds)
            if noise is None:
                noise = torch.randn(gt.shape, device=dev)
            if timesteps is None:
                timesteps = torch.randint(0, self.n_steps, (gt.shape[0],), device=dev).long()
            gt = gt[..., :9].contiguous()               # the reference rebuilds the 9 pose channels (diffusion_model.py:296-305)
            noisy = O.ddpm_add_noise(gt, noise.to(dev).float()[..., :9].contiguous(), timesteps.to(dev), tb.acp_pos, tb.acp_rot)
        if head.attn_rounds * head.feat_scales > 1:
            # every iteration's prediction is supervised (diffusion_model.py:313-323)
            toks = tokens if isinstance(tokens, (list, tuple)) else [tokens]
            xyzs = ctx_xyz if isinstance(ctx_xyz, (list, tuple)) else [ctx_xyz]
            preds = head.forward_multi(noisy, trajectory_mask, timesteps.to(dev), toks, xyzs, instruction, cg, gg,
                                       head.begin_dropout if (head.training and head.dropout_p > 0) else None)
            loss = sum(O.ElemLossFn.apply(p_[..., :3], gt[..., :3], 1, 100.0) + O.ElemLossFn.apply(p_[..., 3:9], gt[..., 3:9], 1, 10.0)
                       for p_ in preds)
            return (loss, preds, gt) if return_pred else loss
        drop = head.begin_dropout()
        ctx, ctx_xyz, instr = head.encode_context(tokens, ctx_xyz, instruction, cg, gg, drop=drop)
        pred = head.forward_tokens(noisy, trajectory_mask, timesteps.to(dev), ctx, ctx_xyz, instr, drop=drop)
        loss = O.ElemLossFn.apply(pred[..., :3], gt[..., :3], 1, 100.0) + O.ElemLossFn.apply(pred[..., 3:9], gt[..., 3:9], 1, 10.0)
        return (loss, pred, gt) if return_pred else loss

    # ---- sampling (diffusion_model.py:86-185)
    @torch.no_grad()
    def compute_trajectory(self, trajectory_mask, rgb_obs, pcd_obs, instruction, curr_gripper, goal_gripper, *,
                           init_noise=None, step_noise=None, visual_tokens=None, use_graph=False, n_steps=None,
                           return_trace=False, fused=None):
        head = self.prediction_head
        dev = pcd_obs.device
        tb = self.tables(dev)
        B, Ln = trajectory_mask.shape
        tokens, ctx_xyz, cg, gg = self._prepare(rgb_obs, pcd_obs, curr_gripper, goal_gripper, visual_tokens)
        multi = head.attn_rounds * head.feat_scales > 1
        if not multi:
            ctx, ctx_xyz, instr = head.encode_context(tokens, ctx_xyz, instruction, cg, gg)
        # conditioning: start pose at index 0, goal at L - pad - 1 and after (no host sync: index arithmetic on device)
        D = cg.shape[-1]
        E = head.curr_gripper_embed.weight.shape[1]
        ar = torch.arange(Ln, device=dev)[None, :]
        cond_mask = (ar == 0)
        cond_data = torch.zeros((B, Ln, D), device=dev)
        cond_data[:, 0] = cg
        if self._use_goal_at_test:
            gidx = (Ln - trajectory_mask.sum(1).long() - 1)[:, None]
            cond_mask = cond_mask | (ar >= gidx)
            cond_data = torch.where((ar == gidx)[..., None], gg[:, None, :], cond_data)
        cond_mask_u8 = cond_mask[..., None].expand(B, Ln, D).to(torch.uint8).contiguous()
        cond_data = cond_data.contiguous()
        if init_noise is None:
            init_noise = torch.randn((B, Ln, D), device=dev)
        if step_noise is None:
            step_noise = torch.randn((self.n_steps, B, Ln, D), device=dev)
        step_noise = step_noise.to(dev).float().contiguous()
        steps = list(range(self.n_steps - 1, -1, -1))
        if n_steps is not None:
            steps = steps[:n_steps]
        traj = (init_noise.to(dev).float() + cond_data).contiguous()
        kmask = trajectory_mask.to(torch.uint8).contiguous()
        trace = []
        # fused per-step kernels (csrc/denoise.hip) whenever the trajectory fits one 16-row tile; else the op-by-op path
        fused = FUSED_DENOISE if fused is None else fused
        # fused kernels (csrc/denoise.hip): per-phase launches serve one 16-row tile; the persistent sampler up to four (L <= 64: the
        # reference's interpolation_length = 50, scripts/train_trajectory.sh:7-8, online_evaluation/eval.sh:17)
        fused = fused and E <= 128 and D <= 16 and min(Ln, 16) * D <= 160 and not multi and (Ln <= 16 or (DN_PERSIST and Ln <= 64))
        if multi:
            # multi-round / multi-scale heads: the fine-scale context follows the previous prediction, so nothing but the
            # image encoding is step-invariant -- every step evaluates the full head (no K/V cache, no fused kernels)
            toks = tokens if isinstance(tokens, (list, tuple)) else [tokens]
            xyzs = ctx_xyz if isinstance(ctx_xyz, (list, tuple)) else [ctx_xyz]
            state, static = None, list(toks) + list(xyzs) + [instruction, cg, gg]
            tmask = trajectory_mask.bool()
        elif fused:
            # a replay of the captured loop addresses the state retained in self._graph: skip building a second set of persistent-
            # sampler buffers (a ctypes table, a pageable host-to-device copy = a host sync, five allocations) that would be thrown away
            gr_ = self._graph
            reuse = (use_graph and not return_trace and gr_ is not None and gr_["key"][:3] == (B, Ln, tuple(steps)) and
                     isinstance(gr_.get("state"), dict) and gr_["state"].get("persist") is not None)
            state = head.build_fused(ctx, ctx_xyz, instr, kmask, self._time_tables["sin"], Ln, with_persist=not reuse)
            if reuse:
                state["persist"] = gr_["state"]["persist"]
            static = list(state["tensors"])
            if Ln > 16 and state.get("persist") is None:           # too many units for the CU count: the op-by-op path serves it
                fused = False
                state = head.build_kv_cache(ctx, ctx_xyz, instr)
                static = [c[k] for c in state["ctx"] for k in ("Ks", "Vt")] + \
                    ([state["lang"]["Ks"], state["lang"]["Vt"]] if "lang" in state else [])
        else:
            state = head.build_kv_cache(ctx, ctx_xyz, instr)
            static = [c[k] for c in state["ctx"] for k in ("Ks", "Vt")] + \
                ([state["lang"]["Ks"], state["lang"]["Vt"]] if "lang" in state else [])
        static = static + [step_noise, cond_data, cond_mask_u8, kmask]

        persist = fused and state.get("persist") is not None and all(a_ - b_ == 1 for a_, b_ in zip(steps, steps[1:]))
        # which sampler serves this call (read by the bench line and the tests)
        self.last_sampler_path = "multi-round" if multi else ("persistent (a3d_dn_persist)" if persist else
                                                               ("per-phase fused launches" if fused else "op-by-op"))

        def run_loop(x):
            if persist and not return_trace:                # the whole loop: one launch
                return head.fused_persist(state, x, steps[0], len(steps), step_noise, cond_data, cond_mask_u8, tb)
            for t in steps:
                nz = step_noise[t] if t > 0 else None
                if multi:
                    tt = torch.full((B,), t, device=dev, dtype=torch.long)
                    out = head.forward_multi(x, tmask, tt, toks, xyzs, instruction, cg, gg)[-1]
                    x = O.ddpm_step(out, x, nz, cond_data, cond_mask_u8, tb.coef_pos, tb.coef_rot, t)
                elif persist:                              # traced: the same kernel, one step per launch
                    x = head.fused_persist(state, x, t, 1, step_noise, cond_data, cond_mask_u8, tb)
                elif fused:
                    x = head.fused_step(state, x, t, nz, cond_data, cond_mask_u8, tb)
                else:
                    out = head.denoise_tokens_cached(x, kmask, t, state, self._time_tables)
                    x = O.ddpm_step(out, x, nz, cond_data, cond_mask_u8, tb.coef_pos, tb.coef_rot, t)
                if return_trace:
                    trace.append(x)
            return x

        if use_graph and not return_trace:
            key = (B, Ln, tuple(steps), fused, tuple((tuple(t_.shape), t_.dtype) for t_ in static))
            if self._graph is not None and self._graph["key"] != key and fused and state.get("persist") is self._graph["state"].get("persist"):
                # the shapes changed after all (another context size): this call needs its own persistent-sampler state
                state["persist"] = head._build_persist(state, B, Ln, head.num_attn_heads, E, state["Sp"], self.n_steps, dev) if DN_PERSIST else None
                persist = fused and state.get("persist") is not None and all(a_ - b_ == 1 for a_, b_ in zip(steps, steps[1:]))
            if self._graph is None or self._graph["key"] != key:
                static_in = traj.clone()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    run_loop(static_in)                     # warm-up (allocator, lazy module state) outside capture
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                # "state" keeps every buffer the captured launches address alive (workspaces, the persistent sampler's tables)
                self._graph = {"key": key, "in": static_in, "static": static, "state": state}
                with torch.cuda.graph(g):
                    self._graph["out"] = run_loop(static_in)
                self._graph["g"] = g
            gr = self._graph
            # refresh the captured buffers in place (same addresses; the key pins every shape and dtype)
            gr["in"].copy_(traj)
            for dst, src in zip(gr["static"], static):
                if dst is not src:
                    dst.copy_(src)
            gr["g"].replay()
            traj = gr["out"]
        else:
            traj = run_loop(traj)
        final = signal_to_pose(traj, self.gripper_loc_bounds)
        return (final, trace) if return_trace else final
