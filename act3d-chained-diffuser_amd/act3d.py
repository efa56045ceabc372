"""Act3D keypose policy on the MI355X hot path.

Drop-in for `model.Act3D` of the reference (model/keypose_optimization/act3d.py:20-535): same constructor keywords,
same `forward(visible_rgb, visible_pcd, instruction, curr_gripper, gt_action=None)` and the same output dictionary,
same parameter names/shapes (checkpoints interchange).  Everything after the frozen backbone + FPN runs as HIP
kernels (ops.py); tokens are batch-first internally and RoPE codes are never materialised.

Additive (non-breaking) keyword arguments for testability: `ghost_points` (inject the sampled points),
`teacher_positions` (teacher-force the per-level prediction), `visual_features` (bypass backbone + FPN).
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops as O
from .nn import FeaturePyramidNetwork, RelativeCrossAttentionModule, load_synthetic_clip, run_frozen_backbone


class _BroadcastRowFn(torch.autograd.Function):
    """weight (1, E) -> (B, N, E).  Backward column-sums straight into weight.grad with the wgrad kernel
    (replaces `embed.weight.unsqueeze(0).repeat(...)`, act3d.py:220,282,455)."""

    @staticmethod
    def forward(ctx, weight, B, N):
        ctx.weight = weight
        # a real copy even for B * N == 1 (expand().contiguous() would alias the parameter, whose storage inside the flat
        # parameter buffer is only 4-byte aligned and is updated in place by the optimizer)
        return weight.detach().view(1, 1, -1).expand(B, N, -1).clone(memory_format=torch.contiguous_format)

    @staticmethod
    def backward(ctx, dy):
        w = ctx.weight
        dy2 = O._c(dy).view(-1, dy.shape[-1])
        ones = torch.ones((dy2.shape[0], 1), device=dy.device, dtype=torch.float32)
        gW = O.grad_buf(w)                       # (1, E) viewed as a (E, 1) weight-gradient: dW[e][0] += sum_m dy[m][e]
        O.L.call("a3d_linear_wgrad", dy2.data_ptr(), dy2.shape[1], ones.data_ptr(), 1, gW.data_ptr(), 1, None,
                 dy2.shape[0], dy2.shape[1], 1, O.L.stream())
        return None, None, None


def broadcast_row(weight, B, N):
    return _BroadcastRowFn.apply(weight, B, N)


# Opt-in: measured on MI355X (B = 16) the fork/join helps the eager step (+33 % hot-path throughput) but costs 4 % under
# hipGraph replay, where the step already has no launch gaps and the extra branch joins are not free.
OVERLAP_STREAMS = os.environ.get("A3D_OVERLAP_STREAMS", "0") == "1"
_SIDE_STREAMS = {}


def _side_stream(device):
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


class Act3D(nn.Module):

    def __init__(self,
                 backbone="clip",
                 image_size=(256, 256),
                 embedding_dim=60,
                 num_attn_heads=4,
                 num_ghost_point_cross_attn_layers=2,
                 num_query_cross_attn_layers=2,
                 num_vis_ins_attn_layers=2,
                 rotation_parametrization="quat_from_query",
                 gripper_loc_bounds=None,
                 num_ghost_points=1000,
                 num_ghost_points_val=10000,
                 weight_tying=True,
                 gp_emb_tying=True,
                 ins_pos_emb=False,
                 num_sampling_level=3,
                 fine_sampling_ball_diameter=0.16,
                 regress_position_offset=False,
                 use_instruction=False,
                 ghost_sampler="philox",
                 sampler_seed=0):
        super().__init__()
        assert backbone in ["resnet", "clip"]
        image_size = tuple(image_size)
        assert image_size in [(128, 128), (256, 256)]
        assert rotation_parametrization in ["quat_from_top_ghost", "quat_from_query", "6D_from_top_ghost", "6D_from_query"]
        assert num_sampling_level in [1, 2, 3, 4]
        assert ghost_sampler in ("philox", "numpy")

        self.image_size = image_size
        self.rotation_parametrization = rotation_parametrization
        self.num_ghost_points = num_ghost_points // num_sampling_level
        self.num_ghost_points_val = num_ghost_points_val // num_sampling_level
        self.num_sampling_level = num_sampling_level
        self.sampling_ball_diameter_pyramid = [None, fine_sampling_ball_diameter, fine_sampling_ball_diameter / 4.0,
                                               fine_sampling_ball_diameter / 16.0]
        self.gripper_loc_bounds = np.array(gripper_loc_bounds)
        self.regress_position_offset = regress_position_offset
        self.weight_tying, self.gp_emb_tying, self.ins_pos_emb = weight_tying, gp_emb_tying, ins_pos_emb
        self.num_attn_heads = num_attn_heads
        self.ghost_sampler = ghost_sampler

        # Frozen backbone (synthetic CLIP-RN50-shaped: the real weights are not available offline)
        self.backbone, self.normalize = load_synthetic_clip()
        for p in self.backbone.parameters():
            p.requires_grad = False
        self.backbone_dtype = torch.float32      # set to torch.bfloat16 to run the frozen backbone under autocast
        self.fpn_dtype = torch.float32           # set to torch.bfloat16 to run the FPN convolutions under autocast

        self.feature_pyramid = FeaturePyramidNetwork([64, 256, 512, 1024, 2048], embedding_dim)
        if self.image_size == (128, 128):
            # the reference sets `coarse_feature_map` here but reads `feature_map_pyramid` (act3d.py:81 vs :378) and
            # crashes; the evident intent is implemented
            self.feature_map_pyramid = ['res2', 'res1', 'res1', 'res1']
            self.downscaling_factor_pyramid = [4, 2, 2, 2]
        else:
            self.feature_map_pyramid = ['res3', 'res1', 'res1', 'res1']
            self.downscaling_factor_pyramid = [8, 2, 2, 2]

        def tied(make, tie):
            if tie:
                m = make()
                return nn.ModuleList([m for _ in range(num_sampling_level)])
            return nn.ModuleList([make() for _ in range(num_sampling_level)])

        self.ghost_points_embed_pyramid = tied(lambda: nn.Embedding(1, embedding_dim), gp_emb_tying)
        self.curr_gripper_embed = nn.Embedding(1, embedding_dim)
        self.query_embed = nn.Embedding(1, embedding_dim)
        self.ghost_point_cross_attn_pyramid = tied(lambda: RelativeCrossAttentionModule(
            embedding_dim, num_attn_heads, num_ghost_point_cross_attn_layers), weight_tying)
        self.use_instruction = use_instruction
        if use_instruction:
            self.vis_ins_attn_pyramid = tied(lambda: RelativeCrossAttentionModule(
                embedding_dim, num_attn_heads, num_vis_ins_attn_layers), weight_tying)
        self.query_cross_attn_pyramid = tied(lambda: RelativeCrossAttentionModule(
            embedding_dim, num_attn_heads, num_query_cross_attn_layers), weight_tying)
        if regress_position_offset:
            self.ghost_point_offset_predictor = nn.Sequential(nn.Linear(embedding_dim, embedding_dim), nn.ReLU(),
                                                              nn.Linear(embedding_dim, 3))
        self.rotation_dim = 4 if "quat" in rotation_parametrization else 6
        self.gripper_state_predictor = nn.Sequential(nn.Linear(embedding_dim, embedding_dim), nn.ReLU(),
                                                     nn.Linear(embedding_dim, self.rotation_dim + 1))
        if use_instruction:
            self.instruction_encoder = nn.Linear(512, embedding_dim)
            if ins_pos_emb:
                self._num_words = 53
                self.instr_position_embedding = nn.Embedding(self._num_words, embedding_dim)
                self.instr_position_norm = nn.LayerNorm(embedding_dim)

        # device-side sampler state {seed, call offset}; advanced by a kernel so that captured graphs draw fresh points
        self.register_buffer("_rng_state", torch.tensor([sampler_seed, 0], dtype=torch.int64), persistent=False)
        self.register_buffer("_bounds", torch.tensor(self.gripper_loc_bounds, dtype=torch.float32), persistent=False)

    # ------------------------------------------------------------------------------------------------ vision (adjacent)
    def _needed_maps(self):
        return sorted(set(self.feature_map_pyramid[:self.num_sampling_level]))

    def backbone_maps(self, visible_rgb, out=None):
        """The frozen half of compute_visual_tokens: normalize -> backbone under no_grad (act3d.py:363-366), {res1..res5} of the
        B * ncam views.  No gradient flows through it and its weights never change, so a training loop may compute the maps of
        batch k + 1 while step k runs (engine.GraphedStep(prefetch=...)) and hand them to compute_visual_tokens(maps=...).
        out: preallocated maps to write into (same shapes / dtypes / memory format as a previous call returned)."""
        x = visible_rgb.flatten(0, 1)
        with torch.no_grad():
            feats = run_frozen_backbone(self.backbone, x, self.backbone_dtype, keep_dtype=self.fpn_dtype != torch.float32,
                                        normalize=self.normalize, out=out)
        return feats

    def compute_visual_tokens(self, visible_rgb, maps=None):
        """normalize -> frozen backbone -> FPN (act3d.py:363-369), returned token-major per level: (B, ncam*h*w, E).
        Convolutions run channels-last so the (cam, h, w, E) token rows are contiguous and need no transpose.
        maps: the backbone's maps of these views when the caller already has them (backbone_maps)."""
        B, ncam = visible_rgb.shape[:2]
        x = visible_rgb.flatten(0, 1)
        feats = maps if maps is not None else self.backbone_maps(visible_rgb)
        out_bias, out_ctx = {}, {}
        if self.fpn_dtype != torch.float32:
            with torch.autocast("cuda", dtype=self.fpn_dtype):
                # channel count padded to a multiple of 64 for MIOpen; the hot path reads the first E channels of each row.
                # The 3x3 output convolutions run bias-free: their bias travels with the tokens (ops.TokenMap.row_bias) and is added
                # to the rows a level gathers
                E_ = self.curr_gripper_embed.weight.shape[1]
                res = self.feature_pyramid(feats, needed=self._needed_maps(), pad_to=(E_ + 63) // 64 * 64,
                                           defer_output_bias=bool(x.is_cuda), sparse_ncam=ncam if x.is_cuda else None)
                pyr, out_bias = res[0], res[1]
                out_ctx = res[2] if len(res) > 2 else {}
        else:
            pyr = self.feature_pyramid(feats, needed=self._needed_maps())
        tokens = {}
        for name, fm in pyr.items():
            n, E, h, w = fm.shape                    # E: the map's channel count incl. padding (bf16 path)
            # (cam, h, w, E) rows of the channels-last map: a view, in the FPN's own dtype -- a bf16 map is gathered in place
            # by a3d_build_context_bf16 (no fp32 copy of the 128 x 128 map, of which a level reads 6 % of the rows)
            tokens[name] = O.TokenMap(fm.permute(0, 2, 3, 1).reshape(B, ncam * h * w, E), out_bias.get(name), out_ctx.get(name))
        return [tokens[self.feature_map_pyramid[i]] for i in range(self.num_sampling_level)]

    # ------------------------------------------------------------------------------------------------ ghost points
    def _sample_ghost_points(self, B, device, level, anchor):
        """act3d.py:394-440.  "philox": device sampler (no host sync, bounded rejection); "numpy": the reference's
        host sampler, consuming the global numpy RNG identically (one device->host copy of the anchor per level)."""
        Ng = self.num_ghost_points if self.training else self.num_ghost_points_val
        if self.ghost_sampler == "numpy":
            lo, hi = self.gripper_loc_bounds[0], self.gripper_loc_bounds[1]

            def cube(b):
                return np.stack([np.random.uniform(b[0][0], b[1][0], Ng), np.random.uniform(b[0][1], b[1][1], Ng),
                                 np.random.uniform(b[0][2], b[1][2], Ng)], axis=1)
            if level == 0:
                pts = np.stack([cube(self.gripper_loc_bounds) for _ in range(B)])
            else:
                a = anchor.detach().cpu().numpy()
                r = self.sampling_ball_diameter_pyramid[level] / 2
                out = []
                for i in range(B):
                    bb = np.stack([np.clip(a[i] - r, lo, hi), np.clip(a[i] + r, lo, hi)])
                    if np.linalg.norm(np.clip(a[i], bb[0], bb[1]) - a[i]) >= r:
                        raise RuntimeError("ghost-point anchor lies outside the workspace by more than the sampling radius: "
                                           "the reference's rejection sampler would never terminate (SURVEY §0)")
                    acc = np.empty((0, 3))
                    while acc.shape[0] < Ng:
                        c = cube(bb)
                        acc = np.concatenate([acc, c[np.linalg.norm(c - a[i], axis=1) < r]])
                    out.append(acc[:Ng])
                pts = np.stack(out)
            return torch.from_numpy(pts).float().to(device)
        radius = 0.0 if level == 0 else self.sampling_ball_diameter_pyramid[level] / 2
        pts = O.sample_ghost_points(self._rng_state, self._bounds, None if level == 0 else anchor, radius, B, Ng, level)
        O.L.call("a3d_rng_advance", self._rng_state.data_ptr(), 1, O.L.stream())
        return pts

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, visible_rgb, visible_pcd, instruction, curr_gripper, gt_action=None, *, ghost_points=None,
                teacher_positions=None, visual_features=None):
        """
        Arguments (as the reference):
            visible_rgb: (B, ncam, 3, H, W) in [0, 1];  visible_pcd: (B, ncam, 3, H, W) world coordinates
            instruction: (B, 53, 512);  curr_gripper: (B, 8);  gt_action: (B, 8) or None
        """
        B, ncam, _, height, width = visible_pcd.shape
        device = visible_pcd.device
        O.L.require_gpu(visible_pcd, curr_gripper)
        E, H, L = self.curr_gripper_embed.weight.shape[1], self.num_attn_heads, self.num_sampling_level
        gt_position = gt_action[:, :3].detach().float() if gt_action is not None else None
        grip_xyz = curr_gripper[:, :3].float().contiguous()

        # plain tensors (complete feature rows) or ops.TokenMap pairs (tokens + the FPN output bias owed to gathered rows)
        feats = [O.TokenMap.of(f) for f in (visual_features if visual_features is not None
                                            else self.compute_visual_tokens(visible_rgb))]
        pcd_by_factor = {}
        pcd_pyramid = []
        for i in range(L):
            f = self.downscaling_factor_pyramid[i]
            if f not in pcd_by_factor:                      # levels >= 1 share one down-sampled cloud
                pcd_by_factor[f] = O.pcd_downsample(visible_pcd, f)
            pcd_pyramid.append(pcd_by_factor[f])

        instr = None
        if self.use_instruction:
            instr = O.linear(instruction.float(), self.instruction_encoder)
            if self.ins_pos_emb:                        # act3d.py:201-209: + LayerNorm(position embedding), all 53 words
                pe = O.LayerNormFn.apply(self.instr_position_embedding.weight, self.instr_position_norm.weight,
                                         self.instr_position_norm.bias)
                instr = O.AddRowsFn.apply(instr, pe)
            instr_xyz = torch.zeros((B, instr.shape[1], 3), device=device, dtype=torch.float32)

        grip_tok = broadcast_row(self.curr_gripper_embed.weight, B, 1)
        O.begin_grad_sinks()
        accum = {}                                       # one gradient buffer per distinct token map (levels >= 1 share one)
        for f_ in feats:
            accum.setdefault(id(f_.tokens), O.GradAccum())
        position_pyramid, ghost_pcd_pyramid, ghost_pcd_masks_pyramid, topk_pyramid = [], [], [], []
        ghost_features_pyramid = []
        query, prev_pos = None, None
        for i in range(L):
            if ghost_points is not None:
                ghost = ghost_points[i].to(device=device, dtype=torch.float32).contiguous()
            else:
                anchor = None if i == 0 else (gt_position if gt_position is not None else prev_pos)
                ghost = self._sample_ghost_points(B, device, i, anchor)
            # ---- context tokens: all coarse tokens, or the k nearest fine tokens to the previous prediction
            if i == 0:
                idx = None
            else:
                idx = O.knn_topk(prev_pos, pcd_pyramid[i], 32 * 32 * ncam)
            ctx = O.BuildContextFn.apply(feats[i].tokens, idx, grip_tok, accum[id(feats[i].tokens)], feats[i].row_bias,
                                         feats[i].conv_ctx)
            ctx_xyz = O.gather_rows(pcd_pyramid[i], idx, grip_xyz[:, None])
            topk_pyramid.append(idx)
            if self.use_instruction:
                ctx = self.vis_ins_attn_pyramid[i](ctx, instr)[-1]
                ctx = O.BuildContextFn.apply(ctx, None, instr)
                ctx_xyz = torch.cat([ctx_xyz, instr_xyz], dim=1)
            ctx = O.attach_grad_sink(ctx)            # the level's four consumers of ctx sum its gradient in ONE buffer
            # ---- query cross-attends to the context (no positions at level 0).  The query stream (one row per sample:
            # ~100 launch-latency-bound kernels per step, forward + backward) is independent of the ghost stream, so it is
            # issued on a side stream (fork / join on events -- capturable, and autograd replays the same streams backward)
            def query_path(query):
                if i == 0:
                    query = broadcast_row(self.query_embed.weight, B, 1)
                    return self.query_cross_attn_pyramid[i](query, ctx)
                return self.query_cross_attn_pyramid[i](query, ctx, prev_pos[:, None].contiguous(), ctx_xyz)

            overlap = OVERLAP_STREAMS and ctx.is_cuda
            if overlap:
                cur = torch.cuda.current_stream(device)
                side = _side_stream(device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    qlist = query_path(query)
            # ---- ghost points cross-attend to the context
            g0 = broadcast_row(self.ghost_points_embed_pyramid[i].weight, B, ghost.shape[1])
            gfeat = self.ghost_point_cross_attn_pyramid[i](g0, ctx, ghost, ctx_xyz)[-1]
            if overlap:
                cur.wait_stream(side)
            else:
                qlist = query_path(query)
            query = qlist[-1]
            masks = [O.MaskLogitsFn.apply(q[:, 0], gfeat) for q in qlist]
            top_idx, pos_i = O.argmax_gather(masks[-1].detach(), ghost)
            position_pyramid.append(pos_i[:, None])
            ghost_pcd_pyramid.append(ghost.transpose(1, 2))
            ghost_pcd_masks_pyramid.append(masks)
            ghost_features_pyramid.append(gfeat)
            prev_pos = pos_i if teacher_positions is None else teacher_positions[i].to(device=device, dtype=torch.float32)

        # ---- action (act3d.py:323-338, 507-535)
        position = position_pyramid[-1][:, 0]
        offsets = None
        if self.regress_position_offset:
            offsets = O.mlp(gfeat, self.ghost_point_offset_predictor[0], self.ghost_point_offset_predictor[2])   # (B, Ng, 3)
            position = position + O.SelectRowFn.apply(offsets, top_idx)
        if self.rotation_parametrization.endswith("from_top_ghost"):
            features = O.SelectRowFn.apply(gfeat, top_idx)
        else:
            features = query[:, 0]
        pred = O.mlp(features, self.gripper_state_predictor[0], self.gripper_state_predictor[2])
        if self.rotation_dim == 4:
            rotation, gripper = O.QuatSigmoidFn.apply(pred)
        else:
            rotation, gripper = O.Ortho6dSigmoidFn.apply(pred)
        return {
            "position": position,
            "rotation": rotation,
            "gripper": gripper,
            "position_pyramid": position_pyramid,
            "visible_rgb_mask_pyramid": [None] * L,
            "ghost_pcd_masks_pyramid": ghost_pcd_masks_pyramid,
            "ghost_pcd_pyramid": ghost_pcd_pyramid,
            "fine_ghost_pcd_offsets": None if offsets is None else offsets.transpose(1, 2),
            # token-major (B, ncam*h*w, E) tensors.  On the bf16 FPN path the output convolution's bias is owed to the
            # gathered rows only, so the entry is the ops.TokenMap pair: .with_bias() materialises the reference's map
            # (act3d.py:352) on demand instead of a 63 MB pass per step nobody reads
            "visible_rgb_features_pyramid": [f.tokens if f.row_bias is None else f for f in feats],
            "visible_pcd_pyramid": pcd_pyramid,
            "query_features": query.transpose(0, 1),            # (1, B, E) as the reference returns it
            "instruction_features": None if instr is None else instr.transpose(0, 1),
            "instruction_dummy_pos": None,
            # additions
            "topk_indices_pyramid": topk_pyramid,
            "ghost_features_pyramid": ghost_features_pyramid,
        }
