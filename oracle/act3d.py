"""ORACLE (test infrastructure only) -- CPU restatement of Act3D's coarse-to-fine keypose forward and its loss.

Functional, batch-first, plain PyTorch-CPU fp32 (autograd gives the oracle gradients).  `P` maps the reference's
state-dict names to tensors, so the same dictionary drives the reference (golden generation), this oracle and the
product model.  Visual features enter as token-major tensors (B, Npts, E) per pyramid level, i.e. AFTER the frozen
backbone + FPN (third-party, parity unpinned -- SURVEY §8c/§8f-1).  Citations are into /root/reference.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import blocks as OB
from . import sampling as OS


def default_cfg(**kw):
    cfg = dict(E=60, H=4, levels=3, ncam=1, n_ghost_layers=2, n_query_layers=2, n_vis_ins_layers=2,
               ball_diameter=0.16, use_instruction=False, knn_per_cam=1024, rotation_parametrization="quat_from_query",
               regress_position_offset=False, ins_pos_emb=False,
               bounds=np.array([[-0.1101, -0.5558, 0.7129], [0.6481, 0.5184, 1.5116]]))
    cfg.update(kw)
    return cfg


def ball_diameters(d):
    """act3d.py:55-60"""
    return [None, d, d / 4.0, d / 16.0]


def act3d_forward(P, cfg, feats_pyramid, pcd_pyramid, curr_gripper, instruction=None, gt_action=None,
                  ghost_points=None, teacher_positions=None, num_ghost_points=333):
    """Act3D.forward (act3d.py:176-357) after _compute_visual_features.

    feats_pyramid[i]: (B, Npts_i, E) visual tokens of level i in (cam, h, w) raster order; pcd_pyramid[i]: (B, Npts_i, 3).
    ghost_points: optional list of (B, Ng, 3) tensors (injection); otherwise the reference's numpy sampler is used
      (consumes the global numpy RNG exactly like act3d.py:394-440).
    teacher_positions: optional list; teacher_positions[i] (B, 3) replaces the level-i prediction for every use at
      level i+1 (k-NN centre, query RoPE, sampling anchor when no gt_action) -- the per-level parity protocol of
      SURVEY §0 "chaotic argmax cascade".
    """
    E, H, levels, ncam = cfg["E"], cfg["H"], cfg["levels"], cfg["ncam"]
    B = curr_gripper.shape[0]
    gt_position = None if gt_action is None else gt_action[:, :3].detach()
    grip_xyz = curr_gripper[:, :3]
    diam = ball_diameters(cfg["ball_diameter"])

    instr = None
    if cfg["use_instruction"]:
        instr = F.linear(instruction, P["instruction_encoder.weight"], P["instruction_encoder.bias"])   # :199
        if cfg["ins_pos_emb"]:                                                                          # :201-209
            pe = F.layer_norm(P["instr_position_embedding.weight"], (E,), P["instr_position_norm.weight"],
                              P["instr_position_norm.bias"])
            instr = instr + pe[None]

    grip_tok = P["curr_gripper_embed.weight"].expand(B, 1, E)                                             # :220
    out = dict(position_pyramid=[], ghost_pcd_pyramid=[], ghost_pcd_masks_pyramid=[], topk_indices=[],
               ghost_features=[], query_features_pyramid=[])
    query = None
    prev_pos = None
    for i in range(levels):
        # ---- ghost points (act3d.py:229-234, 394-440)
        if ghost_points is not None:
            ghost = ghost_points[i]
        else:
            anchor = None
            if i > 0:
                anchor = (gt_position if gt_position is not None else prev_pos).cpu().numpy()
            ghost = torch.from_numpy(OS.ref_sample_ghost_points(cfg["bounds"], B, num_ghost_points, i, anchor, diam[i]))
        # ---- context selection (:236-260)
        if i == 0:
            ctx_vis, ctx_xyz = feats_pyramid[0], pcd_pyramid[0]
            idx = None
        else:
            idx_np, _ = OS.knn_topk(prev_pos.detach().numpy(), pcd_pyramid[i].numpy(), cfg["knn_per_cam"] * ncam)
            idx = torch.from_numpy(idx_np)
            ctx_vis = torch.gather(feats_pyramid[i], 1, idx[..., None].expand(-1, -1, E))
            ctx_xyz = torch.gather(pcd_pyramid[i], 1, idx[..., None].expand(-1, -1, 3))
        out["topk_indices"].append(idx)
        ctx = torch.cat([ctx_vis, grip_tok], dim=1)
        ctx_xyz = torch.cat([ctx_xyz, grip_xyz[:, None]], dim=1)
        if cfg["use_instruction"]:                                                                          # :261-270
            ctx = OB.rel_cross_attn_module(P, f"vis_ins_attn_pyramid.{i}", cfg["n_vis_ins_layers"], ctx, instr, H)[-1]
            ctx = torch.cat([ctx, instr], dim=1)
            ctx_xyz = torch.cat([ctx_xyz, torch.zeros(B, instr.shape[1], 3)], dim=1)
        # ---- ghost features (:442-465): every ghost point starts from the same embedding row
        g0 = P[f"ghost_points_embed_pyramid.{i}.weight"].expand(B, ghost.shape[1], E)
        gfeat = OB.rel_cross_attn_module(P, f"ghost_point_cross_attn_pyramid.{i}", cfg["n_ghost_layers"], g0, ctx, H,
                                         ghost, ctx_xyz)[-1]
        # ---- query (:280-301): no positions at level 0, RoPE(previous position) afterwards
        if i == 0:
            query = P["query_embed.weight"].expand(B, 1, E)
            qlist = OB.rel_cross_attn_module(P, f"query_cross_attn_pyramid.{i}", cfg["n_query_layers"], query, ctx, H)
        else:
            qlist = OB.rel_cross_attn_module(P, f"query_cross_attn_pyramid.{i}", cfg["n_query_layers"], query, ctx, H,
                                             prev_pos[:, None], ctx_xyz)
        query = qlist[-1]
        # ---- mask over ghost points + argmax (:482-505, :312-314)
        masks = [torch.einsum("bc,bnc->bn", q[:, 0], gfeat) for q in qlist]
        top_idx = torch.max(masks[-1], dim=-1).indices
        pos_i = ghost[torch.arange(B), top_idx]
        out["position_pyramid"].append(pos_i[:, None])
        out["ghost_pcd_pyramid"].append(ghost.transpose(1, 2))           # (B, 3, Ng) as the reference returns it
        out["ghost_pcd_masks_pyramid"].append(masks)
        out["ghost_features"].append(gfeat)
        out["query_features_pyramid"].append(query)
        prev_pos = pos_i.detach() if teacher_positions is None else teacher_positions[i]
    # ---- offsets of the last level's ghost points (:323-327) and the action head (:507-535)
    offsets = None
    position = out["position_pyramid"][-1][:, 0]
    ar = torch.arange(B)
    if cfg["regress_position_offset"]:
        offsets = OB.mlp2(gfeat, P, "ghost_point_offset_predictor", "0", "2")              # (B, Ng, 3)
        position = position + offsets[ar, top_idx]
    rp = cfg["rotation_parametrization"]
    features = gfeat[ar, top_idx] if rp.endswith("from_top_ghost") else query[:, 0]
    pred = OB.mlp2(features, P, "gripper_state_predictor", "0", "2")
    if rp.startswith("quat"):
        nrot = 4
        rot = pred[:, :4] / torch.clamp(pred[:, :4].square().sum(-1).sqrt().unsqueeze(-1), min=1e-10)
    else:
        nrot = 6
        rot = ortho6d_to_matrix(pred[:, :6])
    out.update(position=position, rotation=rot, gripper=torch.sigmoid(pred[:, nrot:]), query_features=query, pred_raw=pred,
               fine_ghost_pcd_offsets=None if offsets is None else offsets.transpose(1, 2))
    return out


def ortho6d_to_matrix(d6):
    """compute_rotation_matrix_from_ortho6d (model/utils/utils.py:93-130): Gram-Schmidt frame with columns x, y, z."""
    def unit(v):
        return v / torch.clamp(v.pow(2).sum(1).sqrt(), min=1e-8)[:, None]
    x = unit(d6[:, 0:3])
    z = unit(torch.cross(x, d6[:, 3:6], dim=1))
    y = torch.cross(z, x, dim=1)
    return torch.stack([x, y, z], dim=2)


def keypose_loss(out, gt_action, spread=0.01, position_loss_coeff=1.0, rotation_loss_coeff=10.0,
                 gripper_loss_coeff=1.0, label_smoothing=0.0, position_loss="ce", position_offset_loss_coeff=10000.0):
    """LossAndMetrics.compute_loss with position_loss="ce" / "ce+mse" (main_keypose.py:353-429), incl. the supervised
    offsets of regress_position_offset (:407-419)."""
    gt_pos = gt_action[:, :3]
    losses = {}
    L = len(out["ghost_pcd_masks_pyramid"])
    for i, (ghost, masks) in enumerate(zip(out["ghost_pcd_pyramid"], out["ghost_pcd_masks_pyramid"])):
        l2 = ((ghost - gt_pos.unsqueeze(-1)) ** 2).sum(1).sqrt()
        label = torch.softmax(-l2 / spread, dim=-1).detach()
        losses[f"position_ce_level{i}"] = F.cross_entropy(masks[-1], label, label_smoothing=label_smoothing).mean() \
            * position_loss_coeff / L
    if out.get("fine_ghost_pcd_offsets") is not None:
        pts = out["ghost_pcd_pyramid"][-1] + out["fine_ghost_pcd_offsets"]
        losses["position_offset"] = F.mse_loss(pts, gt_pos.unsqueeze(-1).expand_as(pts)) \
            * (position_offset_loss_coeff * position_loss_coeff)
    if position_loss == "ce+mse":
        losses["position_mse"] = F.mse_loss(out["position"], gt_pos) * position_loss_coeff
    losses["rotation"] = F.mse_loss(out["rotation"], gt_action[:, 3:7]) * rotation_loss_coeff
    losses["gripper"] = F.mse_loss(out["gripper"], gt_action[:, 7:8]) * gripper_loss_coeff
    return losses


def keypose_metrics(out, gt_action, tasks=None, symmetric=False):
    """LossAndMetrics.compute_metrics (main_keypose.py:431-482); per-task entries when `tasks` (list of names) is given;
    symmetric: rotation error against the closer of +-gt (symmetric_rotation_loss, :463-470)."""
    m = {}
    l2 = ((out["position"] - gt_action[:, :3]) ** 2).sum(1).sqrt()
    m["mean/pos_l2_final"] = l2.mean()
    m["mean/pos_l2_final<0.01"] = (l2 < 0.01).float().mean()
    for i, p in enumerate(out["position_pyramid"]):
        m[f"mean/pos_l2_level{i}"] = ((p.squeeze(1) - gt_action[:, :3]) ** 2).sum(1).sqrt().mean()
    m["gripper"] = ((out["gripper"] > 0.5).squeeze(-1) == gt_action[:, 7].bool()).float().mean()
    l1 = (out["rotation"] - gt_action[:, 3:7]).abs().sum(1)
    if symmetric:
        l1 = torch.minimum(l1, (out["rotation"] + gt_action[:, 3:7]).abs().sum(1))
    m["mean/rot_l1"] = l1.mean()
    m["mean/rot_l1<0.05"] = (l1 < 0.05).float().mean()
    m["mean/rot_l1<0.025"] = (l1 < 0.025).float().mean()
    if tasks is not None:
        names = np.asarray(tasks)
        for t in np.unique(names):
            sel = torch.from_numpy(names == t)
            m[f"{t}/pos_l2_final"] = l2[sel].mean()
            m[f"{t}/pos_l2_final<0.01"] = (l2[sel] < 0.01).float().mean()
            m[f"{t}/rot_l1"] = l1[sel].mean()
            m[f"{t}/rot_l1<0.05"] = (l1[sel] < 0.05).float().mean()
            m[f"{t}/rot_l1<0.025"] = (l1[sel] < 0.025).float().mean()
    return m


def keypose_optional_losses(out, gt_action, symmetric, position_loss_coeff=1.0, rotation_loss_coeff=10.0):
    """The non-default branches of LossAndMetrics.compute_loss: position_loss="mse" (main_keypose.py:383-385) and
    symmetric_rotation_loss (:370-376: per-sample minimum of the MSE against gt and against -gt)."""
    losses = {"position_mse": F.mse_loss(out["position"], gt_action[:, :3]) * position_loss_coeff}
    gq = gt_action[:, 3:7]
    if symmetric:
        a = (out["rotation"] - gq).pow(2).mean(1)
        an = (out["rotation"] + gq).pow(2).mean(1)
        losses["rotation"] = torch.where(a < an, a, an).mean() * rotation_loss_coeff
    else:
        losses["rotation"] = F.mse_loss(out["rotation"], gq) * rotation_loss_coeff
    return losses


def optimizer_groups(named_params):
    """engine.py:89-102: names containing "bias" (or the never-matching "LayerNorm.*") -> weight decay 0; the
    rest (including every norm weight, which is called `norm*.weight`) -> weight decay 5e-4."""
    no_decay = ["bias", "LayerNorm.weight", "LayerNorm.bias"]
    g0, g1 = [], []
    for name, p in named_params:
        (g0 if any(nd in name for nd in no_decay) else g1).append(name)
    return g0, g1
