"""ORACLE (test infrastructure only) -- CPU restatement of the ChainedDiffuser trajectory denoiser.

DDPMSchedules restates diffusers' DDPMScheduler, which is NOT in /root/reference (un-pinned pip dependency,
README.md:29): PARITY UNPINNED at that boundary.  It follows Ho et al. 2020 eq. 6-7 with the constructor arguments
the reference passes (diffusion_model.py:51-60): num_train_timesteps=T, beta_schedule "scaled_linear" (position) /
"squaredcos_cap_v2" (rotation), prediction_type="sample", and the library defaults beta_start=1e-4, beta_end=0.02,
variance_type="fixed_small", clip_sample=True (range 1.0).  Self-checks of closed-form identities are in
tests/test_oracle_cpu.py.  Everything else cites /root/reference lines and is pinned by tests/golden.
"""
import math

import torch
import torch.nn.functional as F

from . import blocks as OB


class DDPMSchedules:
    def __init__(self, T=100):
        self.T = T
        betas_pos = torch.linspace(0.0001 ** 0.5, 0.02 ** 0.5, T, dtype=torch.float32) ** 2

        def alpha_bar(s):
            return math.cos((s + 0.008) / 1.008 * math.pi / 2) ** 2

        betas_rot = torch.tensor([min(1 - alpha_bar((i + 1) / T) / alpha_bar(i / T), 0.999) for i in range(T)],
                                 dtype=torch.float32)
        self.acp_pos = torch.cumprod(1.0 - betas_pos, dim=0)
        self.acp_rot = torch.cumprod(1.0 - betas_rot, dim=0)
        self.coef_pos = self._coef_table(self.acp_pos)
        self.coef_rot = self._coef_table(self.acp_rot)

    def _coef_table(self, acp):
        """[T][3] = (coef_x0, coef_xt, sigma) of the posterior q(x_{t-1} | x_t, x0); row 0 is unused by the loop."""
        one = torch.tensor(1.0)
        rows = []
        for t in range(self.T):
            a_t = acp[t]
            a_prev = acp[t - 1] if t > 0 else one
            b_t, b_prev = 1 - a_t, 1 - a_prev
            cur_a = a_t / a_prev
            cur_b = 1 - cur_a
            c_x0 = (a_prev ** 0.5 * cur_b) / b_t
            c_xt = cur_a ** 0.5 * b_prev / b_t
            var = torch.clamp(b_prev / b_t * cur_b, min=1e-20)
            sigma = var ** 0.5 if t > 0 else torch.tensor(0.0)
            rows.append(torch.stack([c_x0, c_xt, sigma]))
        return torch.stack(rows).to(torch.float32).contiguous()

    def add_noise(self, x0, noise, t):
        """DDPMScheduler.add_noise on channels [0:3] (position schedule) and [3:] (rotation schedule)
        (diffusion_model.py:296-305).  t: (B,) long."""
        out = torch.empty_like(x0)
        for sl, acp in ((slice(0, 3), self.acp_pos), (slice(3, None), self.acp_rot)):
            sa = (acp[t] ** 0.5).view(-1, 1, 1)
            sb = ((1 - acp[t]) ** 0.5).view(-1, 1, 1)
            out[..., sl] = sa * x0[..., sl] + sb * noise[..., sl]
        return out

    def step(self, model_out, sample, noise, t):
        """DDPMScheduler.step(...).prev_sample for x0-prediction with clipping, per channel group."""
        out = torch.empty_like(sample)
        for sl, cf in ((slice(0, 3), self.coef_pos), (slice(3, None), self.coef_rot)):
            x0 = model_out[..., sl].clamp(-1.0, 1.0)
            prev = cf[t, 0] * x0 + cf[t, 1] * sample[..., sl]
            if t > 0 and noise is not None:
                prev = prev + cf[t, 2] * noise[..., sl]
            out[..., sl] = prev
        return out

    def step_with_inpaint(self, model_out, sample, noise, cond_data, cond_mask, t):
        """One body of the reference's sampling loop (diffusion_model.py:106-117): inpaint, then step; the final
        iteration (t == 0) returns the inpainted network output itself."""
        out = model_out.clone()
        out[cond_mask] = cond_data[cond_mask]
        if t == 0:
            return out
        return self.step(out, sample, noise, t)


# ------------------------------------------------------------------------------------------ rotation conversions
def normalise_quat(x):
    """model/utils/utils.py:51-52"""
    return x / torch.clamp(x.square().sum(dim=-1).sqrt().unsqueeze(-1), min=1e-10)


def quaternion_to_matrix(q):
    """utils/pytorch3d_transforms.py:44-73 (real part first: r, i, j, k)."""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def _sqrt_positive_part(x):
    ret = torch.zeros_like(x)
    pos = x > 0
    ret[pos] = torch.sqrt(x[pos])
    return ret


def matrix_to_quaternion(matrix):
    """utils/pytorch3d_transforms.py:105-165: pick the best-conditioned of the four candidate quaternions."""
    batch_dim = matrix.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(matrix.reshape(batch_dim + (9,)), dim=-1)
    q_abs = _sqrt_positive_part(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22,
                                             1.0 - m00 - m11 + m22], dim=-1))
    quat_by_rijk = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    flr = torch.tensor(0.1).to(dtype=q_abs.dtype, device=q_abs.device)
    cand = quat_by_rijk / (2.0 * q_abs[..., None].max(flr))
    return cand[F.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5, :].reshape(batch_dim + (4,))


def ortho6d_from_matrix(m):
    """model/utils/utils.py:133-139: first two COLUMNS of the rotation matrix, flattened."""
    return m[..., :, :2].transpose(-1, -2).flatten(-2)


def matrix_from_ortho6d(o):
    """model/utils/utils.py:117-130 (Zhou et al. 2019): x = norm(a1), z = norm(x x a2), y = z x x; columns x,y,z."""
    x_raw, y_raw = o[..., 0:3], o[..., 3:6]

    def nrm(v):
        return v / torch.clamp(v.pow(2).sum(-1, keepdim=True).sqrt(), min=1e-8)

    x = nrm(x_raw)
    z = nrm(torch.cross(x, y_raw, dim=-1))
    y = torch.cross(z, x, dim=-1)
    return torch.stack((x, y, z), dim=-1)


def convert_rot(signal):
    """diffusion_model.py:197-212 with rotation_parametrization='6D': (.., 3+4) -> (.., 3+6)."""
    q = normalise_quat(signal[..., 3:7])
    r6 = ortho6d_from_matrix(quaternion_to_matrix(q))
    return torch.cat([signal[..., :3], r6, signal[..., 7:]], dim=-1)


def unconvert_rot(signal):
    """diffusion_model.py:214-230: (.., 3+6) -> (.., 3+4)."""
    quat = matrix_to_quaternion(matrix_from_ortho6d(signal[..., 3:9]))
    return torch.cat([signal[..., :3], quat, signal[..., 9:]], dim=-1)


def normalize_pos(pos, bounds):
    """diffusion_model.py:187-190"""
    lo, hi = OB._lift(bounds[0].float()), OB._lift(bounds[1].float())
    return (pos - lo) / (hi - lo) * 2.0 - 1.0


def unnormalize_pos(pos, bounds):
    """diffusion_model.py:192-195"""
    lo, hi = OB._lift(bounds[0].float()), OB._lift(bounds[1].float())
    return (pos + 1.0) / 2.0 * (hi - lo) + lo


# ------------------------------------------------------------------------------------------ the prediction head
def head_context(P, ctx_feats, instruction, H, n_vl_layers=2, pre="prediction_head.", drop=None):
    """Step-invariant part of DiffusionHead.forward (diffusion_head.py:222-232, 290-314): instruction encoding and
    the vision->language attention over the visual tokens.  ctx_feats (B, S_vis, E)."""
    instr = F.linear(instruction, P[pre + "instruction_encoder.weight"], P[pre + "instruction_encoder.bias"])
    ctx = OB.parallel_attention(P, pre + "vl_attention.0", n_vl_layers, ctx_feats, None, instr, H, self_attn=False,
                                use_adaln=False, drop=drop, name_root=pre)
    return ctx, instr


def head_forward(P, trajectory, traj_mask, timestep, ctx_feats, ctx_xyz, curr_gripper, goal_gripper, instruction, H,
                 n_traj_layers=4, pre="prediction_head.", drop=None):
    """DiffusionHead.forward / _one_attention_round (diffusion_head.py:200-363) for the script configuration
    (use_instruction, use_goal, 1 scale, 1 round, 6D rotations).  Inputs are already normalised / converted;
    ctx_feats (B, C*1024, E) are the FPN res3 tokens, ctx_xyz their down-sampled (normalised) coordinates.
    Returns the single-element prediction list's tensor (B, L, 9).  drop: oracle.sampling.DropoutTwin for training mode
    (p = 0.1 everywhere in the reference), None for eval."""
    E = ctx_feats.shape[-1]
    B, Ln, _ = trajectory.shape
    tf = OB.mlp2(trajectory, P, pre + "traj_encoder", "0", "3", drop=drop, name_root=pre)
    traj_xyz = trajectory[..., :3]
    time_feats = OB.sinusoidal(timestep, E)
    ctx, instr = head_context(P, ctx_feats, instruction, H, pre=pre, drop=drop)
    cg = F.linear(curr_gripper, P[pre + "curr_gripper_encoder.weight"], P[pre + "curr_gripper_encoder.bias"])[:, None] \
        + P[pre + "curr_gripper_embed.weight"][None]
    gg = F.linear(goal_gripper, P[pre + "goal_gripper_encoder.weight"], P[pre + "goal_gripper_encoder.bias"])[:, None] \
        + P[pre + "goal_gripper_embed.weight"][None]
    ctx = torch.cat([ctx, cg, gg], dim=1)
    cxyz = torch.cat([ctx_xyz, curr_gripper[:, None, :3], goal_gripper[:, None, :3]], dim=1)
    sem = OB.sinusoidal(torch.arange(Ln, dtype=torch.float32), E)[None].expand(B, -1, -1)
    tf = OB.parallel_attention(P, pre + "traj_lang_attention.0", 1, tf, traj_mask, instr, H, seq1_sem=sem,
                               self_attn=False, apply_ffn=False, use_adaln=False, drop=drop, name_root=pre)
    kw = dict(seq1_xyz=traj_xyz, seq2_xyz=cxyz, seq1_sem=sem, ada=time_feats, drop=drop, name_root=pre)
    tf = OB.parallel_attention(P, pre + "traj_attention.0", n_traj_layers, tf, traj_mask, ctx, H, **kw)
    pf = OB.parallel_attention(P, pre + "pos_attention.0", 2, tf, traj_mask, ctx, H, **kw)
    rf = OB.parallel_attention(P, pre + "rot_attention.0", 2, tf, traj_mask, ctx, H, **kw)
    upd = torch.cat([OB.mlp2(pf, P, pre + "pos_regressor.0", "0", "3", drop=drop, name_root=pre),
                     OB.mlp2(rf, P, pre + "rot_regressor.0", "0", "3", drop=drop, name_root=pre)], dim=-1)
    return torch.cat([traj_xyz + upd[..., :3], upd[..., 3:]], dim=-1)


def head_forward_multi(P, trajectory, traj_mask, timestep, ctx_feats_pyr, ctx_xyz_pyr, curr_gripper, goal_gripper, instruction,
                       H, attn_rounds=1, feat_scales=1, use_goal=True, n_traj_layers=4, pre="prediction_head."):
    """DiffusionHead.forward with attn_rounds / feat_scales_to_use > 1 (diffusion_head.py:249-275, eval mode / no dropout).
    Every (round, scale) iteration l = round * feat_scales + scale evaluates its OWN module set `*.{l}` on the SAME
    trajectory encoding and trajectory positions (the reference never feeds traj_feats / traj_pos forward, :286-288 are
    locals of _one_attention_round); only the running `trajectory` chains: xyz accumulates the updates, the rotation
    channels are replaced.  For scale > 0 with a goal the context is restricted to the nn_ * L fine tokens nearest to the
    PREVIOUS iteration's predicted trajectory (find_traj_nn, nn_ = 64 at scale 1, 16 beyond).  Returns the list."""
    from . import sampling as OS
    E = ctx_feats_pyr[0].shape[-1]
    B, Ln, _ = trajectory.shape
    tf0 = OB.mlp2(trajectory, P, pre + "traj_encoder", "0", "3", name_root=pre)
    traj_xyz = trajectory[..., :3]
    time_feats = OB.sinusoidal(timestep, E)
    instr = F.linear(instruction, P[pre + "instruction_encoder.weight"], P[pre + "instruction_encoder.bias"])
    cg = F.linear(curr_gripper, P[pre + "curr_gripper_encoder.weight"], P[pre + "curr_gripper_encoder.bias"])[:, None] \
        + P[pre + "curr_gripper_embed.weight"][None]
    extra, extra_xyz = [cg], [curr_gripper[:, None, :3]]
    if use_goal:
        gg = F.linear(goal_gripper, P[pre + "goal_gripper_encoder.weight"], P[pre + "goal_gripper_encoder.bias"])[:, None] \
            + P[pre + "goal_gripper_embed.weight"][None]
        extra.append(gg)
        extra_xyz.append(goal_gripper[:, None, :3])
    sem = OB.sinusoidal(torch.arange(Ln, dtype=torch.float32), E)[None].expand(B, -1, -1)
    outs, nn_indices = [], []
    for rnd in range(attn_rounds):
        for scale in range(feat_scales):
            l = rnd * feat_scales + scale
            feats, xyz = ctx_feats_pyr[scale], ctx_xyz_pyr[scale]
            if use_goal and scale > 0:
                idx_np, _ = OS.traj_nn_topk(outs[-1][..., :3].detach().numpy(), xyz.numpy(), (64 if scale == 1 else 16) * Ln)
                idx = torch.from_numpy(idx_np)
                nn_indices.append(idx)
                feats = torch.gather(feats, 1, idx[..., None].expand(-1, -1, E))
                xyz = torch.gather(xyz, 1, idx[..., None].expand(-1, -1, 3))
            ctx = OB.parallel_attention(P, pre + f"vl_attention.{l}", 2, feats, None, instr, H, self_attn=False,
                                        use_adaln=False, name_root=pre)
            ctx = torch.cat([ctx] + extra, dim=1)
            cxyz = torch.cat([xyz] + extra_xyz, dim=1)
            tf = OB.parallel_attention(P, pre + f"traj_lang_attention.{l}", 1, tf0, traj_mask, instr, H, seq1_sem=sem,
                                       self_attn=False, apply_ffn=False, use_adaln=False, name_root=pre)
            kw = dict(seq1_xyz=traj_xyz, seq2_xyz=cxyz, seq1_sem=sem, ada=time_feats, name_root=pre)
            tf = OB.parallel_attention(P, pre + f"traj_attention.{l}", n_traj_layers, tf, traj_mask, ctx, H, **kw)
            pf = OB.parallel_attention(P, pre + f"pos_attention.{l}", 2, tf, traj_mask, ctx, H, **kw)
            rf = OB.parallel_attention(P, pre + f"rot_attention.{l}", 2, tf, traj_mask, ctx, H, **kw)
            upd = torch.cat([OB.mlp2(pf, P, pre + f"pos_regressor.{l}", "0", "3", name_root=pre),
                             OB.mlp2(rf, P, pre + f"rot_regressor.{l}", "0", "3", name_root=pre)], dim=-1)
            prev = trajectory if not outs else outs[-1]
            outs.append(torch.cat([prev[..., :3] + upd[..., :3], upd[..., 3:]], dim=-1))
    return outs, nn_indices


def planner_loss(P, sched, gt_trajectory, traj_mask, ctx_feats, ctx_xyz_world, instruction, curr_gripper, goal_gripper,
                 bounds, noise, timesteps, H, ctx_xyz_norm=None, drop=None):
    """DiffusionPlanner.forward training branch (diffusion_model.py:253-324) with injected noise / timesteps.
    ctx_xyz_world: down-sampled point cloud in world metres (normalised here, which commutes with the bilinear
    down-sampling up to rounding)."""
    gt = gt_trajectory.clone()
    gt[..., :3] = normalize_pos(gt[..., :3], bounds)
    cxyz = ctx_xyz_norm if ctx_xyz_norm is not None else normalize_pos(ctx_xyz_world, bounds)
    cg, gg = curr_gripper.clone(), goal_gripper.clone()
    cg[:, :3] = normalize_pos(cg[:, :3], bounds)
    gg[:, :3] = normalize_pos(gg[:, :3], bounds)
    gt, cg, gg = convert_rot(gt), convert_rot(cg), convert_rot(gg)
    noisy = sched.add_noise(gt, noise, timesteps)
    pred = head_forward(P, noisy, traj_mask, timesteps, ctx_feats, cxyz, cg, gg, instruction, H, drop=drop)
    if OB.KINKS is not None:
        OB.KINKS.append(("l1", (pred - gt).detach().abs().min().item()))
    loss = 100 * F.l1_loss(pred[..., :3], gt[..., :3]) + 10 * F.l1_loss(pred[..., 3:9], gt[..., 3:9])
    return loss, pred, gt


def make_conditioning(traj_mask, curr_gripper9, goal_gripper9, use_goal_at_test=True):
    """diffusion_model.py:148-168: start pose at index 0; goal at index L - pad - 1 and everything after it."""
    B, Ln = traj_mask.shape
    cond = torch.zeros(B, Ln, curr_gripper9.shape[-1])
    mask = torch.zeros(B, Ln, curr_gripper9.shape[-1], dtype=torch.bool)
    cond[:, 0] = curr_gripper9
    mask[:, 0] = True
    if use_goal_at_test:
        for b in range(B):
            pad = int(traj_mask[b].sum())
            cond[b, Ln - pad - 1] = goal_gripper9[b]
            mask[b, Ln - pad - 1:] = True
    return cond, mask


def compute_trajectory(P, sched, traj_mask, ctx_feats, ctx_xyz_world, instruction, curr_gripper, goal_gripper, bounds,
                       init_noise, step_noise, H, n_steps=None, ctx_xyz_norm=None):
    """DiffusionPlanner.compute_trajectory + conditional_sample (diffusion_model.py:86-185) with injected noise:
    init_noise (B, L, 9), step_noise (T, B, L, 9) indexed by t.  n_steps limits the loop for fixtures (the first
    n_steps timesteps T-1, T-2, ...; the final un-normalisation is applied to whatever state is reached)."""
    cxyz = ctx_xyz_norm if ctx_xyz_norm is not None else normalize_pos(ctx_xyz_world, bounds)
    cg, gg = curr_gripper.clone(), goal_gripper.clone()
    cg[:, :3] = normalize_pos(cg[:, :3], bounds)
    gg[:, :3] = normalize_pos(gg[:, :3], bounds)
    cg, gg = convert_rot(cg), convert_rot(gg)
    cond, cmask = make_conditioning(traj_mask, cg, gg)
    traj = init_noise + cond
    steps = list(range(sched.T - 1, -1, -1))
    if n_steps is not None:
        steps = steps[:n_steps]
    trace = []
    for t in steps:
        out = head_forward(P, traj, traj_mask, torch.full((traj.shape[0],), t, dtype=torch.long), ctx_feats, cxyz, cg,
                           gg, instruction, H)
        traj = sched.step_with_inpaint(out, traj, step_noise[t], cond, cmask, t)
        trace.append(traj.clone())
    final = unconvert_rot(traj)
    final = torch.cat([unnormalize_pos(final[..., :3], bounds), final[..., 3:]], dim=-1)
    return final, trace


def traj_metrics(pred, gt):
    """TrajectoryCriterion.compute_metrics (main_trajectory.py:306-343): position error, symmetric quaternion L1 and
    their threshold rates over all steps ("traj_" keys), at the last step (plain keys), and per trajectory."""
    def errs(p, g):
        pos = (p[..., :3] - g[..., :3]).pow(2).sum(-1).sqrt()
        a, an = (p[..., 3:7] - g[..., 3:7]).abs().sum(-1), (p[..., 3:7] + g[..., 3:7]).abs().sum(-1)
        return pos, torch.where(a < an, a, an)
    pos, rot = errs(pred, gt)
    summary = {"traj_action_mse": F.mse_loss(pred, gt), "traj_pos_l2": pos.mean(), "traj_pos_acc_001": (pos < 0.01).float().mean(),
               "traj_rot_l1": rot.mean(), "traj_rot_acc_0025": (rot < 0.025).float().mean()}
    per = {"traj_pos_l2": pos.mean(-1), "traj_pos_acc_001": (pos < 0.01).float().mean(-1), "traj_rot_l1": rot.mean(-1),
           "traj_rot_acc_0025": (rot < 0.025).float().mean(-1)}
    pl, rl = errs(pred[:, -1], gt[:, -1])
    summary.update({"pos_l2": pl.mean(), "pos_acc_001": (pl < 0.01).float().mean(), "rot_l1": rl.mean(),
                    "rot_acc_0025": (rl < 0.025).float().mean()})
    return summary, per
