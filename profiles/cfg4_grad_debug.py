#!/usr/bin/env python3
"""Where does the device's gradient leave the oracle's at the configs[3] trajectory shape (B = 16, L = 50, S = 3074)?  Compares the
gradient of every ParallelAttentionLayer OUTPUT (device: forward hooks + retain_grad; oracle: a wrapper around
oracle.blocks.parallel_attention_layer) against the float64 oracle, for the loss linearised at the float64 sign pattern.
    python profiles/cfg4_grad_debug.py > gpurun_out/r06/cfg4_grad_chain.txt"""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import common as C
from oracle import diffusion as OD
from oracle import sampling as OS
from oracle import blocks as OB
from test_oracle_golden import _diffusion_params, load
a3d = importlib.import_module("act3d-chained-diffuser_amd")
dev = torch.device("cuda:0")
r = load("diffusion.pt")
E, B, Ln, ncam, H = 120, int(os.environ.get("DBG_B", "16")), int(os.environ.get("DBG_L", "50")), int(os.environ.get("DBG_CAM", "3")), 8
m = a3d.DiffusionPlanner(embedding_dim=E, output_dim=7, num_vis_ins_attn_layers=2, num_query_cross_attn_layers=6,
                         use_instruction=True, use_goal=True, use_goal_at_test=True, weight_tying=True,
                         gripper_loc_bounds=C.DIFFUSION_BOUNDS, rotation_parametrization="6D", diffusion_timesteps=100, dropout=0.0)
P = _diffusion_params(r)
m.load_state_dict(P, strict=False)
m.to(dev).train()
inp = C.trajectory_inputs(95, B, Ln, ncam, E, pad_last=6)
tokens = C.tokens_from_maps(inp["fmap"])
g = torch.Generator().manual_seed(95)
noise = torch.randn(B, Ln, 9, generator=g)
timesteps = torch.randint(0, 100, (B,), generator=g)
d = {k: v.to(dev) for k, v in inp.items()}
dev_out = {}
head = m.prediction_head
for stack in ("traj_lang_attention", "traj_attention", "pos_attention", "rot_attention", "vl_attention"):
    mod = getattr(head, stack, None)
    if mod is None:
        continue
    for i, lay in enumerate(mod[0].layers):
        def hook(_m, _i, out, key=f"prediction_head.{stack}.0.layers.{i}"):
            if out.requires_grad:
                out.retain_grad()
            dev_out[key] = out
        lay.register_forward_hook(hook)
O_ = a3d.ops
blk_out = []
_ab, _mlp = O_.attn_block, O_.mlp
def ab(*a_, **k_):
    o = _ab(*a_, **k_)
    if o.requires_grad: o.retain_grad()
    blk_out.append(("attn", o))
    return o
def mlp(*a_, **k_):
    o = _mlp(*a_, **k_)
    if o.requires_grad: o.retain_grad()
    blk_out.append(("mlp", o))
    return o
O_.attn_block, O_.mlp = ab, mlp
a3d.nn.O.attn_block, a3d.nn.O.mlp = ab, mlp
loss, pred, gt = m(d["trajectory"], d["mask"], None, d["pcd"], d["instr"], d["curr_gripper"], d["goal_gripper"], noise=noise.to(dev),
                   timesteps=timesteps.to(dev), visual_tokens=tokens.to(dev), return_pred=True)
bounds = torch.from_numpy(C.DIFFUSION_BOUNDS)
pcdn = OD.normalize_pos(inp["pcd"].permute(0, 1, 3, 4, 2), bounds).permute(0, 1, 4, 2, 3).contiguous()
cxyz_n = torch.from_numpy(OS.pcd_downsample(pcdn.numpy(), 8))
orig = OB.parallel_attention_layer


def oracle(dt):
    outs = {}
    lns = []
    orig_ln = OB.layer_norm

    def ln(*a, **k):
        o = orig_ln(*a, **k)
        if o.requires_grad:
            o.retain_grad()
        lns.append(o)
        return o
    OB.layer_norm = ln

    def wrapped(P_, prefix, *a, **k):
        o = orig(P_, prefix, *a, **k)
        o.retain_grad()
        outs[prefix] = o
        return o
    OB.parallel_attention_layer = wrapped
    OB.LIFT = None if dt == torch.float32 else dt
    try:
        leaf, Po = {}, {}
        for n, t in P.items():
            if id(t) not in leaf:
                leaf[id(t)] = (t.to(dt) if t.dtype.is_floating_point else t).clone().requires_grad_(t.dtype.is_floating_point)
            Po[n] = leaf[id(t)]
        cv = lambda x: x.to(dt) if x.dtype.is_floating_point else x
        ol, op, og = OD.planner_loss(Po, OD.DDPMSchedules(100), cv(inp["trajectory"]), inp["mask"], cv(tokens), None, cv(inp["instr"]),
                                     cv(inp["curr_gripper"]), cv(inp["goal_gripper"]), cv(bounds), cv(noise), timesteps, H, ctx_xyz_norm=cv(cxyz_n))
        return Po, ol, op, og, (outs, lns)
    finally:
        OB.layer_norm = orig_ln
        OB.LIFT = None
        OB.parallel_attention_layer = orig


Po32, l32, p32, g32, (o32, ln32) = oracle(torch.float32)
Po64, l64, p64, g64, (o64, ln64) = oracle(torch.float64)
resid = (p64 - g64).detach()
w = torch.empty_like(resid); w[..., :3] = 100.0 / resid[..., :3].numel(); w[..., 3:] = 10.0 / resid[..., 3:].numel()
up = torch.sign(resid) * w
(pred * up.float().to(dev)).sum().backward()
(p32 * up.float()).sum().backward()
(p64 * up).sum().backward()
print("layer-output gradients vs the float64 oracle (relative L2 | max err / scale): device ... fp32 oracle")
for key in sorted(o64):
    gx = o64[key].grad
    if gx is None or key not in dev_out or dev_out[key].grad is None:
        print("%-55s (no gradient captured)" % key)
        continue
    gd = dev_out[key].grad.detach().double().cpu()
    go = o32[key].grad.double()
    sc = gx.abs().max().item()
    fo = (dev_out[key].detach().double().cpu() - o64[key].detach()).abs().max().item() / o64[key].detach().abs().max().item()
    print("%-55s dev %.2e | %.2e   o32 %.2e | %.2e   (forward output: dev %.2e of scale)" % (
        key, ((gd - gx).norm() / gx.norm()).item(), (gd - gx).abs().max().item() / sc, ((go - gx).norm() / gx.norm()).item(),
        (go - gx).abs().max().item() / sc, fo))

# sub-block outputs (each ends in a LayerNorm): match the device's attn_block / mlp outputs with the oracle's layer_norm outputs by shape and value
print("\nsub-block output gradients (device attn_block / mlp outputs matched to oracle LayerNorm outputs by forward value):")
dev_blocks = [(k, o) for k, o in blk_out if o.dim() == 3 and o.shape[-1] == E]
used = set()
for bi, (kind, o) in enumerate(dev_blocks):
    od = o.detach().double().cpu()
    best = None
    for li, lo in enumerate(ln64):
        if li in used or tuple(lo.shape) != tuple(od.shape):
            continue
        e = (lo.detach() - od).abs().max().item()
        if best is None or e < best[0]:
            best = (e, li)
    if best is None or best[0] > 1e-3:
        continue
    used.add(best[1])
    lo = ln64[best[1]]
    if lo.grad is None or o.grad is None:
        print("%3d %-5s shape %s: no gradient" % (bi, kind, tuple(od.shape)))
        continue
    gx, gd, go = lo.grad, o.grad.detach().double().cpu(), ln32[best[1]].grad.double()
    print("%3d %-5s shape %-18s fwd match %.1e   grad: dev %.2e | %.2e   o32 %.2e | %.2e" % (
        bi, kind, tuple(od.shape), best[0], ((gd - gx).norm() / gx.norm()).item(), (gd - gx).abs().max().item() / gx.abs().max().item(),
        ((go - gx).norm() / gx.norm()).item(), (go - gx).abs().max().item() / gx.abs().max().item()))
