#!/usr/bin/env python3
"""ChainedDiffuser workloads of the benchmark (BASELINE.json configs[2] and the diffusion half of the metric): used by
bench.py for the `secondary` entries of its JSON line, and runnable on its own:

  python bench_denoise.py [--mode sample|train] [--batch 64] [--horizon 16] [--cams 3] [--reps 5] [--no-graph]

  sample: DiffusionPlanner.compute_trajectory, 100 denoise steps, context + K/V cache built once, loop hipGraph-captured
          -> trajectories/s; roofline = the cross-attention against the K/V cache (HBM-bound) on
          ALGORITHMIC bytes (64 B per key and head: a 16-channel bf16 K row + V row).
  train : one training step of main_trajectory.py:177-204 -- zero_grad + frozen backbone (bf16) + FPN + denoiser forward
          with the reference's dropout 0.1 + L1 loss + backward + fused AdamW, hipGraph-captured (noise, timesteps and
          dropout masks drawn on the device inside the graph) -> trajectories/s; roofline = the trajectory->context
          cross-attention forward + backward (MFMA-bound) on algorithmic FLOPs.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
DIFFUSION_BOUNDS = np.array([[-0.7342, -0.7915, 0.7098], [0.6944, 0.8437, 1.8645]])
E, H = 120, 8


def build_planner(a3d, dev, train):
    torch.manual_seed(0)
    m = a3d.DiffusionPlanner(embedding_dim=E, output_dim=7, num_vis_ins_attn_layers=2, num_query_cross_attn_layers=6,
                             use_instruction=True, use_goal=True, use_goal_at_test=True, weight_tying=True,
                             gripper_loc_bounds=DIFFUSION_BOUNDS, rotation_parametrization="6D", diffusion_timesteps=100,
                             dropout=0.1).to(dev)        # scripts/train_trajectory.sh:21-31; dropout as the reference (layers.py:10)
    for mod in m.modules():                      # AdaLN is zero-initialised in the reference; give it non-trivial weights
        if isinstance(mod, a3d.nn.AdaLN):
            torch.nn.init.normal_(mod.modulation[1].weight, std=0.02)
    m.train(train)
    m.prediction_head.backbone_dtype = torch.bfloat16
    if os.environ.get("A3D_DIFFUSION_FPN_FP32", "0") != "1":       # A/B: the FPN in fp32 on fp32 copies of the backbone maps (rounds 1 - 5)
        m.prediction_head.fpn_dtype = torch.bfloat16
    return m


def synthetic_inputs(B, Ln, C, dev, seed=1):
    g = torch.Generator().manual_seed(seed)
    lo, hi = torch.tensor(DIFFUSION_BOUNDS[0], dtype=torch.float32), torch.tensor(DIFFUSION_BOUNDS[1], dtype=torch.float32)
    rgb = torch.rand(B, C, 3, 256, 256, generator=g).to(dev)
    pcd = (lo.view(1, 1, 3, 1, 1) + torch.rand(B, C, 3, 256, 256, generator=g) * (hi - lo).view(1, 1, 3, 1, 1)).to(dev)

    def pose(n):
        q = torch.randn(*n, 4, generator=g)
        return torch.cat([lo + 0.15 * (hi - lo) + torch.rand(*n, 3, generator=g) * 0.7 * (hi - lo), q / q.norm(dim=-1, keepdim=True)], -1)

    cg, gg = pose((B,)), pose((B,))
    w = torch.linspace(0, 1, Ln).view(1, Ln, 1)
    traj = cg[:, None] * (1 - w) + gg[:, None] * w + 0.01 * torch.randn(B, Ln, 7, generator=g)
    traj[..., 3:] = traj[..., 3:] / traj[..., 3:].norm(dim=-1, keepdim=True)
    return {"rgbs": rgb, "pcds": pcd, "curr_gripper": cg.to(dev), "action": gg.to(dev), "trajectory": traj.to(dev),
            "instr": torch.randn(B, 53, 512, generator=g).to(dev), "trajectory_mask": torch.zeros(B, Ln, dtype=torch.bool, device=dev)}


def _time(fn, iters):
    for _ in range(3):
        fn()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e-3


# HBM bytes per a3d_dn_cross launch at cfg-3 from the committed counter passes: FETCH_SIZE 111215.5 KiB x 2 (gfx950 wide-read
# correction) + WRITE_SIZE 1088 KiB
PMC_DN_CROSS_BYTES = (111215.5 * 2 + 1088.0) * 1024      # round 2 (superseded by the committed round-5 records read below)


# committed counter records of the persistent sampler, by (B, horizon, context tokens)
_PERSIST_PMC = {(64, 16, 3074): "r06_pmc_denoise_persist.json", (24, 50, 3074): "r06_pmc_denoise_persist_L50.json"}


def _pmc_bytes(fname, key):
    """HBM bytes per launch of `key` from a committed counter record (profiles/pmc_json_cmd.sh), or None"""
    import json
    try:
        with open(os.path.join(ROOT, "profiles", fname)) as fh:
            return json.load(fh)["kernels"][key]["hbm_bytes"]
    except Exception:
        return None


def cached_attention_roofline(a3d, B, Ln, S, dev, nlayers=8):
    """a3d_dn_cross (AdaLN + q-projection + RoPE + the L-query flash attention against the cached context) at the sampling
    shapes, timed with events on the launch stream AS THE SAMPLING LOOP RUNS IT: the launches rotate over the `nlayers`
    layers' distinct K/V caches (8 x 205 MB at cfg-3 -- a single cache re-read back to back would sit in the 256 MiB
    Infinity Cache and overstate the rate).  ALGORITHMIC bytes: a 16-channel bf16 K row and V row per key and
    head (64 B) + the query rows and the partial outputs; `stored_bytes` is what the cache actually holds (two-part fp16 K rows 64 B +
    two-part fp16 V planes 64 B per key and head: the operand formats of attention16.hip, round 6)."""
    import ctypes
    Lb = a3d.lib
    Sp = (S + 63) // 64 * 64
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, Ln, E, generator=g).to(dev)
    traj = torch.randn(B, Ln, 9, generator=g).to(dev)
    Kf = [torch.randn(B, H, Sp, 32, device=dev).to(torch.float16) for _ in range(nlayers)]        # rows16: hi | lo
    Vt = [torch.randn(B, H, 2, 16, Sp, device=dev).to(torch.float16) for _ in range(nlayers)]     # planes16: hi / lo
    qw, qb = (torch.randn(E, E, generator=g) / 11).to(dev), torch.randn(E, generator=g).to(dev)
    mod, sem = (torch.randn(2 * E, generator=g) * 0.1).to(dev), torch.randn(Ln, E, generator=g).to(dev)
    freq = a3d.ops.rope_freq(E, dev)
    ns = max(1, min(8, Sp // 128, -(-a3d.diffusion.DN_TARGET_WGS // (B * H))))       # the split the sampling loop uses
    ws = torch.empty((Lb.load().a3d_dn_cross_ws_floats(B, H, ns),), device=dev)
    cps = [Lb.DnCrossParams(sem=sem.data_ptr(), mod=mod.data_ptr(), q_w=qw.data_ptr(), q_b=qb.data_ptr(), freq=freq.data_ptr(),
                            Kf=Kf[i].data_ptr(), Vt=Vt[i].data_ptr()) for i in range(nlayers)]

    def one_round():
        for cp in cps:
            Lb.call("a3d_dn_cross", x.data_ptr(), traj.data_ptr(), 9, ctypes.byref(cp), ws.data_ptr(), B, Ln, E, H, S, Sp, ns,
                    Lb.stream())
    t = _time(one_round, 5) / nlayers
    alg = B * H * S * 64.0 + B * Ln * E * 4.0 + ns * B * H * 16 * 17 * 4.0
    stored = B * H * Sp * 128.0
    return {"bound": "hbm", "kernel": "dn_cross (trajectory -> context cross-attention against the two-part fp16 K / V cache)",
            "achieved": alg / t / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg / t / 1e9 / 8000.0, "ms": t * 1e3,
            "timing": f"mean over launches rotating through {nlayers} distinct caches ({nlayers * stored / 1e9:.2f} GB working set)",
            "traffic": _pmc_bytes("r05_pmc_denoise_perphase.json", "dn_cross") if (B, Ln, S) == (64, 16, 3074) else None,
            "traffic_source": "profiles/r05_pmc_denoise_perphase.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE per launch, same "
                              f"kernel and shapes, nsplit {ns})",
            "algorithmic_bytes_per_launch": alg, "stored_bytes_per_launch": stored,
            "stored_bytes_rate_GBps": stored / t / 1e9, "launches_per_denoise_step": 8, "nsplit": ns}


def training_attention_roofline(a3d, B, Ln, S, dev):
    """Forward + backward of the trajectory -> context cross-attention core at the training shapes (MFMA-bound):
    algorithmic FLOPs 4 Lq S E B forward (QK^T + PV) + 10 Lq S E B backward (five contractions)."""
    O = a3d.ops
    g = torch.Generator().manual_seed(1)
    q_pre = torch.randn(B * Ln, E, generator=g).to(dev)
    kv_pre = torch.randn(B * S, 2 * E, generator=g).to(dev)
    q_xyz, k_xyz = torch.rand(B, Ln, 3, generator=g).to(dev), torch.rand(B, S, 3, generator=g).to(dev)
    f16 = O.ATTN_MODE != "bf16x3"                     # the family the training step runs (ops.AttnBlockFn)
    build = O.attn_operands16 if f16 else O.attn_operands
    Qs, Ks, Vt, Lqp, Sp, scale, freq, extra = build(q_pre.data_ptr(), E, kv_pre.data_ptr(), 2 * E, kv_pre.data_ptr() + E * 4, 2 * E,
                                                    q_xyz, k_xyz, B, Ln, S, E, H, dev, need_bwd=True)
    ns = O.pick_nsplit(B, H, Lqp, Sp)
    Oo, LSE = O.attn_core_fwd(Qs, Ks, Vt, None, B, H, Ln, Lqp, S, Sp, ns)
    dO = torch.randn_like(Oo)
    tf = _time(lambda: O.attn_core_fwd(Qs, Ks, Vt, None, B, H, Ln, Lqp, S, Sp, ns), 20)
    tb = _time(lambda: O.attn_core_bwd(Qs, Ks, Vt, None, Oo, dO, LSE, B, H, Ln, Lqp, S, Sp, ns, extra=extra), 20)
    fl = 14.0 * Ln * S * E * B
    traffic, src = None, None
    for rnd in ("r06", "r04"):          # HBM bytes of the same micro-benchmark from the newest committed counter passes (profiles/pmc_json_cmd.sh)
        try:
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"{rnd}_pmc_diffusion_attn_B{B}_L{Ln}.json")
            with open(path) as fh:
                k = json.load(fh)["kernels"]
            traffic = k["attn_fwd"]["hbm_bytes"] + k["attn_bwd"]["hbm_bytes"]
            src = f"profiles/{rnd}_pmc_diffusion_attn_B{B}_L{Ln}.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, forward + backward launch)"
            break
        except Exception:
            continue
    return {"bound": "mfma", "kernel": "attn_fwd + attn_bwd (trajectory -> context cross-attention)", "traffic_source": src,
            "achieved": fl / (tf + tb) / 1e12, "peak": 2500.0, "unit": "TFLOP/s", "frac": fl / (tf + tb) / 1e12 / 2500.0,
            "ms": (tf + tb) * 1e3, "ms_fwd": tf * 1e3, "ms_bwd": tb * 1e3, "traffic": traffic, "launches_per_step": 8,
            "dtype": "fp16 / bf16 MFMA on two-part operands" if f16 else "bf16 (split operands)", "family": O.ATTN_MODE}


def sampling_bench(a3d, dev, B=64, Ln=16, C=3, reps=5, graph=True):
    m = build_planner(a3d, dev, train=False)
    s = synthetic_inputs(B, Ln, C, dev)
    with torch.no_grad():
        tokens = m.prediction_head.encode_images(s["rgbs"], None).contiguous()      # one-off per trajectory batch (adjacent)
    kw = dict(visual_tokens=tokens, use_graph=graph)

    def run():
        return m.compute_trajectory(s["trajectory_mask"], None, s["pcds"], s["instr"], s["curr_gripper"], s["action"],
                                    init_noise=torch.randn(B, Ln, 9, device=dev),
                                    step_noise=torch.randn(100, B, Ln, 9, device=dev), **kw)

    out = run()
    out = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    assert torch.isfinite(out).all()
    # the timed path (captured graph, replayed) must reproduce the eager loop on the same noise: a replay that runs on stale
    # synchronisation state would be fast and wrong
    g = torch.Generator(device=dev).manual_seed(5)
    n0, n1 = torch.randn(B, Ln, 9, device=dev, generator=g), torch.randn(100, B, Ln, 9, device=dev, generator=g)
    args = (s["trajectory_mask"], None, s["pcds"], s["instr"], s["curr_gripper"], s["action"])
    ref = m.compute_trajectory(*args, init_noise=n0, step_noise=n1, visual_tokens=tokens, use_graph=False)
    got = m.compute_trajectory(*args, init_noise=n0, step_noise=n1, visual_tokens=tokens, use_graph=graph)
    torch.cuda.synchronize()
    replay_err = (got - ref).abs().max().item()
    assert replay_err <= 1e-4, "the timed (graph) path differs from the eager loop by %.3e" % replay_err
    S = C * 1024 + 2
    rl = cached_attention_roofline(a3d, B, Ln, S, dev) if Ln <= 16 else None
    # whole-loop view: per denoise step every cross-attention layer reads the sample's cached context once -- algorithmic bytes
    # (one 16-bit K row and V row per key and head) and the bytes the cache really holds (fp32 K + two-part bf16 V)
    nl = len(m.prediction_head._cross_layers())
    Sp = (S + 63) // 64 * 64
    alg_step, stored_step = nl * B * H * S * 64.0, nl * B * H * Sp * 128.0
    step_s = dt / 100.0
    loop = {"bound": "hbm", "kernel": "sampling loop (a3d_dn_persist: one launch)" if "persistent" in str(getattr(m, "last_sampler_path", "")) else "sampling loop",
            "achieved": alg_step / step_s / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg_step / step_s / 1e9 / 8000.0,
            "algorithmic_bytes_per_step": alg_step, "stored_bytes_per_step": stored_step, "stored_frac": stored_step / step_s / 1e9 / 8000.0,
            "traffic": (lambda v: None if v is None else v / 100.0)(_pmc_bytes(_PERSIST_PMC.get((B, Ln, S), "none"), "dn_persist"))
            if "persistent" in str(getattr(m, "last_sampler_path", "")) else None,
            "traffic_source": "profiles/%s: dn_persist_kernel, one launch = 100 steps (FETCH_SIZE x2 + WRITE_SIZE) / 100" % _PERSIST_PMC.get((B, Ln, S), "(no counter record for this shape)"),
            "note": "whole denoise step incl. the per-sample chain (head, 8 layer remainders, tail) and the per-call context build "
                                     "amortised over 100 steps; the streaming itself is VALU / MFMA-issue bound (exp2, bf16 split, fp32 QK^T), see DESIGN 4"}
    return {
        "metric": "DDPM trajectory sampling, 100 denoise steps (trajectories/s)", "value": B / dt, "unit": "trajectories/s",
        "ms_per_100_step_batch": dt * 1e3, "ms_per_denoise_step": dt * 10, "higher_is_better": True,
        "dtype": "f32 MFMA logits (fp32 K cache), bf16 MFMA PV on two-part operands, f32 MFMA dense layers", "data": "synthetic",
        "config": {"workload": f"ChainedDiffuser compute_trajectory: B={B}, horizon={Ln}, {C} cameras "
                               f"(S={S} context tokens), E=120, H=8, 100 steps, context + K/V cache built once per call (inside the timed region)",
                   "sampler": getattr(m, "last_sampler_path", None),
                   "hipgraph": graph, "graph_vs_eager_max_abs_diff": replay_err},
        "roofline": loop, "dn_cross_standalone": rl,
    }


def training_bench(a3d, dev, B=22, Ln=50, C=3, steps=10, warmup=3, graph=True):
    Eg = a3d.engine
    m = build_planner(a3d, dev, train=True)
    s = synthetic_inputs(B, Ln, C, dev)
    crit = a3d.TrajectoryCriterion()

    prefetch = graph and os.environ.get("A3D_PREFETCH_BACKBONE", "1") == "1"      # as bench.py: the next batch's frozen backbone inside this step's graph

    def fwd_bwd(sample):
        kw = {}
        if sample.get("backbone_maps") is not None:
            kw["visual_tokens"] = m.prediction_head.encode_images(sample["rgbs"], None, maps=sample["backbone_maps"])
        loss = crit.compute_loss(m(sample["trajectory"], sample["trajectory_mask"], sample["rgbs"], sample["pcds"],
                                   sample["instr"], sample["curr_gripper"], sample["action"], **kw))
        loss.backward()
        return loss.detach()

    active = Eg.discover_active_parameters(m, lambda: fwd_bwd(s))
    flat, opt = Eg.get_optimizer(m, lr=1e-4, active_names=active)
    graphed, err = None, None
    if graph:
        try:
            graphed = Eg.GraphedStep(fwd_bwd, opt, s, warmup=2, prefetch=m.prediction_head.backbone_maps if prefetch else None)
        except Exception as e:                               # recorded in the output: never a silent fallback
            err = repr(e)[:200]
            torch.cuda.synchronize()
            flat.rebind_grads()

    def step():
        if graphed is not None:
            return graphed()
        opt.zero_grad()
        loss = fwd_bwd(s)
        opt.step()
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert torch.isfinite(loss).all()
    S = C * 1024 + 2
    rl = training_attention_roofline(a3d, B, Ln, S, dev)
    res = {
        "metric": "train samples/sec (ChainedDiffuser trajectory-diffusion fwd+bwd+AdamW step)", "value": B / dt,
        "unit": "samples/s", "ms_per_step": dt * 1e3, "steps": steps, "warmup": warmup, "higher_is_better": True,
        "dtype": "fp16 / bf16 MFMA on two-part operands in attention; fp32-accurate linears (f32 MFMA, bf16x3 MFMA from 4096 rows); bf16 frozen backbone; fp32 FPN",
        "data": "synthetic",
        "config": {"workload": f"DiffusionPlanner training step (main_trajectory.py:177-204): B={B} trajectories, horizon "
                               f"{Ln}, {C} cameras 256x256 (S={S}), E=120, H=8, dropout 0.1, frozen synthetic CLIP-RN50-shaped "
                               "backbone + trainable FPN included", "hipgraph": graphed is not None,
                   "backbone_prefetch": bool(graphed is not None and prefetch),
                   "final_loss": float(loss.item())},
        "roofline": rl,
    }
    if err:
        res["config"]["graph_capture_error"] = err
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="sample", choices=["sample", "train", "attn"],
                    help="attn: only the trajectory -> context attention micro-benchmark of the training roofline (for counter passes)")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--horizon", type=int, default=None)
    ap.add_argument("--cams", type=int, default=3)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    a3d = importlib.import_module("act3d-chained-diffuser_amd")
    a3d.lib.load()
    if args.mode == "attn":
        res = training_attention_roofline(a3d, args.batch or 22, args.horizon or 50, args.cams * 1024 + 2, dev)
    elif args.mode == "sample":
        res = sampling_bench(a3d, dev, args.batch or 64, args.horizon or 16, args.cams, args.reps, not args.no_graph)
    else:
        res = training_bench(a3d, dev, args.batch or 22, args.horizon or 50, args.cams, args.reps, 3, not args.no_graph)
    res["n_gpus"] = 1
    print(json.dumps(res))


if __name__ == "__main__":
    main()
