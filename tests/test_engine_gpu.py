"""GPU tests of the training-step harness (engine.py) and of the full model path including the frozen backbone + FPN:
  * BASELINE.json configs[0] end to end -- one 128x128 camera, one ghost-point level, batch 1, RGB in -> action out,
    free-running device sampler -- against the CPU oracle fed with the same FPN tokens and ghost points;
  * engine.GraphedStep (the thing bench.py times) replays == eager engine.train_one_step, step for step."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import common as C  # noqa: E402
from oracle import act3d as OA  # noqa: E402
from oracle import sampling as OS  # noqa: E402

pytestmark = pytest.mark.gpu


def _sample(B, ncam, image, dev, seed):
    inp = C.keypose_inputs(seed, B, ncam, 60, 1, image=image)
    rs = np.random.RandomState(seed + 1)
    rgb = torch.from_numpy(rs.uniform(0, 1, size=(B, ncam, 3, image, image)).astype(np.float32))
    s = {"rgbs": rgb, "pcds": inp["pcd"], "instr": inp["instr"], "curr_gripper": inp["curr_gripper"], "action": inp["action"]}
    s = {k: v.to(dev) for k, v in s.items()}
    s["task"] = ["t"] * B
    return s


@pytest.mark.parametrize("train", [True, False])
def test_cfg1_end_to_end_128px_one_level(a3d, dev, train):
    torch.manual_seed(0)
    Ng = 1000 if train else 10000
    m = a3d.Act3D(image_size=(128, 128), embedding_dim=60, num_attn_heads=4, gripper_loc_bounds=C.PERACT_BOUNDS,
                  num_ghost_points=1000, num_ghost_points_val=10000, num_sampling_level=1, sampler_seed=11).to(dev)
    assert m.feature_map_pyramid[0] == "res2" and m.downscaling_factor_pyramid[0] == 4        # act3d.py:78-82
    m.train(train)
    s = _sample(1, 1, 128, dev, 21)
    with torch.set_grad_enabled(train):
        out = m(s["rgbs"], s["pcds"], s["instr"], s["curr_gripper"], gt_action=s["action"] if train else None)
    feats = out["visible_rgb_features_pyramid"][0]
    assert feats.shape == (1, 32 * 32, 60)                                                   # res2 map at 1/4 resolution
    ghost = out["ghost_pcd_pyramid"][0].transpose(1, 2).contiguous()
    assert ghost.shape == (1, Ng, 3)
    lo, hi = torch.tensor(C.PERACT_BOUNDS[0], device=dev), torch.tensor(C.PERACT_BOUNDS[1], device=dev)
    assert ((ghost >= lo) & (ghost <= hi)).all(), "ghost points outside the workspace"
    if train:
        feats.retain_grad()
    # oracle on the same tokens / points
    P = {}
    for k, v in m.state_dict().items():
        if not k.startswith("backbone") and "feature_pyramid" not in k:
            P[k] = v.detach().cpu().clone().requires_grad_(train)
    of = feats.detach().cpu().clone().requires_grad_(train)
    pcds = [torch.from_numpy(OS.pcd_downsample(s["pcds"].cpu().numpy(), 4))]
    cfg = OA.default_cfg(E=60, levels=1, ncam=1)
    with torch.set_grad_enabled(train):
        oout = OA.act3d_forward(P, cfg, [of], pcds, s["curr_gripper"].cpu(), None,
                                gt_action=s["action"].cpu() if train else None, ghost_points=[ghost.cpu()])
    for l in range(2):
        got, ref = out["ghost_pcd_masks_pyramid"][0][l].detach().cpu(), oout["ghost_pcd_masks_pyramid"][0][l].detach()
        err = (got - ref).abs().max().item()
        print(f"[parity] cfg1 e2e mask layer{l}: max_abs_err={err:.3e} ref_absmax={ref.abs().max().item():.3e}")
        assert err <= 1e-3 * max(1.0, ref.abs().max().item())
    top2 = oout["ghost_pcd_masks_pyramid"][0][-1].detach().topk(2, -1).values
    if (top2[:, 0] - top2[:, 1]).min() > 1e-3:
        assert torch.equal(out["position"].cpu(), oout["position"].detach()), "argmax ghost point"
    assert (out["rotation"].detach().cpu() - oout["rotation"].detach()).abs().max() < 1e-3
    assert (out["gripper"].detach().cpu() - oout["gripper"].detach()).abs().max() < 1e-3
    if not train:
        return
    crit = a3d.LossAndMetrics(position_loss="ce", rotation_parametrization="quat_from_query", ground_truth_gaussian_spread=0.01)
    loss = sum(crit.compute_loss(out, s).values())
    oloss = sum(OA.keypose_loss(oout, s["action"].cpu()).values())
    assert abs(loss.item() - oloss.item()) <= 1e-3 * max(1.0, abs(oloss.item()))
    loss.backward()
    oloss.backward()
    named = dict(m.named_parameters())
    for n, p in P.items():
        if p.grad is None or n not in named:
            continue
        ref = p.grad
        err = (named[n].grad.cpu() - ref).abs().max().item()
        assert err <= 1.5e-3 * ref.abs().max().item() + 1e-5, f"grad {n}: {err:.3e} vs absmax {ref.abs().max().item():.3e}"
    err = (feats.grad.cpu() - of.grad).abs().max().item()
    assert err <= 1.5e-3 * of.grad.abs().max().item() + 1e-7, f"d tokens: {err:.3e}"
    # the FPN (trainable, adjacent) received gradients through the token view
    g = m.feature_pyramid.layer_blocks[1][0].weight.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().max() > 0


def test_graphed_step_with_prefetched_backbone_equals_the_sequential_step(a3d, dev):
    """GraphedStep(prefetch=model.backbone_maps): the frozen backbone of batch k + 1 runs on a side stream inside step k's graph and
    step k + 1 reads its maps.  Over a sequence of DIFFERENT batches: (i) the maps every step consumes are, bit for bit, the
    backbone's maps of THAT step's images (deterministic library solvers for this check), also when the two graph sets alternate
    and when the first launch passes other images than the capture saw; (ii) the losses equal those of the sequential GraphedStep
    on an identically initialised model fed the same sequence."""
    E = a3d.engine
    keep = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        def make():
            torch.manual_seed(0)
            m = a3d.Act3D(image_size=(128, 128), embedding_dim=60, num_attn_heads=4, gripper_loc_bounds=C.PERACT_BOUNDS,
                          num_ghost_points=128, num_ghost_points_val=128, num_sampling_level=2, sampler_seed=5).to(dev)
            m.backbone_dtype = m.fpn_dtype = torch.bfloat16
            return m.train()

        seq = [_sample(2, 2, 128, dev, 40 + i) for i in range(5)]
        crit = a3d.LossAndMetrics(position_loss="ce", rotation_parametrization="quat_from_query", ground_truth_gaussian_spread=0.01)
        mA, mB = make(), make()
        flatA, optA = E.get_optimizer(mA, lr=1e-4)
        flatB, optB = E.get_optimizer(mB, lr=1e-4)
        cap = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in seq[0].items()}      # static buffers of the two captures
        capB = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in seq[0].items()}
        gA = E.GraphedStep(lambda smp: E.fwd_bwd_keypose(mA, crit, smp), optA, cap, warmup=2)
        gB = E.GraphedStep(lambda smp: E.fwd_bwd_keypose(mB, crit, smp), optB, capB, warmup=2, prefetch=mB.backbone_maps)
        assert len(gB.g_fb) == 2 and len(gA.g_fb) == 1
        with torch.no_grad():                       # same state before the compared sequence (captures / warm-ups drew the same numbers)
            flatB.flat.copy_(flatA.flat)
            optB.exp_avg.copy_(optA.exp_avg)
            optB.exp_avg_sq.copy_(optA.exp_avg_sq)
        assert torch.equal(optA.step_count, optB.step_count) and torch.equal(mA._rng_state, mB._rng_state)
        order = [1, 2, 3, 4, 0, 2]                  # the first launch passes other images than the capture saw
        for i, k in enumerate(order):
            nxt = seq[order[i + 1]]["rgbs"] if i + 1 < len(order) else None
            lA = gA(seq[k]).clone()
            used = gB.maps[gB.parity]               # the set this launch reads (launch() flips the parity afterwards)
            lB = gB(seq[k], next_rgbs=nxt).clone()
            torch.cuda.synchronize()
            with torch.no_grad():
                ref = mA.backbone_maps(seq[k]["rgbs"])
            for name in ref:
                assert torch.equal(used[name], ref[name]), (i, k, name)
            print(f"[parity] prefetched-backbone step {i} (batch {k}): loss {lB.item():.6f} vs sequential {lA.item():.6f}")
            assert abs(lA.item() - lB.item()) <= (2e-5 if i == 0 else 2e-3) * max(1.0, abs(lA.item())), (i, lA.item(), lB.item())
        # the maps the LAST launch prefetched are those of its next_rgbs default (the same static images again)
        torch.cuda.synchronize()
        with torch.no_grad():
            ref = mA.backbone_maps(seq[order[-1]]["rgbs"])
        assert all(torch.equal(gB.maps[gB.parity][n], ref[n]) for n in ref)
    finally:
        torch.backends.cudnn.deterministic = keep


def test_graphed_step_equals_eager_train_one_step(a3d, dev):
    """GraphedStep (capture of zero_grad + forward + loss + backward + AdamW, what bench.py replays) against the eager
    engine.train_one_step on an identically initialised model: same losses and parameters after every step.  The device
    Philox sampler state, the AdamW step counter and the BatchNorm running statistics all advance inside the graph."""
    E = a3d.engine

    def make():
        torch.manual_seed(0)
        m = a3d.Act3D(image_size=(128, 128), embedding_dim=60, num_attn_heads=4, gripper_loc_bounds=C.PERACT_BOUNDS,
                      num_ghost_points=128, num_ghost_points_val=128, num_sampling_level=2, sampler_seed=5).to(dev)
        return m.train()

    s = _sample(2, 2, 128, dev, 33)
    crit = a3d.LossAndMetrics(position_loss="ce", rotation_parametrization="quat_from_query", ground_truth_gaussian_spread=0.01)
    mA, mB = make(), make()
    flatA, optA = E.get_optimizer(mA, lr=1e-4)
    flatB, optB = E.get_optimizer(mB, lr=1e-4)
    assert torch.equal(flatA.flat, flatB.flat)
    warm, replays = 2, 3
    for i in range(warm):
        E.train_one_step(mA, crit, optA, i, s)

    def fwd_bwd(sample):
        out = mB(sample["rgbs"], sample["pcds"], sample["instr"], sample["curr_gripper"], gt_action=sample["action"])
        loss = sum(crit.compute_loss(out, sample).values())
        loss.backward()
        return loss.detach()

    graphed = E.GraphedStep(fwd_bwd, optB, s, warmup=warm)          # `warm` eager steps, then the capture
    assert torch.equal(optA.step_count, optB.step_count) and torch.equal(mA._rng_state, mB._rng_state)
    # Two runs of the same step differ by accumulation-order noise (float atomics in the small-M weight gradients), and
    # AdamW turns that noise into +-lr steps on elements whose gradient is mathematically zero (e.g. the last ghost
    # LayerNorm's bias: q * sum_n (softmax - label)_n = 0), after which the two models drift apart like any two runs.
    # So: put B in exactly A's state (in place -- the graph keeps its buffers), then compare ONE step tightly.
    with torch.no_grad():
        flatB.flat.copy_(flatA.flat)
        optB.exp_avg.copy_(optA.exp_avg)
        optB.exp_avg_sq.copy_(optA.exp_avg_sq)
        for (n, a), (_, b) in zip(mA.backbone.named_buffers(), mB.backbone.named_buffers()):
            b.copy_(a)
    lossA = E.train_one_step(mA, crit, optA, warm, s)
    lossB = graphed(s).clone()
    torch.cuda.synchronize()
    # (the convolutions of backbone / FPN are MIOpen's: replay and eager may run different, equally valid algorithms, so
    # "same state" still means fp32 re-association noise of ~3e-6 in the loss, not bit equality)
    print(f"[parity] graphed vs eager loss, one step from the same state: {lossB.item():.7f} vs {lossA.item():.7f}")
    assert abs(lossB.item() - lossA.item()) <= 1e-5 * max(1.0, abs(lossA.item())), (lossB.item(), lossA.item())
    gA, gB = flatA.grad, flatB.grad
    gscale = gA.abs().max().item()
    gdiff = (gA - gB).abs().max().item()
    print(f"[parity] graphed vs eager gradients, one step from the same state: max abs diff {gdiff:.3e} (scale {gscale:.3e})")
    assert gdiff <= 2e-4 * gscale, (gdiff, gscale)
    solid = gA.abs() > 1e-3 * gscale
    diff = (flatA.flat - flatB.flat).abs()
    print(f"[parity] graphed vs eager parameters after that step: max abs diff {diff[solid].max().item():.3e} on "
          f"{int(solid.sum())} of {solid.numel()} elements with a solid gradient; {diff.max().item():.3e} overall")
    assert diff[solid].max().item() <= 1e-6
    assert diff.max().item() <= 2.01e-4                              # at most opposite +-lr steps on the noise elements
    for (n, a), (_, b) in zip(mA.backbone.named_buffers(), mB.backbone.named_buffers()):
        assert torch.allclose(a.float(), b.float(), rtol=1e-5, atol=1e-6), n
    # further replays keep tracking the eager run (loosely: the models are now two independent runs)
    for i in range(1, replays):
        lossA = E.train_one_step(mA, crit, optA, warm + i, s)
        lossB = graphed(s).clone()
        assert abs(lossB.item() - lossA.item()) <= 2e-3 * max(1.0, abs(lossA.item())), (i, lossB.item(), lossA.item())
    torch.cuda.synchronize()
    assert torch.equal(optA.step_count, optB.step_count)
    assert torch.equal(mA._rng_state, mB._rng_state)


# ------------------------------------------------------------------------------------------------ data parallel, 2 ranks
def _dp_gpu_worker(rank, world, port, overlap, graphed, q):
    """Two ranks share the one device (gloo moves the device buffers; RCCL refuses two ranks per GPU).  Each rank runs the
    keypose step on ITS batch through FlatDataParallel; rank 0 additionally computes, without any collective, the gradients
    of both batches from the same initial state and their mean -- the gradient of the mean-of-means loss DDP defines."""
    import importlib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        a3d = importlib.import_module("act3d-chained-diffuser_amd")
        E = a3d.engine
        dev = torch.device("cuda:0")
        crit = a3d.LossAndMetrics(position_loss="ce", rotation_parametrization="quat_from_query", ground_truth_gaussian_spread=0.01)

        def make(seed):
            torch.manual_seed(seed)
            m = a3d.Act3D(image_size=(128, 128), embedding_dim=60, num_attn_heads=4, gripper_loc_bounds=C.PERACT_BOUNDS,
                          num_ghost_points=128, num_ghost_points_val=128, num_sampling_level=2, sampler_seed=5).to(dev)
            return m.train()

        batches = [_sample(2, 1, 128, dev, 50 + r) for r in range(world)]
        m = make(100 + rank)                                   # different weights per rank: the broadcast must fix that
        flat, opt = E.get_optimizer(m, lr=1e-4)
        ddp = E.FlatDataParallel(flat, overlap=overlap, model=m)
        ddp.broadcast_parameters()                             # trainable buffer + the (frozen, seed-dependent) backbone
        assert ddp.overlap == overlap
        p0 = flat.flat.clone()
        res = {"rank": rank}

        def fwd_bwd_tf(model, sample, cb=None):
            """engine.fwd_bwd_keypose with the k-NN centres teacher-forced to the ground truth: the gradients are then a
            smooth function of the features (no argmax cascade), so that two runs of the same batch agree to rounding
            whatever convolution algorithm MIOpen picked; the data-parallel machinery under test is untouched."""
            tokens = model.compute_visual_tokens(sample["rgbs"], maps=sample.get("backbone_maps"))

            def hot(leaves):
                out = model(sample["rgbs"], sample["pcds"], sample["instr"], sample["curr_gripper"], gt_action=sample["action"],
                            visual_features=leaves, teacher_positions=[sample["action"][:, :3].contiguous()] * 2)
                return sum(crit.compute_loss(out, sample).values())
            return E._split_backward(tokens, hot, cb)

        def settle(model, fl):
            """One throw-away fwd + bwd: MIOpen may run a convolution configuration with a different algorithm on its first
            call than on later ones, and the untrained model's near-tied mask logits turn such 1e-6 feature differences into
            a different argmax -> different k-NN context -> visibly different gradients (seen as an intermittent 5e-4-of-scale
            mismatch).  Train-mode BatchNorm does not read the running statistics the warm-up updates."""
            st = model._rng_state.clone()
            fwd_bwd_tf(model, batches[rank])
            fl.zero_grad()
            model._rng_state.copy_(st)

        settle(m, flat)
        if rank == 0:
            # reference: rank 0's post-broadcast weights are make(100); both batches, same sampler state, no collective
            ref = make(100)
            rflat, _ = E.get_optimizer(ref, lr=1e-4)
            assert torch.equal(rflat.flat, p0)
            settle(ref, rflat)
            grads = []
            for b in batches:
                ref._rng_state.copy_(m._rng_state)
                rflat.zero_grad()
                fwd_bwd_tf(ref, b)
                grads.append(rflat.grad.clone())
            g_ref = sum(grads) / world
        if graphed:
            def fwd_bwd(sample, cb=None):
                return fwd_bwd_tf(m, sample, cb)
            state = m._rng_state.clone()
            # graphed == "prefetch": the next batch's frozen backbone forked inside the first of the three graphs, joined where it ends
            step = E.GraphedStep(fwd_bwd, opt, batches[rank], ddp=ddp, warmup=1, prefetch=m.backbone_maps if graphed == "prefetch" else None)
            # the warm-up step moved the weights: restore the broadcast state (in place) and replay ONE step
            flat.flat.copy_(p0)
            opt.reset_state()
            m._rng_state.copy_(state)
            for (_, a), (_, b) in zip(make(100).backbone.named_buffers(), m.backbone.named_buffers()):
                b.copy_(a)
            step(batches[rank])
            torch.cuda.synchronize()
            g = flat.grad / world
        else:
            opt.zero_grad()
            ddp.arm(True)
            fwd_bwd_tf(m, batches[rank], ddp.hot_path_done)
            scale = ddp.sync_gradients()
            torch.cuda.synchronize()
            g = flat.grad * scale
        if rank == 0:
            sc = g_ref.abs().max().item()
            res.update(err=(g - g_ref).abs().max().item(), scale=sc, late=flat.late_range, n=flat.n,
                       err_late=(g - g_ref)[flat.late_range[0]:flat.late_range[1]].abs().max().item())
        gathered = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(gathered, g)
        res["same_on_all_ranks"] = all(torch.equal(gathered[0], t) for t in gathered)
        q.put(res)
        dist.destroy_process_group()
    except Exception as e:                                      # surface the failure instead of a queue timeout
        import traceback
        q.put({"rank": rank, "error": traceback.format_exc()[-1500:]})


@pytest.mark.parametrize("overlap,graphed", [(False, False), (True, False), (True, True), (True, "prefetch")])
def test_data_parallel_two_ranks_equals_mean_of_rank_gradients(dev, overlap, graphed):
    """FlatDataParallel on the device: averaged 2-rank gradients == the mean of the two per-batch gradients (DDP's
    mean-of-means), with the hot-path segments reduced early on the side stream (overlap) or in one piece, eagerly and
    through the three-graph GraphedStep.  Reference semantics: DistributedDataParallel at engine.py:121-124."""
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    for attempt in range(2):                       # one retry if the rendezvous itself fails (port race on a shared box)
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        q = ctx.Queue()
        procs = [ctx.Process(target=_dp_gpu_worker, args=(r, 2, port, overlap, graphed, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = [q.get(timeout=600) for _ in range(2)]
        for p in procs:
            p.join(timeout=60)
        errs = [r["error"] for r in res if "error" in r]
        if not errs or attempt == 1 or not any(k in e for e in errs for k in ("Address already in use", "Connection", "timed out")):
            break
        print("[dp test] retrying after a rendezvous failure:", errs[0][-300:])
    for r in res:
        assert "error" not in r, r["error"]
        assert r["same_on_all_ranks"]
    r0 = [r for r in res if r["rank"] == 0][0]
    print(f"[parity] DP overlap={overlap} graphed={graphed}: max grad err {r0['err']:.3e} (scale {r0['scale']:.3e}), "
          f"FPN segment {r0['late']} of {r0['n']}: {r0['err_late']:.3e}")
    assert r0["late"][1] > r0["late"][0], "the test must exercise the late (FPN) segment"
    assert r0["err"] <= 2e-4 * r0["scale"]
