"""BASELINE.json configs[3] on one GPU box: the TRAJECTORY model through the data-parallel step (engine.fwd_bwd_trajectory +
FlatDataParallel, reference main_trajectory.py:177-204 under engine.py:121-124), and the joint keypose + trajectory
iteration (engine.JointStep: two models, two optimizers).

  * 2 ranks on the one device (gloo moves the device buffers; RCCL refuses two ranks per GPU): the averaged gradients of the
    trajectory model -- dropout 0.1 with PER-RANK generator seeds, overlap on / off, eager and three-graph GraphedStep -- equal
    the mean of the per-rank gradients computed without any collective; the same for both models of a JointStep;
  * 1 process: one JointStep iteration == train_one_step followed by train_one_step_trajectory from the same state.
"""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import common as C  # noqa: E402

pytestmark = pytest.mark.gpu
IMG = 128


def _kp_sample(B, ncam, dev, seed):
    inp = C.keypose_inputs(seed, B, ncam, 60, 1, image=IMG)
    rs = np.random.RandomState(seed + 1)
    rgb = torch.from_numpy(rs.uniform(0, 1, size=(B, ncam, 3, IMG, IMG)).astype(np.float32))
    s = {"rgbs": rgb, "pcds": inp["pcd"], "instr": inp["instr"], "curr_gripper": inp["curr_gripper"], "action": inp["action"]}
    s = {k: v.to(dev) for k, v in s.items()}
    s["task"] = ["t"] * B
    return s


def _tr_sample(B, Ln, ncam, dev, seed):
    """A trajectory batch with the DDPM noise / timesteps carried along (engine.fwd_bwd_trajectory passes them through), so
    that a step is a deterministic function of (weights, batch, dropout generator state)."""
    g = torch.Generator().manual_seed(seed)
    lo, hi = torch.tensor(C.DIFFUSION_BOUNDS[0], dtype=torch.float32), torch.tensor(C.DIFFUSION_BOUNDS[1], dtype=torch.float32)
    rgb = torch.rand(B, ncam, 3, IMG, IMG, generator=g)
    pcd = lo.view(1, 1, 3, 1, 1) + torch.rand(B, ncam, 3, IMG, IMG, generator=g) * (hi - lo).view(1, 1, 3, 1, 1)

    def pose(n):
        q = torch.randn(*n, 4, generator=g)
        return torch.cat([lo + 0.15 * (hi - lo) + torch.rand(*n, 3, generator=g) * 0.7 * (hi - lo), q / q.norm(dim=-1, keepdim=True)], -1)

    cg, gg = pose((B,)), pose((B,))
    w = torch.linspace(0, 1, Ln).view(1, Ln, 1)
    traj = cg[:, None] * (1 - w) + gg[:, None] * w + 0.01 * torch.randn(B, Ln, 7, generator=g)
    traj[..., 3:] = traj[..., 3:] / traj[..., 3:].norm(dim=-1, keepdim=True)
    s = {"rgbs": rgb, "pcds": pcd, "curr_gripper": cg, "action": gg, "trajectory": traj, "instr": torch.randn(B, 53, 512, generator=g),
         "trajectory_mask": torch.zeros(B, Ln, dtype=torch.bool), "noise": torch.randn(B, Ln, 9, generator=g),
         "timesteps": torch.randint(0, 100, (B,), generator=g)}
    return {k: v.to(dev) for k, v in s.items()}


def _make_keypose(a3d, dev, seed):
    torch.manual_seed(seed)
    m = a3d.Act3D(image_size=(IMG, IMG), embedding_dim=60, num_attn_heads=4, gripper_loc_bounds=C.PERACT_BOUNDS,
                  num_ghost_points=128, num_ghost_points_val=128, num_sampling_level=2, sampler_seed=5).to(dev)
    return m.train()


def _make_planner(a3d, dev, seed, dropout_seed):
    torch.manual_seed(seed)
    m = a3d.DiffusionPlanner(image_size=(IMG, IMG), embedding_dim=120, output_dim=7, num_vis_ins_attn_layers=1,
                             num_query_cross_attn_layers=2, use_instruction=True, use_goal=True, use_goal_at_test=True,
                             weight_tying=True, gripper_loc_bounds=C.DIFFUSION_BOUNDS, rotation_parametrization="6D",
                             diffusion_timesteps=100, dropout=0.1, dropout_seed=dropout_seed).to(dev)
    for mod in m.modules():                      # AdaLN is zero-initialised in the reference; give it non-trivial weights
        if isinstance(mod, a3d.nn.AdaLN):
            torch.nn.init.normal_(mod.modulation[1].weight, std=0.02)
    return m.train()


def _set_drop(planner, seed, counter=0):
    planner.prediction_head._drop_state.copy_(torch.tensor([seed, counter], dtype=torch.int64))


def _kp_fwd_bwd_tf(E, model, crit, sample, cb=None):
    """engine.fwd_bwd_keypose with the k-NN centres teacher-forced (see tests/test_engine_gpu.py: the argmax cascade of an
    untrained model turns 1e-6 feature noise into a different context, which is not what these tests are about)."""
    tokens = model.compute_visual_tokens(sample["rgbs"])

    def hot(leaves):
        out = model(sample["rgbs"], sample["pcds"], sample["instr"], sample["curr_gripper"], gt_action=sample["action"],
                    visual_features=leaves, teacher_positions=[sample["action"][:, :3].contiguous()] * 2)
        return sum(crit.compute_loss(out, sample).values())
    return E._split_backward(tokens, hot, cb)


# ------------------------------------------------------------------------------------------------ 2 ranks on one device
def _worker(rank, world, port, mode, q):
    import importlib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        a3d = importlib.import_module("act3d-chained-diffuser_amd")
        E = a3d.engine
        dev = torch.device("cuda:0")
        overlap = mode != "traj_blocking"
        tcrit = a3d.TrajectoryCriterion()
        tr_batches = [_tr_sample(2, 8, 1, dev, 70 + r) for r in range(world)]
        seeds = [900 + 13 * r for r in range(world)]                     # per-rank dropout generator seeds
        tr = _make_planner(a3d, dev, 200 + rank, seeds[rank])            # different weights per rank: the broadcast fixes that
        tflat, topt = E.get_optimizer(tr, lr=1e-4)
        tddp = E.FlatDataParallel(tflat, overlap=overlap, model=tr)
        tddp.broadcast_parameters()
        joint = mode == "joint"
        res = {"rank": rank}
        if joint:
            kcrit = a3d.LossAndMetrics(position_loss="ce", rotation_parametrization="quat_from_query", ground_truth_gaussian_spread=0.01)
            kp_batches = [_kp_sample(2, 1, dev, 50 + r) for r in range(world)]
            kp = _make_keypose(a3d, dev, 100 + rank)
            kflat, kopt = E.get_optimizer(kp, lr=1e-4)
            kddp = E.FlatDataParallel(kflat, overlap=True, model=kp)
            kddp.broadcast_parameters()

        def settle_tr(model, fl, seed):
            E.fwd_bwd_trajectory(model, tcrit, tr_batches[rank])          # MIOpen picks its algorithms on the first call
            fl.zero_grad()
            _set_drop(model, seed)

        settle_tr(tr, tflat, seeds[rank])
        if joint:
            st = kp._rng_state.clone()
            _kp_fwd_bwd_tf(E, kp, kcrit, kp_batches[rank])
            kflat.zero_grad()
            kp._rng_state.copy_(st)

        if rank == 0:
            # reference: rank 0's post-broadcast weights; every rank's batch with that rank's dropout seed, no collective
            ref = _make_planner(a3d, dev, 200, seeds[0])
            rflat, _ = E.get_optimizer(ref, lr=1e-4)
            assert torch.equal(rflat.flat, tflat.flat)
            settle_tr(ref, rflat, seeds[0])
            grads = []
            for r, b in enumerate(tr_batches):
                _set_drop(ref, seeds[r])
                rflat.zero_grad()
                E.fwd_bwd_trajectory(ref, tcrit, b)
                grads.append(rflat.grad.clone())
            g_ref = sum(grads) / world
            assert (grads[0] - grads[1]).abs().max() > 1e-3 * g_ref.abs().max(), "the two ranks' gradients must differ"
            if joint:
                kref = _make_keypose(a3d, dev, 100)
                krflat, _ = E.get_optimizer(kref, lr=1e-4)
                st = kref._rng_state.clone()
                _kp_fwd_bwd_tf(E, kref, kcrit, kp_batches[0])
                kgrads = []
                for b in kp_batches:
                    kref._rng_state.copy_(kp._rng_state)
                    krflat.zero_grad()
                    _kp_fwd_bwd_tf(E, kref, kcrit, b)
                    kgrads.append(krflat.grad.clone())
                kg_ref = sum(kgrads) / world

        p0 = tflat.flat.clone()
        if mode == "traj_graphed":
            def fwd_bwd(sample, cb=None):
                return E.fwd_bwd_trajectory(tr, tcrit, sample, cb)
            step = E.GraphedStep(fwd_bwd, topt, tr_batches[rank], ddp=tddp, warmup=1)
            # the warm-up step moved the weights: restore the broadcast state in place and replay ONE step
            tflat.flat.copy_(p0)
            topt.reset_state()
            _set_drop(tr, seeds[rank])
            for (_, a), (_, b) in zip(_make_planner(a3d, dev, 200, 0).prediction_head.backbone.named_buffers(),
                                      tr.prediction_head.backbone.named_buffers()):
                b.copy_(a)
            step(tr_batches[rank])
            torch.cuda.synchronize()
            g = tflat.grad / world
        elif joint:
            # JointStep with the keypose forward teacher-forced: same object, the keypose fwd + bwd swapped for the smooth one
            js = E.JointStep(kp, kcrit, kopt, tr, tcrit, topt, kp_ddp=kddp, tr_ddp=tddp)
            orig = E.fwd_bwd_keypose
            E.fwd_bwd_keypose = lambda model, crit, sample, use_gt, cb: _kp_fwd_bwd_tf(E, model, crit, sample, cb)
            try:
                js(kp_batches[rank], tr_batches[rank])
            finally:
                E.fwd_bwd_keypose = orig
            torch.cuda.synchronize()
            g = tflat.grad / world
            kg = kflat.grad / world
        else:
            topt.zero_grad()
            tddp.arm(True)
            E.fwd_bwd_trajectory(tr, tcrit, tr_batches[rank], tddp.hot_path_done)
            scale = tddp.sync_gradients()
            torch.cuda.synchronize()
            g = tflat.grad * scale
        if rank == 0:
            sc = g_ref.abs().max().item()
            res.update(err=(g - g_ref).abs().max().item(), scale=sc, late=tflat.late_range, n=tflat.n)
            if joint:
                res.update(kerr=(kg - kg_ref).abs().max().item(), kscale=kg_ref.abs().max().item())
        gathered = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(gathered, g)
        res["same_on_all_ranks"] = all(torch.equal(gathered[0], t) for t in gathered)
        res["drop_state"] = tr.prediction_head._drop_state.cpu().tolist()
        q.put(res)
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put({"rank": rank, "error": traceback.format_exc()[-2500:]})


@pytest.mark.parametrize("mode", ["traj_blocking", "traj_overlap", "traj_graphed", "joint"])
def test_trajectory_and_joint_data_parallel_two_ranks(dev, mode):
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    for attempt in range(2):
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = [q.get(timeout=900) for _ in range(2)]
        for p in procs:
            p.join(timeout=60)
        errs = [r["error"] for r in res if "error" in r]
        if not errs or attempt == 1 or not any(k in e for e in errs for k in ("Address already in use", "Connection", "timed out")):
            break
    for r in res:
        assert "error" not in r, r["error"]
        assert r["same_on_all_ranks"]
    r0 = [r for r in res if r["rank"] == 0][0]
    print(f"[parity] DP {mode}: trajectory max grad err {r0['err']:.3e} (scale {r0['scale']:.3e}); FPN segment {r0['late']} of {r0['n']}"
          + (f"; keypose {r0['kerr']:.3e} (scale {r0['kscale']:.3e})" if mode == "joint" else ""))
    assert r0["late"][1] > r0["late"][0], "the test must exercise the late (FPN) segment"
    assert r0["err"] <= 2e-4 * r0["scale"]
    if mode == "joint":
        assert r0["kerr"] <= 2e-4 * r0["kscale"]
    # every rank drew its masks from ITS generator and advanced it once for the measured step
    assert sorted(r["drop_state"][0] for r in res) == [900, 913]
    assert all(r["drop_state"][1] == 1 for r in res)


# ------------------------------------------------------------------------------------------------ joint == separate
def test_joint_iteration_equals_the_two_steps_run_separately(a3d, dev):
    """engine.JointStep (keypose fwd/bwd -> trajectory fwd/bwd -> both AdamW steps) against engine.train_one_step followed by
    engine.train_one_step_trajectory on identically initialised models: same losses, gradients and updated parameters."""
    E = a3d.engine
    kcrit = a3d.LossAndMetrics(position_loss="ce", rotation_parametrization="quat_from_query", ground_truth_gaussian_spread=0.01)
    tcrit = a3d.TrajectoryCriterion()
    ks, ts = _kp_sample(2, 2, dev, 33), _tr_sample(2, 8, 1, dev, 44)
    kA, kB = _make_keypose(a3d, dev, 0), _make_keypose(a3d, dev, 0)
    tA, tB = _make_planner(a3d, dev, 1, 77), _make_planner(a3d, dev, 1, 77)
    (kfA, koA), (kfB, koB) = E.get_optimizer(kA, lr=1e-4), E.get_optimizer(kB, lr=1e-4)
    (tfA, toA), (tfB, toB) = E.get_optimizer(tA, lr=1e-4), E.get_optimizer(tB, lr=1e-4)
    assert torch.equal(kfA.flat, kfB.flat) and torch.equal(tfA.flat, tfB.flat)
    joint = E.JointStep(kA, kcrit, koA, tA, tcrit, toA)
    # both paths call engine.fwd_bwd_keypose; teacher-force its k-NN centres (the untrained model's argmax cascade turns
    # accumulation-order noise into a different context -- not what this test is about)
    orig = E.fwd_bwd_keypose
    E.fwd_bwd_keypose = lambda model, crit, sample, use_gt=True, cb=None: _kp_fwd_bwd_tf(E, model, crit, sample, cb)
    try:
        _joint_vs_separate(E, joint, ks, ts, kcrit, tcrit, kA, kB, tA, tB, kfA, kfB, tfA, tfB, koA, koB, toA, toB)
    finally:
        E.fwd_bwd_keypose = orig


def _joint_vs_separate(E, joint, ks, ts, kcrit, tcrit, kA, kB, tA, tB, kfA, kfB, tfA, tfB, koA, koB, toA, toB):
    # two settling iterations on both sides (MIOpen may change a configuration's algorithm between its first calls: seen once as
    # a 1e-2 bf16-accumulation-order difference of the FPN weight gradients in a full-suite run), then both into the same state
    for _ in range(2):
        joint(ks, ts)
        E.train_one_step(kB, kcrit, koB, 0, ks)
        E.train_one_step_trajectory(tB, tcrit, toB, 0, ts)
    with torch.no_grad():
        for fa, fb, oa, ob in ((kfA, kfB, koA, koB), (tfA, tfB, toA, toB)):
            fb.flat.copy_(fa.flat)
            ob.exp_avg.copy_(oa.exp_avg)
            ob.exp_avg_sq.copy_(oa.exp_avg_sq)
            ob.step_count.copy_(oa.step_count); ob.seg_state.copy_(oa.seg_state)
        for (_, a), (_, b) in zip(kA.backbone.named_buffers(), kB.backbone.named_buffers()):
            b.copy_(a)
        for (_, a), (_, b) in zip(tA.prediction_head.backbone.named_buffers(), tB.prediction_head.backbone.named_buffers()):
            b.copy_(a)
        kB._rng_state.copy_(kA._rng_state)
        tB.prediction_head._drop_state.copy_(tA.prediction_head._drop_state)
    lkA, ltA = joint(ks, ts)
    lkB = E.train_one_step(kB, kcrit, koB, 1, ks)
    ltB = E.train_one_step_trajectory(tB, tcrit, toB, 1, ts)
    torch.cuda.synchronize()
    print(f"[parity] joint vs separate losses: keypose {lkA.item():.7f} / {lkB.item():.7f}, trajectory {ltA.item():.7f} / {ltB.item():.7f}")
    assert abs(lkA.item() - lkB.item()) <= 1e-5 * max(1.0, abs(lkB.item()))
    assert abs(ltA.item() - ltB.item()) <= 1e-5 * max(1.0, abs(ltB.item()))
    for name, fa, fb in (("keypose", kfA, kfB), ("trajectory", tfA, tfB)):
        gs = fb.grad.abs().max().item()
        gd = (fa.grad - fb.grad).abs().max().item()
        solid = fb.grad.abs() > 1e-3 * gs
        pd = (fa.flat - fb.flat).abs()
        print(f"[parity] joint vs separate {name}: grad diff {gd:.3e} (scale {gs:.3e}), parameter diff {pd[solid].max().item():.3e} on "
              f"solid-gradient elements, {pd.max().item():.3e} overall")
        worst = sorted(((fa.grad[a:b] - fb.grad[a:b]).abs().max().item(), n) for n, (a, b) in fb.slices.items())[-3:]
        print(f"[parity] joint vs separate {name}: largest per-parameter gradient differences " + ", ".join(f"{n} {d:.2e}" for d, n in worst))
        assert gd <= 2e-4 * gs
        assert pd[solid].max().item() <= 1e-6 and pd.max().item() <= 2.01e-4
    assert torch.equal(koA.step_count, koB.step_count) and torch.equal(toA.step_count, toB.step_count)
    assert torch.equal(tA.prediction_head._drop_state, tB.prediction_head._drop_state)


# ------------------------------------------------------------------------------------------------ graph replays of the trajectory step
@pytest.mark.parametrize("prefetch", [False, True])
def test_graphed_trajectory_step_gradients_equal_eager_on_every_replay(a3d, dev, prefetch):
    """The captured diffusion training step (engine.GraphedStep over fwd_bwd_trajectory, what bench.py replays) against the eager
    step, over FOUR replays from the same weights / batch / dropout state (learning rate 0, so every replay must reproduce the
    same gradients).  Horizon 40 puts a3d_adaln_bwd on its split path (per-split partial sums accumulated into a zeroed dmod):
    that zeroing was a captured hipMemsetAsync, which this stack replays right once and stale afterwards (round-5 advisor
    finding) -- it is a kernel node now, and this test is what would have caught it.
    prefetch: the next batch's frozen backbone forked inside the step's graph (two alternating graph / map sets; round 6)."""
    E = a3d.engine
    tcrit = a3d.TrajectoryCriterion()
    ts = _tr_sample(3, 40, 1, dev, 91)
    tr = _make_planner(a3d, dev, 5, 321)
    flat, opt = E.get_optimizer(tr, lr=0.0)

    def eager():
        _set_drop(tr, 321, 0)
        opt.zero_grad()
        loss = E.fwd_bwd_trajectory(tr, tcrit, ts)
        torch.cuda.synchronize()
        return loss.item(), flat.grad.clone()

    l_ref, g_ref = eager()
    l_again, g_again = eager()                                       # run-to-run noise of the eager step itself (float atomics)
    gs = g_ref.abs().max().item()
    noise = (g_again - g_ref).abs().max().item()
    graphed = E.GraphedStep(lambda s: E.fwd_bwd_trajectory(tr, tcrit, s), opt, ts, warmup=2,
                            prefetch=tr.prediction_head.backbone_maps if prefetch else None)
    assert len(graphed.g_fb) == (2 if prefetch else 1)
    mods = [n for n in flat.slices if "adaln" in n and "modulation" in n]
    assert mods
    for rep in range(4):
        _set_drop(tr, 321, 0)
        loss = graphed(ts).clone()
        torch.cuda.synchronize()
        g = flat.grad
        gd = (g - g_ref).abs().max().item()
        md = max((g[a:b] - g_ref[a:b]).abs().max().item() for a, b in (flat.slices[n] for n in mods))
        print(f"[parity] graphed trajectory step, replay {rep}: loss {loss.item():.7f} vs eager {l_ref:.7f}; gradient diff {gd:.3e} "
              f"(AdaLN modulation weights {md:.3e}; scale {gs:.3e}; eager run-to-run {noise:.3e})")
        assert abs(loss.item() - l_ref) <= 1e-5 * max(1.0, abs(l_ref))
        assert gd <= max(2e-4 * gs, 4 * noise), (rep, gd, gs)
