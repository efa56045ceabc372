"""Training-step harness: flat parameter / gradient buffers, fused AdamW, data-parallel gradient all-reduce over
RCCL, checkpoint format and whole-step hipGraph capture.

Mirrors the hot-path-relevant surface of the reference's engine.BaseTrainTester (engine.py:18-230) and
TrainTester.train_one_step (main_keypose.py:207-234, main_trajectory.py:177-204):
  get_optimizer   -> FlatAdamW with the reference's two parameter groups (names containing "bias" -> no decay)
  DDP wrap        -> FlatDataParallel: one flat fp32 gradient buffer, all-reduce(SUM) then 1/world folded into AdamW
  save/load_checkpoint -> {"weight" (module.-prefixed), "optimizer", "iter", "best_loss"}
The reference's CLI, data loaders, tensorboard and evaluation loop are out of scope.
"""
import os

import torch
import torch.distributed as dist

from . import lib as L

NO_DECAY = ["bias", "LayerNorm.weight", "LayerNorm.bias"]       # engine.py:95 (the LayerNorm patterns never match)


def _is_no_decay(name):
    return any(nd in name for nd in NO_DECAY)


def discover_active_parameters(model, run_fwd_bwd):
    """Names of the parameters that receive a gradient in one forward/backward (the reference relies on
    find_unused_parameters=True and AdamW skipping grad=None parameters, engine.py:121-124)."""
    for p in model.parameters():
        p.grad = None
    run_fwd_bwd()
    active = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is not None]
    for p in model.parameters():
        p.grad = None
    return active


class FlatParams:
    """Re-homes the active parameters into ONE flat fp32 buffer and their .grad into one flat gradient buffer.

    Layout: [ no-decay | decay ]; inside each, parameters of `late_prefixes` (the FPN, whose gradients are produced
    last in backward) are placed at the inner edge so that they form one contiguous middle segment:
        [ hot no-decay | late no-decay | late decay | hot decay ]
    which lets the hot-path segments be all-reduced while the FPN backward is still running."""

    def __init__(self, model, active_names=None, late_prefixes=("feature_pyramid",)):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if active_names is not None:
            keep = set(active_names)
            named = [(n, p) for n, p in named if n in keep]
        is_late = lambda n: any(pre in n for pre in late_prefixes)
        segs = [[], [], [], []]
        for n, p in named:
            nd, late = _is_no_decay(n), is_late(n)
            segs[0 if (nd and not late) else 1 if (nd and late) else 2 if late else 3].append((n, p))
        self.order = [x for s in segs for x in s]
        # torch.optim.AdamW indices of the reference's optimizer (engine.py:89-102): every named parameter, the "bias"
        # group first, each group in named_parameters() order -- used to exchange optimizer state with reference checkpoints
        every = [n for n, _ in model.named_parameters()]
        self.torch_groups = [[n for n in every if _is_no_decay(n)], [n for n in every if not _is_no_decay(n)]]
        self.torch_index = {n: i for i, n in enumerate(self.torch_groups[0] + self.torch_groups[1])}
        sizes = [sum(p.numel() for _, p in s) for s in segs]
        self.n = sum(sizes)
        self.n_nodecay = sizes[0] + sizes[1]
        self.late_range = (sizes[0], sizes[0] + sizes[1] + sizes[2])
        dev = self.order[0][1].device
        self.flat = torch.empty(self.n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(self.n, device=dev, dtype=torch.float32)
        self.slices = {}
        off = 0
        with torch.no_grad():
            for n, p in self.order:
                k = p.numel()
                self.flat[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + k].view(p.shape)
                p.grad = self.grad[off:off + k].view(p.shape)
                self.slices[n] = (off, off + k)
                off += k

    def rebind_grads(self):
        """Re-attach .grad views (after something set p.grad = None)."""
        for n, p in self.order:
            a, b = self.slices[n]
            p.grad = self.grad[a:b].view(p.shape)

    def zero_grad(self):
        self.grad.zero_()
        for n, p in self.order:
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + self.slices[n][0] * 4:
                self.rebind_grads()
                break


class FlatAdamW:
    """torch.optim.AdamW semantics (lr, betas, eps, two weight-decay groups) as one fused kernel over FlatParams."""

    def __init__(self, flat, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4):
        self.flat, self.betas, self.eps, self.weight_decay = flat, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        self.step_count = torch.zeros(1, device=flat.flat.device, dtype=torch.float32)     # optimizer.step() calls so far
        # per-parameter table of the fused kernel: element offsets of the parameters in buffer order, and per parameter
        # {step (torch's state[p]["step"]), active, lr / bc1, sqrt(bc2)}.  A parameter is updated from the first step in which
        # its gradient segment holds a non-zero element (torch: .grad is not None) and then every step, all elements
        self.seg_off = torch.tensor([flat.slices[n][0] for n, _ in flat.order] + [flat.n], dtype=torch.int64, device=flat.flat.device)
        self.seg_state = torch.zeros((len(flat.order), 4), device=flat.flat.device, dtype=torch.float32)
        self.param_groups = [{"lr": lr, "weight_decay": 0.0}, {"lr": lr, "weight_decay": weight_decay}]

    @property
    def lr(self):
        """The learning rate the fused kernel uses: ONE value for both groups (the reference builds both groups with
        args.lr and resets every group to it on resume, engine.py:89-102,200-201)."""
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, value):
        for g in self.param_groups:
            g["lr"] = float(value)

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()

    def step(self, grad_scale=1.0):
        f = self.flat
        L.call("a3d_adamw_step", f.flat.data_ptr(), f.grad.data_ptr(), self.exp_avg.data_ptr(),
               self.exp_avg_sq.data_ptr(), self.step_count.data_ptr(), self.seg_off.data_ptr(), self.seg_state.data_ptr(),
               len(f.order), f.n, f.n_nodecay, float(self.lr),
               self.betas[0], self.betas[1], self.eps, 0.0, float(self.weight_decay), float(grad_scale), L.stream())

    @property
    def param_steps(self):
        """Per-parameter step counts (buffer order): torch.optim.AdamW's state[p]["step"]; 0 = never had a gradient."""
        return self.seg_state[:, 0]

    def reset_state(self):
        """Fresh-optimizer state: zero moments and step counts (parameters re-join with their first gradient)."""
        self.exp_avg.zero_(); self.exp_avg_sq.zero_(); self.step_count.zero_(); self.seg_state.zero_()

    def state_dict(self):
        """torch.optim.AdamW's state_dict layout for the reference's optimizer (engine.py:89-102: two groups over ALL named
        parameters, int-indexed state with per-parameter `step`), so that the reference's `optimizer.load_state_dict`
        (engine.py:199) accepts a checkpoint written here and vice versa.  Parameters outside the flat buffer (frozen
        backbone, never-used FPN blocks) have no state, exactly like parameters whose .grad stayed None in torch."""
        f = self.flat
        state = {}
        steps = self.param_steps.detach().cpu()
        for (n, p), st in zip(f.order, steps):
            if float(st) <= 0:
                continue               # never received a gradient: torch writes no state for it
            a, b = f.slices[n]
            state[f.torch_index[n]] = {"step": st.clone().reshape(()), "exp_avg": self.exp_avg[a:b].detach().clone().view(p.shape),
                                       "exp_avg_sq": self.exp_avg_sq[a:b].detach().clone().view(p.shape)}
        groups, off = [], 0
        for names, wd in zip(f.torch_groups, (0.0, self.weight_decay)):
            groups.append({"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": wd,
                           "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                           "differentiable": False, "fused": None, "decoupled_weight_decay": True,
                           "params": list(range(off, off + len(names)))})
            off += len(names)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        """Accepts the torch.optim.AdamW layout (a reference checkpoint's "optimizer" entry).  Raises on a layout that does
        not describe this model's parameters (unknown indices, size mismatches, different group sizes).  Parameters WITHOUT
        an entry are legal: torch.optim.AdamW writes no state for parameters whose .grad stayed None (the FPN blocks of maps
        a configuration never reads, find_unused_parameters=True in engine.py:121-124); their moments and step count stay
        zero and, as in torch, they start at step 1 with the first gradient they receive.  Per-parameter step counts are
        kept as stored (they may differ between parameters)."""
        f = self.flat
        if not isinstance(sd, dict) or "state" not in sd or "param_groups" not in sd:
            raise ValueError("optimizer state: expected torch.optim.AdamW's {'state', 'param_groups'} layout")
        sizes = [len(g["params"]) for g in sd["param_groups"]]
        if sizes != [len(g) for g in f.torch_groups]:
            raise ValueError("optimizer state: parameter groups of sizes %s do not match this model's %s "
                             "(reference grouping, engine.py:89-102)" % (sizes, [len(g) for g in f.torch_groups]))
        by_index = {i: n for n, i in f.torch_index.items()}
        seg_of = {n: i for i, (n, _) in enumerate(f.order)}
        steps = torch.zeros(len(f.order), dtype=torch.float32)
        loaded = []
        for idx, st in sd["state"].items():
            n = by_index.get(int(idx))
            if n is None:
                raise ValueError("optimizer state: index %r is not a parameter of this model" % (idx,))
            if n not in f.slices:
                continue                       # state of a parameter that is not trained here (no gradient path)
            a, b = f.slices[n]
            if st["exp_avg"].numel() != b - a:
                raise ValueError("optimizer state of %s has %d elements, the parameter %d" % (n, st["exp_avg"].numel(), b - a))
            loaded.append((a, b, st))
            steps[seg_of[n]] = float(st["step"])
        self.reset_state()
        for a, b, st in loaded:
            self.exp_avg[a:b].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[a:b].copy_(st["exp_avg_sq"].reshape(-1))
        self.seg_state[:, 0].copy_(steps)
        self.step_count.fill_(float(steps.max()) if len(loaded) else 0.0)
        self.lr = sd["param_groups"][0].get("lr", self.lr)


def get_optimizer(model, lr=1e-4, active_names=None):
    """engine.py:89-102 on the flat buffers.  Returns (FlatParams, FlatAdamW).  With active_names=None every trainable
    parameter is placed in the buffer; parameters that never receive a gradient are left untouched by the fused AdamW kernel
    (as torch.optim.AdamW skips parameters whose .grad is None -- no weight decay on unused FPN blocks)."""
    flat = FlatParams(model, active_names)
    return flat, FlatAdamW(flat, lr=lr)


DP_ONESHOT = os.environ.get("A3D_DP_ONESHOT", "0") == "1"


class FlatDataParallel:
    """Data parallelism over one flat gradient buffer (replaces DistributedDataParallel, engine.py:121-124).

    broadcast_parameters(): rank 0 -> all (DDP's initial broadcast).
    sync_gradients(): all-reduce(SUM); the 1/world_size averaging is returned as the grad_scale for FlatAdamW.step.
    With `overlap=True` the hot-path segments [hot no-decay] and [hot decay] of the buffer are reduced on a side stream as
    soon as `hot_path_done()` is called -- by the split backward of `fwd_bwd_keypose` / `fwd_bwd_trajectory`, the moment
    the gradients of the FPN's output tokens exist and before the FPN / convolution backward is enqueued -- while that
    backward runs on the main stream; the FPN segment follows in sync_gradients().  `arm(False)` disables the early
    reduction for the non-final micro-batches of a gradient-accumulation window.
    Works with backend "nccl" (= RCCL on ROCm) and with "gloo" (device or CPU buffers; tests)."""

    def __init__(self, flat, process_group=None, overlap=True, model=None):
        self.flat = flat
        self.model = model
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # A3D_DP_ONESHOT=1: ONE all-reduce of the whole flat buffer after the backward instead of the three overlapped messages
        # (hot no-decay | hot decay early on the side stream, FPN segment late) -- the A/B switch for the first multi-GPU run
        self.overlap = overlap and flat.flat.is_cuda and not DP_ONESHOT
        self._pending = []
        self._side = torch.cuda.Stream() if self.overlap else None
        self._early_done = False
        self._begun = False
        self._armed = True

    def broadcast_parameters(self, src=0):
        """rank `src`'s state to every rank, as DistributedDataParallel does at construction: the flat (trainable) buffer
        and, when the model was given, every other state_dict entry -- the frozen backbone's weights and its BatchNorm
        statistics.  (Per-rank generator states -- ghost sampler, dropout -- are non-persistent buffers and stay per rank.)"""
        if self.world == 1:
            return
        dist.broadcast(self.flat.flat, src=src, group=self.pg)
        if self.model is not None:
            flat_ptrs = {p.data_ptr() for _, p in self.flat.order}
            for name, t in self.model.state_dict().items():
                if t.data_ptr() in flat_ptrs:
                    continue
                if t.is_contiguous():
                    dist.broadcast(t, src=src, group=self.pg)
                else:                                   # channels-last convolution weights of the bf16 backbone
                    c = t.contiguous()
                    dist.broadcast(c, src=src, group=self.pg)
                    t.copy_(c)

    def _segments(self):
        a, b = self.flat.late_range
        return [(0, a), (b, self.flat.n)], (a, b)

    def arm(self, on=True):
        self._armed = bool(on)

    def hot_path_done(self):
        """Call when every hot-path gradient has been produced (FPN backward not yet enqueued)."""
        if self.world == 1 or not self.overlap or not self._armed or self._early_done:
            return
        hot, _ = self._segments()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._side.wait_event(ev)
        with torch.cuda.stream(self._side):
            for a, b in hot:
                if b > a:
                    self._pending.append(dist.all_reduce(self.flat.grad[a:b], op=dist.ReduceOp.SUM, group=self.pg,
                                                         async_op=True))
        self._early_done = True

    def begin_sync(self):
        """Enqueue the all-reduce of everything that has not been reduced yet WITHOUT waiting for it: on the side stream
        (device buffers with overlap) so that whatever the caller runs next on the main stream -- the other model's step of a
        joint iteration (JointStep) -- hides it.  finish_sync() joins.  Without a side stream this is the plain blocking
        all-reduce."""
        if self.world == 1 or self._begun:
            return
        hot, late = self._segments()
        todo = [late] if self._early_done else [(0, self.flat.n)]
        if self._side is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._side.wait_event(ev)
            with torch.cuda.stream(self._side):
                for a, b in todo:
                    if b > a:
                        self._pending.append(dist.all_reduce(self.flat.grad[a:b], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        else:
            for a, b in todo:
                if b > a:
                    dist.all_reduce(self.flat.grad[a:b], op=dist.ReduceOp.SUM, group=self.pg)
        self._begun = True

    def finish_sync(self):
        """Join the reductions started by hot_path_done() / begin_sync().  Returns the grad_scale (1 / world)."""
        if self.world == 1:
            return 1.0
        for w in self._pending:
            w.wait()
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        self._pending, self._early_done, self._begun = [], False, False
        return 1.0 / self.world

    def sync_gradients(self):
        """Returns the grad_scale (1/world) to pass to the optimizer."""
        if self.world == 1:
            return 1.0
        self.begin_sync()
        return self.finish_sync()


# ------------------------------------------------------------------------------------------------ the step, split at the tokens
def _split_backward(tokens, run_hot, on_hot_done):
    """Backward in two stages around the FPN's output tokens: the hot path runs on detached leaves, its backward fills
    every hot-path parameter gradient and the token gradients; `on_hot_done()` (the early all-reduce) is called; then the
    token gradients are pushed through the FPN (MIOpen convolution backward).  Same gradients as one backward() call."""
    from .ops import TokenMap
    uniq, leaves = {}, []
    for t in tokens:
        tm = TokenMap.of(t)                  # tokens + the FPN output bias owed to gathered rows (Act3D.compute_visual_tokens)
        if id(tm.tokens) not in uniq:
            uniq[id(tm.tokens)] = (tm.tokens, tm.leaf())
        leaf = uniq[id(tm.tokens)][1]
        leaves.append(leaf if isinstance(t, TokenMap) else leaf.tokens)      # plain tensors in, plain leaves out
    loss = run_hot(leaves)
    loss.backward()
    if on_hot_done is not None:
        on_hot_done()
    pairs = [(t, l.tokens.grad) for t, l in uniq.values() if t.requires_grad and l.tokens.grad is not None]
    if pairs:
        torch.autograd.backward([t for t, _ in pairs], [g for _, g in pairs])
    return loss.detach()


def fwd_bwd_keypose(model, criterion, sample, use_gt_sampling=True, on_hot_done=None):
    """forward + loss + backward of main_keypose.py:207-224 with the backward split at the FPN tokens"""
    maps = sample.get("backbone_maps")               # prefetched by the previous step (GraphedStep(prefetch=...)); else computed here
    tokens = model.compute_visual_tokens(sample["rgbs"]) if maps is None else model.compute_visual_tokens(sample["rgbs"], maps=maps)

    def hot(leaves):
        out = model(sample["rgbs"], sample["pcds"], sample["instr"], sample["curr_gripper"],
                    gt_action=sample["action"] if use_gt_sampling else None, visual_features=leaves)
        return sum(criterion.compute_loss(out, sample).values())

    return _split_backward(tokens, hot, on_hot_done)


def fwd_bwd_trajectory(model, criterion, sample, on_hot_done=None):
    """forward + loss + backward of main_trajectory.py:177-195 with the backward split at the FPN tokens"""
    maps = sample.get("backbone_maps")               # prefetched by the previous step (GraphedStep(prefetch=...)); else computed here
    head = model.prediction_head
    tokens = head.encode_images(sample["rgbs"], None) if maps is None else head.encode_images(sample["rgbs"], None, maps=maps)
    multi = isinstance(tokens, (list, tuple))                  # one token tensor per scale for a multi-scale head

    # additive test hooks: a batch may carry the DDPM noise / timesteps to use instead of the device draws
    kw = {k: sample[k] for k in ("noise", "timesteps") if k in sample}

    def hot(leaves):
        return criterion.compute_loss(model(sample["trajectory"], sample["trajectory_mask"], sample["rgbs"], sample["pcds"],
                                            sample["instr"], sample["curr_gripper"], sample["action"],
                                            visual_tokens=list(leaves) if multi else leaves[0], **kw))

    return _split_backward(list(tokens) if multi else [tokens], hot, on_hot_done)


def _finish_step(optimizer, ddp, step_id, accumulate_grad_batches):
    if step_id % accumulate_grad_batches == accumulate_grad_batches - 1:
        scale = ddp.sync_gradients() if ddp is not None else 1.0
        optimizer.step(grad_scale=scale) if isinstance(optimizer, FlatAdamW) else optimizer.step()


def train_one_step(model, criterion, optimizer, step_id, sample, ddp=None, accumulate_grad_batches=1,
                   use_ground_truth_position_for_sampling_train=True):
    """TrainTester.train_one_step for the keypose model (main_keypose.py:207-234): zero_grad, forward, loss, backward,
    (all-reduce, overlapped with the FPN backward), optimizer step.  Returns the detached total loss."""
    if step_id % accumulate_grad_batches == 0:
        optimizer.zero_grad()
    last = step_id % accumulate_grad_batches == accumulate_grad_batches - 1
    if ddp is not None:
        ddp.arm(last)                       # only the window's final micro-batch may start reducing its gradients early
    loss = fwd_bwd_keypose(model, criterion, sample, use_ground_truth_position_for_sampling_train,
                           None if ddp is None else ddp.hot_path_done)
    _finish_step(optimizer, ddp, step_id, accumulate_grad_batches)
    return loss


def train_one_step_trajectory(model, criterion, optimizer, step_id, sample, ddp=None, accumulate_grad_batches=1):
    """TrainTester.train_one_step for the trajectory model (main_trajectory.py:177-204)."""
    if step_id % accumulate_grad_batches == 0:
        optimizer.zero_grad()
    last = step_id % accumulate_grad_batches == accumulate_grad_batches - 1
    if ddp is not None:
        ddp.arm(last)
    loss = fwd_bwd_trajectory(model, criterion, sample, None if ddp is None else ddp.hot_path_done)
    _finish_step(optimizer, ddp, step_id, accumulate_grad_batches)
    return loss


class JointStep:
    """One joint iteration of BASELINE.json configs[3]: an Act3D keypose training step AND a trajectory-diffusion training
    step -- two models, two optimizers, as the reference trains them (main_keypose.py:207-234, main_trajectory.py:177-204),
    each data-parallel over the same ranks (engine.py:121-124).

    Order of one iteration:  keypose forward + backward (hot segments reduced early, FlatDataParallel.hot_path_done) ->
    begin_sync() of its FPN segment on the side stream -> trajectory forward + backward ON THE MAIN STREAM, which hides the
    keypose reductions -> trajectory reduction -> both AdamW steps.  The result is the same as running the two
    train_one_step functions one after the other (tests/test_joint_gpu.py)."""

    def __init__(self, kp_model, kp_criterion, kp_optimizer, tr_model, tr_criterion, tr_optimizer, kp_ddp=None, tr_ddp=None,
                 use_ground_truth_position_for_sampling_train=True):
        self.kp = (kp_model, kp_criterion, kp_optimizer, kp_ddp)
        self.tr = (tr_model, tr_criterion, tr_optimizer, tr_ddp)
        self.use_gt = use_ground_truth_position_for_sampling_train

    def __call__(self, kp_sample, tr_sample):
        km, kc, ko, kd = self.kp
        tm, tc, to, td = self.tr
        ko.zero_grad()
        to.zero_grad()
        if kd is not None:
            kd.arm(True)
        loss_k = fwd_bwd_keypose(km, kc, kp_sample, self.use_gt, None if kd is None else kd.hot_path_done)
        if kd is not None:
            kd.begin_sync()
        if td is not None:
            td.arm(True)
        loss_t = fwd_bwd_trajectory(tm, tc, tr_sample, None if td is None else td.hot_path_done)
        if td is not None:
            td.begin_sync()
        ko.step(grad_scale=kd.finish_sync() if kd is not None else 1.0)
        to.step(grad_scale=td.finish_sync() if td is not None else 1.0)
        return loss_k, loss_t


def save_checkpoint(path, model, optimizer, step_id, best_loss=None):
    """engine.py:214-230 format: {"weight" (DDP's "module." prefix), "optimizer" (torch.optim.AdamW layout), "iter",
    "best_loss"} -- loadable by the reference's load_checkpoint (engine.py:195-212)."""
    weight = {"module." + k: v.detach().clone() for k, v in model.state_dict().items()}
    torch.save({"weight": weight, "optimizer": optimizer.state_dict(), "iter": step_id + 1, "best_loss": best_loss}, path)


def load_checkpoint(path, model, optimizer=None, strict=True):
    """engine.py:195-212 (+ online_evaluation/eval1.py:138-152 prefix stripping).  Returns (start_iter, best_loss).
    strict (the reference's load_state_dict default): missing / unexpected weight keys raise.  Parameters keep their
    storage (the flat buffer views): values are copied in place."""
    d = torch.load(path, map_location="cpu", weights_only=False)
    weight = {(k[7:] if k.startswith("module.") else k): v for k, v in d["weight"].items()}
    res = model.load_state_dict(weight, strict=False)
    if strict and (res.missing_keys or res.unexpected_keys):
        raise RuntimeError("checkpoint %s does not match the model: missing %s, unexpected %s" %
                           (path, res.missing_keys[:5], res.unexpected_keys[:5]))
    if optimizer is not None:
        if "optimizer" not in d:
            raise RuntimeError("checkpoint %s holds no optimizer state" % path)
        optimizer.load_state_dict(d["optimizer"])
    return d.get("iter", 0), d.get("best_loss", None)


# A/B switch (measured, round 6: both lose -- "main" 23.0 ms, default 18.9 ms per keypose step): "main" captures the prefetching step on
# a high-priority stream, "side" gives the prefetch stream the high priority instead
PREFETCH_HIPRIO = os.environ.get("A3D_PREFETCH_HIPRIO", "0")
# BatchNorm-apply workgroups of the prefetched backbone (a3d_bn_grid_cap; 0: leave the library default)
PREFETCH_BN_GRID = int(os.environ.get("A3D_PREFETCH_BN_GRID", "256"))
# A3D_PREFETCH_CHECK=1: every launch verifies that its images are the ones the previous launch announced (one host sync per step)
PREFETCH_CHECK = os.environ.get("A3D_PREFETCH_CHECK", "0") == "1"
# (forking the backbone after the FPN forward instead of at the start of the step measured the same, 18.86 vs 18.91 ms,
# profiles/r06_prefetch_ab.json: the knob was removed)


class GraphedStep:
    """Captures one full training step into hipGraphs and replays it.

    Every kernel of the hot path is enqueued on the caller's stream with static shapes and no host sync, the ghost
    sampler, the dropout generator and the AdamW step counter live on the device, so the step is capturable as is.
    Inputs are copied into static buffers before each replay.
      world == 1 : ONE graph  [zero_grad + forward + loss + backward + AdamW].
      world  > 1 : `step_fwd_bwd(inputs, on_hot_done)` must split its backward at the FPN tokens (fwd_bwd_keypose /
                   fwd_bwd_trajectory).  Three graphs around the two collectives:
                   [zero_grad + forward + hot-path backward] -> all-reduce(hot segments) on a side stream, concurrent with
                   [FPN backward] -> all-reduce(FPN segment) -> [AdamW with 1/world].
    The collectives themselves are issued eagerly between the replays (RCCL calls are not captured).

    prefetch (round 6; `model.backbone_maps`, a callable (rgbs, out=None) -> {name: map}): the FROZEN backbone of the NEXT batch runs
    on a side stream inside step k's graph, next to step k's FPN + hot path + backward + AdamW, which read the maps step k - 1
    left for them (`inputs["backbone_maps"]`, engine.fwd_bwd_keypose).  The backbone has no gradient and its weights never change,
    so the result of every step is the one of the sequential order (act3d.py:363-366 runs under no_grad in the reference too); what
    changes is that its ~9 ms of HBM-bound kernels fill the chip while the hot path's launch- and issue-bound kernels run.  Two map
    sets and two graph sets alternate (step k reads set k % 2 and writes set (k + 1) % 2).  Contract: launch(inputs, next_rgbs)
    -- `next_rgbs` are the images of the batch the NEXT launch will pass (None: the same static images again, a fixed synthetic
    batch); the first launch primes its own maps eagerly."""

    def __init__(self, step_fwd_bwd, optimizer, static_inputs, ddp=None, warmup=3, prefetch=None):
        self.static_inputs = static_inputs
        self.optimizer = optimizer
        self.ddp = ddp
        self.world = ddp.world if ddp is not None else 1
        self.prefetch = prefetch
        split = self.world > 1
        self.parity = 0
        self._primed = False
        sets = 1
        gkw = {}
        if prefetch is not None:
            sets = 2
            self.next_rgbs = static_inputs["rgbs"].clone()
            with torch.no_grad():
                m0 = prefetch(static_inputs["rgbs"])
            self.maps = [m0, {k: torch.empty_like(v) for k, v in m0.items()}]
            self._pf_stream = torch.cuda.Stream(priority=-1) if PREFETCH_HIPRIO == "side" else torch.cuda.Stream()
            if PREFETCH_HIPRIO in ("1", "main"):     # capture stream above the prefetch stream: the hot path's short kernels dispatch first
                gkw["stream"] = torch.cuda.Stream(priority=-1)
        inputs = [dict(static_inputs) for _ in range(sets)]
        if prefetch is not None:
            for p_ in range(sets):
                inputs[p_]["backbone_maps"] = self.maps[p_]
        call = (lambda p_, cb: step_fwd_bwd(inputs[p_], cb)) if split else (lambda p_, cb: step_fwd_bwd(inputs[p_]))
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                optimizer.zero_grad()
                call(0, ddp.hot_path_done if split else None)
                scale = ddp.sync_gradients() if ddp is not None else 1.0
                optimizer.step(grad_scale=scale)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()

        def fork(p_):
            """the next batch's backbone on the side stream (joined by join()); no-op without prefetch"""
            if prefetch is None:
                return
            self._pf_stream.wait_stream(torch.cuda.current_stream())
            # the overlapped backbone's BatchNorm-apply passes with one workgroup per CU: the hot path's kernels find free wave slots
            # (18.87 -> 18.45 ms per keypose step; grids are baked into the captured graph, so only this capture sees the setting)
            lib = L.load()
            prev = lib.a3d_bn_grid_cap(PREFETCH_BN_GRID) if PREFETCH_BN_GRID > 0 else 0
            try:
                with torch.cuda.stream(self._pf_stream), torch.no_grad():
                    prefetch(self.next_rgbs, out=self.maps[1 - p_])
            finally:
                if PREFETCH_BN_GRID > 0:
                    lib.a3d_bn_grid_cap(prev)

        def join():
            if prefetch is not None:
                torch.cuda.current_stream().wait_stream(self._pf_stream)

        self.g_fb, self.g_late, self.loss = [], [], []
        self.g_opt = None
        pool = None
        for p_ in range(sets):
            g = torch.cuda.CUDAGraph()
            self.g_fb.append(g)
            if not split:
                with (torch.cuda.graph(g, **gkw) if pool is None else torch.cuda.graph(g, pool=pool, **gkw)):
                    fork(p_)
                    optimizer.zero_grad()
                    self.loss.append(call(p_, None))
                    optimizer.step(grad_scale=1.0)
                    join()
                pool = g.pool()
                continue
            # world > 1: the capture of the first graph ends inside the callback (hot path done), the second one starts there
            g2 = torch.cuda.CUDAGraph()
            self.g_late.append(g2)
            ctx = {}

            def switch_graphs():
                join()                               # the prefetched backbone overlaps the forward + hot-path backward
                ctx["first"].__exit__(None, None, None)
                ctx["second"] = torch.cuda.graph(g2, pool=g.pool() if pool is None else pool, **gkw)
                ctx["second"].__enter__()

            ctx["first"] = torch.cuda.graph(g, **gkw) if pool is None else torch.cuda.graph(g, pool=pool, **gkw)
            ctx["first"].__enter__()
            try:
                fork(p_)
                optimizer.zero_grad()
                self.loss.append(call(p_, switch_graphs))
            finally:
                if "second" not in ctx:
                    join()                           # a step that never reached its split point: do not leave the side stream unjoined
                (ctx.get("second") or ctx["first"]).__exit__(None, None, None)
            pool = g.pool()
        if split:
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt, pool=pool, **gkw):
                optimizer.step(grad_scale=1.0 / self.world)
        self._last = 0

    def launch(self, inputs=None, next_rgbs=None):
        """Enqueue the step up to (not including) the point where the gradient reductions must have finished: input copies,
        forward + backward replay(s) and, under DP, the all-reduces (started, not awaited).  finish() completes the step.
        next_rgbs (prefetch only): the images of the batch the next launch will pass; None: the static images again."""
        if inputs is not None:
            for k, v in inputs.items():
                if torch.is_tensor(v) and k in self.static_inputs and v.data_ptr() != self.static_inputs[k].data_ptr():
                    self.static_inputs[k].copy_(v, non_blocking=True)
        p_ = 0
        if self.prefetch is not None:
            p_ = self.parity
            if self._primed and PREFETCH_CHECK and not torch.equal(self.static_inputs["rgbs"], self.next_rgbs):
                # (a host synchronisation per step: a debugging aid, off by default)
                raise RuntimeError("GraphedStep(prefetch): this launch's images are not the next_rgbs the previous launch was given -- "
                                   "the maps prefetched for it belong to another batch")
            if not self._primed:                 # the first launch computes its own maps (nobody prefetched them)
                with torch.no_grad():
                    self.prefetch(self.static_inputs["rgbs"], out=self.maps[p_])
                self._primed = True
            src = next_rgbs if next_rgbs is not None else self.static_inputs["rgbs"]
            if src.data_ptr() != self.next_rgbs.data_ptr():
                self.next_rgbs.copy_(src, non_blocking=True)
            self.parity ^= 1
        self._last = p_
        self.g_fb[p_].replay()
        if self.world > 1:
            self.ddp.arm(True)
            self.ddp.hot_path_done()           # hot segments on the side stream (whole buffer later if overlap is off) ...
            self.g_late[p_].replay()           # ... while the FPN backward runs
            self.ddp.begin_sync()

    def finish(self):
        if self.world > 1:
            self.ddp.finish_sync()
            self.g_opt.replay()
        return self.loss[self._last]

    def __call__(self, inputs=None, next_rgbs=None):
        self.launch(inputs, next_rgbs)
        return self.finish()


class GraphedJointStep:
    """JointStep on captured graphs: one GraphedStep per model; the keypose reductions are started before the trajectory
    graphs are replayed and awaited after them, so the trajectory forward / backward hides them (same order of operations as
    JointStep.__call__)."""

    def __init__(self, keypose_step, trajectory_step):
        self.kp, self.tr = keypose_step, trajectory_step

    def __call__(self, kp_sample=None, tr_sample=None):
        self.kp.launch(kp_sample)
        self.tr.launch(tr_sample)
        return self.kp.finish(), self.tr.finish()
