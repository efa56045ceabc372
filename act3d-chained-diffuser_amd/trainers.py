"""Train / evaluate drivers on the MI355X path, behind the reference's driver API.

API contract (names, arguments, return values): `BaseTrainTester` <-> `engine.py:18-248`, `KeyposeTrainTester` <->
`main_keypose.py:97-281` (`TrainTester`), `TrajectoryTrainTester` <-> `main_trajectory.py:86-274`; `args` is any namespace
with the reference's `Arguments` fields.  The machinery underneath is this package's own:

  StepRunner   owns (model, criterion, optimizer, ddp) and a small set of captured hipGraphs keyed by the batch signature:
               the second time a batch shape shows up its whole step (zero_grad + forward + loss + backward [+ the
               all-reduces between three graphs under DP] + fused AdamW) is captured (engine.GraphedStep) and from then on
               replayed -- `main()` therefore runs what `bench.py` times; odd shapes and gradient accumulation run eagerly.
  MetricTable  evaluation statistics as (sum, count) per key on the device: one row per batch, per-task rows through a group
               mask, ranks combined with ONE all-reduce of the table (the reference gathers python dicts of growing tensors).
  _Schedule    the iteration plan (what happens after step i) as data instead of inline modulo tests.
Batches arrive through `data.DeviceLoader` (pinned copies + the GPU Resize augmentation one batch ahead).
"""
import json
import os
import pickle
import random

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, default_collate
from torch.utils.data.distributed import DistributedSampler

from . import data as D
from . import engine as E
from .act3d import Act3D
from .diffusion import DiffusionPlanner
from .losses import LossAndMetrics, TrajectoryCriterion


# ------------------------------------------------------------------------------------------------ process group helpers
def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def all_gather(data):
    """A picklable object from every rank, ordered by rank (engine.py:251-297's contract)."""
    if get_world_size() == 1:
        return [data]
    bucket = [None] * get_world_size()
    dist.all_gather_object(bucket, data)
    return bucket


# ------------------------------------------------------------------------------------------------ small file readers
def load_instructions(instructions, tasks=None, variations=None):
    """{task: {variation: embedding}} from the pickle at `instructions`, restricted to `tasks` / `variations` when given
    (utils/utils_without_rlbench.py:79-98's contract); None passes through."""
    if instructions is None:
        return None
    with open(instructions, "rb") as fh:
        table = pickle.load(fh)
    keep_task = (lambda t: True) if tasks is None else set(tasks).__contains__
    keep_var = (lambda v: True) if variations is None else set(variations).__contains__
    return {t: {v: emb for v, emb in per_var.items() if keep_var(v)} for t, per_var in table.items() if keep_task(t)}


def get_gripper_loc_bounds(path, buffer=0.0, task=None):
    """(2, 3) array [min corner, max corner] of the workspace: the named task's box if the json has it, else the box around
    every task's, padded by `buffer` on each side (utils/utils_without_rlbench.py:54-68's contract)."""
    with open(path) as fh:
        boxes = {k: np.asarray(v, dtype=np.float64) for k, v in json.load(fh).items()}
    if task in boxes:
        box = boxes[task]
    else:
        stack = np.stack(list(boxes.values()))                       # (tasks, 2, 3)
        box = np.stack([stack[:, 0].min(0), stack[:, 1].max(0)])
    return box + np.array([[-buffer], [buffer]])


class _Scalars:
    """Where logged scalars go: the last value per key is always kept (`.last`), tensorboard gets a copy when importable."""

    def __init__(self, log_dir):
        self.last = {}
        try:
            from torch.utils.tensorboard import SummaryWriter
            self._tb = SummaryWriter(log_dir=log_dir)
        except Exception:
            self._tb = None

    def add_scalar(self, key, value, step):
        self.last[key] = (float(value), step)
        if self._tb is not None:
            self._tb.add_scalar(key, value, step)


# ------------------------------------------------------------------------------------------------ evaluation statistics
class MetricTable:
    """Running (sum, count) per metric key, kept on the device.  `add` takes one scalar per call (a batch statistic),
    `add_grouped` one per-sample vector plus the samples' group labels (per-task means of a batch).  `means()` reduces over
    ranks with a single all-reduce and returns python floats: sum / count == the mean over the concatenation of every
    rank's per-batch entries, which is what the reference's gather-and-cat computes (engine.py:232-245)."""

    def __init__(self, device):
        self.device = device
        self._slot = {}
        self._acc = torch.zeros((0, 2), device=device, dtype=torch.float64)

    def _index(self, key):
        if key not in self._slot:
            self._slot[key] = len(self._slot)
            self._acc = torch.cat([self._acc, torch.zeros((1, 2), device=self.device, dtype=torch.float64)])
        return self._slot[key]

    def add(self, key, value):
        i = self._index(key)
        self._acc[i, 0] += value.detach().reshape(()).to(self.device, torch.float64)
        self._acc[i, 1] += 1

    def add_grouped(self, prefix, name, per_sample, labels):
        """one entry per distinct label: the mean of `per_sample` over the samples carrying it"""
        labels = np.asarray(labels)
        for lab in np.unique(labels):
            member = torch.from_numpy(labels == lab).to(per_sample.device)
            self.add(f"{prefix}/{lab}/{name}", per_sample[member].mean())

    def means(self, across_ranks=False):
        keys, acc = list(self._slot), self._acc
        if across_ranks and get_world_size() > 1:
            # ranks may have met different keys (tasks): agree on the union first, then one all-reduce of the table
            union = sorted(set().union(*all_gather(keys)))
            full = torch.zeros((len(union), 2), device=self.device, dtype=torch.float64)
            for k in keys:
                full[union.index(k)] = acc[self._slot[k]]
            dist.all_reduce(full)
            keys, acc = union, full
            return {k: (acc[i, 0] / acc[i, 1]).item() for i, k in enumerate(keys) if acc[i, 1] > 0}
        return {k: (acc[self._slot[k], 0] / acc[self._slot[k], 1]).item() for k in keys}


# ------------------------------------------------------------------------------------------------ the step
def _signature(sample):
    return tuple(sorted((k, tuple(v.shape), str(v.dtype)) for k, v in sample.items() if torch.is_tensor(v)))


class StepRunner:
    """Runs training steps for one (model, criterion, optimizer[, ddp]): captured hipGraph replays for batch shapes that
    recur, the eager engine step otherwise.  `fwd_bwd(model, criterion, sample, on_hot_done)` is the split forward /
    backward of the model family (engine.fwd_bwd_keypose / fwd_bwd_trajectory); `eager_step(step_id, sample)` the full eager
    step incl. gradient accumulation."""

    def __init__(self, model, criterion, optimizer, ddp, fwd_bwd, eager_step, accumulate=1, max_graphs=4, enable=True):
        self.model, self.criterion, self.optimizer, self.ddp = model, criterion, optimizer, ddp
        self.fwd_bwd, self.eager_step = fwd_bwd, eager_step
        self.accumulate = max(1, int(accumulate))
        self.max_graphs = max_graphs
        self.enable = enable and self.accumulate == 1 and os.environ.get("A3D_TRAINER_GRAPHS", "1") == "1"
        self._seen = {}
        self._graphs = {}
        self.replays = 0

    def _capture(self, sample):
        static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in sample.items()}
        fb = (lambda s, cb=None: self.fwd_bwd(self.model, self.criterion, s, cb)) if self.ddp is not None and self.ddp.world > 1 \
            else (lambda s: self.fwd_bwd(self.model, self.criterion, s, None))
        # warmup = 0: an eager step with exactly these shapes has already run (first sighting), so every lazily created
        # workspace / MIOpen plan exists and no optimizer step is spent on warm-up replays of one batch
        return E.GraphedStep(fb, self.optimizer, static, ddp=self.ddp if (self.ddp is not None and self.ddp.world > 1) else None,
                             warmup=0)

    def __call__(self, step_id, sample):
        if not self.enable or not self.model.training:
            return self.eager_step(step_id, sample)
        sig = _signature(sample)
        graph = self._graphs.get(sig)
        if graph is None:
            self._seen[sig] = self._seen.get(sig, 0) + 1
            if self._seen[sig] < 2 or len(self._graphs) >= self.max_graphs:
                return self.eager_step(step_id, sample)
            try:
                graph = self._graphs[sig] = self._capture(sample)
            except Exception as exc:                      # a shape that cannot be captured keeps running eagerly, loudly
                print(f"[StepRunner] graph capture failed for {sig[:2]}...: {exc!r}; this shape runs eagerly")
                self._seen[sig] = -(1 << 30)
                return self.eager_step(step_id, sample)
        self.replays += 1
        return graph(sample)


class _Schedule:
    """What follows training step `i`: an evaluation round every `val_freq` steps (which is also when scalars are logged)."""

    def __init__(self, start, total, val_freq):
        self.start, self.total, self.val_freq = start, total, val_freq

    def steps(self):
        return range(self.start, self.total)

    def evaluates_after(self, step_id):
        return (step_id + 1) % self.val_freq == 0


def _cycle(loader):
    """batches forever: a fresh pass over `loader` whenever it runs out"""
    while True:
        got = False
        for batch in loader:
            got = True
            yield batch
        if not got:
            raise RuntimeError("the training loader yields no batches")


def _seed_worker(worker_id):
    """DataLoader worker seeding: python's RNG from the worker's torch seed, numpy's from that seed + the worker id -- the
    state the reference's seed_worker leaves behind (engine.py:44-49), which the dataset's draws depend on."""
    base = torch.initial_seed() % (1 << 32)
    random.seed(base)
    np.random.seed(base + worker_id)


# ------------------------------------------------------------------------------------------------ drivers
class BaseTrainTester:
    """The driver skeleton.  Subclasses say what the datasets, the model, the criterion, one training step and one
    evaluation round are; `main` wires loaders -> model -> optimizer -> (DP) -> resume -> the scheduled loop."""

    split_fwd_bwd = None            # engine.fwd_bwd_* of the model family (set by the subclass)

    def __init__(self, args):
        self.args = args
        self.writer = _Scalars(getattr(args, "log_dir", None)) if get_rank() == 0 else None
        self.ddp = None
        self._runner = None

    @property
    def scalars(self):
        return {} if self.writer is None else self.writer.last

    @property
    def device(self):
        return torch.device("cuda", getattr(self.args, "local_rank", 0))

    # ---- to be provided
    def get_datasets(self):
        return None, None

    def get_model(self):
        return None

    def get_criterion(self):
        return None

    def train_one_step(self, model, criterion, optimizer, step_id, sample):
        pass

    @torch.no_grad()
    def evaluate_nsteps(self, model, criterion, loader, step_id, val_iters, split='val'):
        return None

    # ---- shared
    def get_loaders(self, collate_fn=default_collate):
        """Rank-sharded loaders whose batches arrive on the device.  Train: drop_last, shuffled by the DistributedSampler;
        test: shuffled as well (the reference does, engine.py:66), no worker processes."""
        train_set, test_set = self.get_datasets()
        gen = torch.Generator()
        gen.manual_seed(0)
        shard = dict(num_replicas=get_world_size(), rank=get_rank())
        common = dict(shuffle=False, worker_init_fn=_seed_worker, collate_fn=collate_fn, pin_memory=True, generator=gen)
        train = DataLoader(train_set, batch_size=self.args.batch_size, num_workers=self.args.num_workers, drop_last=True,
                           sampler=DistributedSampler(train_set, **shard), **common)
        test = DataLoader(test_set, batch_size=self.args.batch_size_val, num_workers=0, drop_last=False,
                          sampler=DistributedSampler(test_set, shuffle=True, **shard), **common)
        return D.DeviceLoader(train, self.device), D.DeviceLoader(test, self.device)

    def get_optimizer(self, model):
        """The reference's two AdamW groups (engine.py:89-102) as ONE fused optimizer over flat buffers (`optimizer.flat`)."""
        return E.get_optimizer(model, lr=self.args.lr)[1]

    def _val_iters(self):
        return max(5, int(4 * len(self.args.tasks) / self.args.batch_size_val))

    def _evaluation_round(self, model, criterion, train_loader, test_loader, step_id):
        model.eval()
        self.evaluate_nsteps(model, criterion, train_loader, step_id, val_iters=self._val_iters(), split='train')
        model.eval()
        new_loss = self.evaluate_nsteps(model, criterion, test_loader, step_id, val_iters=self._val_iters())
        model.train()
        return new_loss

    def main(self, collate_fn=default_collate):
        train_loader, test_loader = self.get_loaders(collate_fn)
        model = self.get_model().to(self.device)
        criterion = self.get_criterion()
        optimizer = self.get_optimizer(model)
        if get_world_size() > 1:
            self.ddp = E.FlatDataParallel(optimizer.flat, overlap=True, model=model)
            self.ddp.broadcast_parameters()
        start, best_loss = 0, None
        if getattr(self.args, "checkpoint", None):
            if not os.path.isfile(self.args.checkpoint):
                raise FileNotFoundError(self.args.checkpoint)
            start, best_loss = self.load_checkpoint(model, optimizer)
        if bool(getattr(self.args, "eval_only", 0)):
            model.eval()
            self.evaluate_nsteps(model, criterion, test_loader, step_id=-1, val_iters=self._val_iters())
            return model
        plan = _Schedule(start, self.args.train_iters, self.args.val_freq)
        batches = _cycle(train_loader)
        model.train()
        for step_id in plan.steps():
            self.train_one_step(model, criterion, optimizer, step_id, next(batches))
            if plan.evaluates_after(step_id):
                new_loss = self._evaluation_round(model, criterion, train_loader, test_loader, step_id)
                if get_rank() == 0:
                    best_loss = self.save_checkpoint(model, optimizer, step_id, new_loss, best_loss)
        return model

    def _runner_for(self, model, criterion, optimizer, eager_step):
        r = self._runner
        if r is None or r.model is not model or r.optimizer is not optimizer or r.criterion is not criterion:
            r = self._runner = StepRunner(model, criterion, optimizer, self.ddp, type(self).split_fwd_bwd, eager_step,
                                          accumulate=getattr(self.args, "accumulate_grad_batches", 1))
        return r

    def _log_train(self, loss, step_id):
        if self.writer is not None and (step_id + 1) % self.args.val_freq == 0:
            self.writer.add_scalar("lr", self.args.lr, step_id)
            self.writer.add_scalar("train-loss/noise_mse", loss, step_id)

    def load_checkpoint(self, model, optimizer):
        """Weights + optimizer state from args.checkpoint; the learning rate is reset to args.lr as the reference does
        (engine.py:195-212).  Returns (start_iter, best_loss)."""
        start_iter, best_loss = E.load_checkpoint(self.args.checkpoint, model, optimizer)
        optimizer.lr = self.args.lr
        print(f"resumed from {self.args.checkpoint} at step {start_iter}")
        return start_iter, best_loss

    def save_checkpoint(self, model, optimizer, step_id, new_loss, best_loss):
        """last.pth every time; best.pth too unless the new validation loss is worse than the best so far (a missing loss
        counts as an improvement, engine.py:214-230).  Returns the best loss."""
        out = str(self.args.log_dir)
        improved = new_loss is None or best_loss is None or new_loss <= best_loss
        if improved:
            best_loss = new_loss
            E.save_checkpoint(os.path.join(out, "best.pth"), model, optimizer, step_id, best_loss)
        E.save_checkpoint(os.path.join(out, "last.pth"), model, optimizer, step_id, best_loss)
        return best_loss

    def synchronize_between_processes(self, a_dict):
        """{key: 1-D tensor} from every rank -> on rank 0 the per-key concatenation over ranks, elsewhere the input
        (engine.py:232-245's contract; the drivers here reduce a MetricTable instead)."""
        per_rank = all_gather(a_dict)
        if get_rank() != 0:
            return a_dict
        first = per_rank[0]
        return {k: torch.cat([d[k].to(first[k].device) for d in per_rank if k in d]) for k in first}

    def _report(self, values, step_id, to_writer=True):
        if get_rank() != 0:
            return
        if to_writer:
            for key, val in values.items():
                self.writer.add_scalar(key, val, step_id)
        print(f"[step {step_id}] " + "  ".join(f"{k}={v:.3f}" for k, v in values.items()))


def _dataset_kwargs(a, instruction, trajectories):
    taskvar = [(task, var) for task, per_var in instruction.items() for var in per_var]
    return dict(instructions=instruction, taskvar=taskvar, max_episode_length=a.max_episode_length,
                max_episodes_per_task=a.max_episodes_per_task, cameras=a.cameras, gripper_loc_bounds=a.gripper_loc_bounds,
                image_rescale=tuple(float(x) for x in a.image_rescale.split(",")),
                point_cloud_rotate_yaw_range=a.point_cloud_rotate_yaw_range, return_low_lvl_trajectory=trajectories,
                dense_interpolation=bool(a.dense_interpolation) if trajectories else False,
                interpolation_length=a.interpolation_length if trajectories else 0,
                action_dim=a.action_dim if trajectories else 8, predict_short=False)


def _datasets(a, trajectories):
    instruction = load_instructions(a.instructions, tasks=a.tasks, variations=a.variations)
    if instruction is None:
        raise NotImplementedError("training without an instruction file is not implemented (nor in the reference)")
    kw = _dataset_kwargs(a, instruction, trajectories)
    return (D.RLBenchDataset(root=a.dataset, cache_size=a.cache_size, num_iters=a.train_iters, training=True, **kw),
            D.RLBenchDataset(root=a.valset, cache_size=a.cache_size_val, training=False, **kw))


class KeyposeTrainTester(BaseTrainTester):
    """Act3D keypose training / evaluation (main_keypose.py:97-281)."""

    @staticmethod
    def split_fwd_bwd(model, criterion, sample, cb, use_gt=True):
        return E.fwd_bwd_keypose(model, criterion, sample, use_gt, cb)

    def get_datasets(self):
        return _datasets(self.args, trajectories=False)

    def get_model(self):
        a = self.args
        return Act3D(backbone=a.backbone, image_size=tuple(int(x) for x in a.image_size.split(",")),
                     embedding_dim=a.embedding_dim,
                     num_ghost_point_cross_attn_layers=a.num_ghost_point_cross_attn_layers,
                     num_query_cross_attn_layers=a.num_query_cross_attn_layers,
                     num_vis_ins_attn_layers=a.num_vis_ins_attn_layers, rotation_parametrization=a.rotation_parametrization,
                     gripper_loc_bounds=a.gripper_loc_bounds, num_ghost_points=a.num_ghost_points,
                     num_ghost_points_val=a.num_ghost_points_val, weight_tying=bool(a.weight_tying),
                     gp_emb_tying=bool(a.gp_emb_tying), num_sampling_level=a.num_sampling_level,
                     fine_sampling_ball_diameter=a.fine_sampling_ball_diameter,
                     regress_position_offset=bool(a.regress_position_offset), use_instruction=bool(a.use_instruction),
                     sampler_seed=_rank_seed(a, 0x5A17))

    def get_criterion(self):
        a = self.args
        return LossAndMetrics(position_prediction_only=bool(a.position_prediction_only),
                              rotation_parametrization=a.rotation_parametrization, position_loss=a.position_loss,
                              compute_loss_at_all_layers=bool(a.compute_loss_at_all_layers),
                              ground_truth_gaussian_spread=a.ground_truth_gaussian_spread, label_smoothing=a.label_smoothing,
                              position_loss_coeff=a.position_loss_coeff,
                              position_offset_loss_coeff=a.position_offset_loss_coeff,
                              rotation_loss_coeff=a.rotation_loss_coeff, gripper_loss_coeff=a.gripper_loss_coeff,
                              symmetric_rotation_loss=bool(a.symmetric_rotation_loss))

    def train_one_step(self, model, criterion, optimizer, step_id, sample):
        use_gt = bool(self.args.use_ground_truth_position_for_sampling_train)

        def eager(i, s):
            return E.train_one_step(model, criterion, optimizer, i, s, ddp=self.ddp,
                                    accumulate_grad_batches=self.args.accumulate_grad_batches,
                                    use_ground_truth_position_for_sampling_train=use_gt)

        runner = self._runner_for(model, criterion, optimizer, eager)
        runner.fwd_bwd = lambda m, c, s, cb: E.fwd_bwd_keypose(m, c, s, use_gt, cb)
        runner.eager_step = eager
        loss = runner(step_id, sample)
        self._log_train(loss, step_id)
        return loss

    @torch.no_grad()
    def evaluate_nsteps(self, model, criterion, loader, step_id, val_iters, split='val'):
        """Free-running forward (no ground-truth anchor for the ghost points) over `val_iters` batches; every metric of
        LossAndMetrics.compute_metrics averaged over the batches.  Returns the entry 'val-losses/action_mse' -- a key the
        keypose metrics never produce, so None, as in the reference (main_keypose.py:236-281): every validation round
        overwrites best.pth."""
        table = MetricTable(next(model.parameters()).device)
        model.eval()
        for _, sample in zip(range(val_iters), loader):
            pred = model(sample["rgbs"], sample["pcds"], sample["instr"], sample["curr_gripper"], gt_action=None)
            for name, value in criterion.compute_metrics(pred, sample).items():
                table.add(f"{split}-losses/{name}", value)
        values = table.means()
        self._report(values, step_id)
        return values.get('val-losses/action_mse')


class TrajectoryTrainTester(BaseTrainTester):
    """ChainedDiffuser trajectory-diffusion training / evaluation (main_trajectory.py:86-274)."""

    @staticmethod
    def split_fwd_bwd(model, criterion, sample, cb):
        return E.fwd_bwd_trajectory(model, criterion, sample, cb)

    def get_datasets(self):
        return _datasets(self.args, trajectories=True)

    def get_model(self):
        a = self.args
        return DiffusionPlanner(backbone=a.backbone, image_size=tuple(int(x) for x in a.image_size.split(",")),
                                embedding_dim=a.embedding_dim, output_dim=a.action_dim,
                                num_vis_ins_attn_layers=a.num_vis_ins_attn_layers,
                                num_query_cross_attn_layers=a.num_query_cross_attn_layers,
                                use_instruction=bool(a.use_instruction), use_goal=bool(a.use_goal),
                                use_goal_at_test=bool(a.use_goal_at_test), feat_scales_to_use=a.feat_scales_to_use,
                                attn_rounds=a.attn_rounds, weight_tying=bool(a.weight_tying),
                                gripper_loc_bounds=a.gripper_loc_bounds, rotation_parametrization=a.rotation_parametrization,
                                diffusion_timesteps=a.diffusion_timesteps, dropout_seed=_rank_seed(a, 0xD807))

    @staticmethod
    def get_criterion():
        return TrajectoryCriterion()

    def train_one_step(self, model, criterion, optimizer, step_id, sample):
        def eager(i, s):
            return E.train_one_step_trajectory(model, criterion, optimizer, i, s, ddp=self.ddp,
                                               accumulate_grad_batches=self.args.accumulate_grad_batches)

        runner = self._runner_for(model, criterion, optimizer, eager)
        runner.eager_step = eager
        loss = runner(step_id, sample)
        self._log_train(loss, step_id)
        return loss

    @torch.no_grad()
    def evaluate_nsteps(self, model, criterion, loader, step_id, val_iters, split='val'):
        """100-step sampling per batch (run_inference=True); summary metrics per batch and per-task means of the
        per-trajectory metrics, combined over ranks.  Returns the mean 'val-losses/traj_action_mse'
        (main_trajectory.py:206-274; its tensorboard trajectory plots need matplotlib and are out of scope)."""
        device = next(model.parameters()).device
        table = MetricTable(device)
        model.eval()
        for _, sample in zip(range(val_iters), loader):
            s = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in sample.items()}
            traj = model(s["trajectory"], s["trajectory_mask"], s["rgbs"], s["pcds"], s["instr"], s["curr_gripper"], s["action"],
                         run_inference=True)
            summary, per_traj = criterion.compute_metrics(traj, s["trajectory"], s["trajectory_mask"])
            for name, value in summary.items():
                table.add(f"{split}-losses/{name}", value)
            for name, vec in per_traj.items():
                table.add_grouped(f"{split}-loss", name, vec, sample["task"])
        values = table.means(across_ranks=True)
        self._report(values, step_id, to_writer=step_id > -1)
        return values.get('val-losses/traj_action_mse')


def _rank_seed(args, salt):
    """Seed of a model-owned device generator (ghost sampler, dropout): args.seed mixed with the rank, so that ranks and
    restarts with different seeds draw different streams (the reference seeds torch / numpy globally per process)."""
    return (int(getattr(args, "seed", 0)) * 1000003 + get_rank() * 7919 + salt) & 0x7FFFFFFF
