"""The train / evaluate drivers of the reference on the MI355X path.

`BaseTrainTester` mirrors `engine.py:18-248` (get_loaders / get_optimizer / main / load_checkpoint / save_checkpoint /
synchronize_between_processes), `KeyposeTrainTester` mirrors `main_keypose.py:97-281` and `TrajectoryTrainTester`
`main_trajectory.py:86-274`: same method names, argument meaning and return values, with `args` any namespace carrying
the reference's `Arguments` fields.  What differs underneath: parameters / gradients live in flat buffers
(`engine.FlatParams`), the optimizer is the fused `FlatAdamW` (state-dict layout of torch.optim.AdamW), DDP is
`engine.FlatDataParallel` (one all-reduce of the flat gradient buffer, overlapped with the FPN backward), batches arrive
through `data.DeviceLoader` (pinned copies + GPU augmentation one batch ahead).  Logging goes to tensorboard when the
package is importable and to `self.scalars` always.
"""
import os
import pickle

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


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def all_gather(data):
    """engine.py:251-297: gather an arbitrary picklable object from every rank (list ordered by rank)."""
    world = get_world_size()
    if world == 1:
        return [data]
    out = [None] * world
    dist.all_gather_object(out, data)
    return out


class _Scalars:
    """SummaryWriter stand-in: keeps the last value per key (and forwards to tensorboard when it is installed)."""

    def __init__(self, log_dir):
        self.last = {}
        self._tb = None
        try:
            from torch.utils.tensorboard import SummaryWriter
            self._tb = SummaryWriter(log_dir=log_dir)
        except Exception:
            pass

    def add_scalar(self, key, value, step):
        self.last[key] = (float(value), step)
        if self._tb is not None:
            self._tb.add_scalar(key, value, step)


def load_instructions(instructions, tasks=None, variations=None):
    """utils/utils_without_rlbench.py:79-98"""
    if instructions is None:
        return None
    with open(instructions, "rb") as fid:
        data = pickle.load(fid)
    if tasks is not None:
        data = {task: vi for task, vi in data.items() if task in tasks}
    if variations is not None:
        data = {task: {var: ins for var, ins in vi.items() if var in variations} for task, vi in data.items()}
    return data


def get_gripper_loc_bounds(path, buffer=0.0, task=None):
    """utils/utils_without_rlbench.py:54-68: one task's workspace, or the union over all tasks, grown by `buffer`."""
    import json
    with open(path, "r") as f:
        bounds = json.load(f)
    if task is not None and task in bounds:
        lo, hi = np.array(bounds[task][0]), np.array(bounds[task][1])
    else:
        lo = np.min(np.stack([b[0] for b in bounds.values()]), axis=0)
        hi = np.max(np.stack([b[1] for b in bounds.values()]), axis=0)
    return np.stack([lo - buffer, hi + buffer])


class BaseTrainTester:
    """engine.py:18-248"""

    def __init__(self, args):
        self.args = args
        self.writer = _Scalars(getattr(args, "log_dir", None)) if get_rank() == 0 else None
        self.ddp = None

    @property
    def scalars(self):
        return {} if self.writer is None else self.writer.last

    def get_datasets(self):
        return None, None

    def get_loaders(self, collate_fn=default_collate):
        """engine.py:38-77: DistributedSampler + DataLoader (pinned), wrapped so that batches arrive on the device."""
        import random

        def seed_worker(worker_id):
            worker_seed = torch.initial_seed() % 2 ** 32
            np.random.seed(worker_seed)
            random.seed(worker_seed)
            np.random.seed(np.random.get_state()[1][0] + worker_id)

        train_dataset, test_dataset = self.get_datasets()
        g = torch.Generator()
        g.manual_seed(0)
        world, rank = get_world_size(), get_rank()
        device = torch.device("cuda", getattr(self.args, "local_rank", 0))
        train_loader = DataLoader(train_dataset, batch_size=self.args.batch_size, shuffle=False,
                                  num_workers=self.args.num_workers, worker_init_fn=seed_worker, collate_fn=collate_fn,
                                  pin_memory=True, sampler=DistributedSampler(train_dataset, num_replicas=world, rank=rank),
                                  drop_last=True, generator=g)
        test_loader = DataLoader(test_dataset, batch_size=self.args.batch_size_val, shuffle=False, num_workers=0,
                                 worker_init_fn=seed_worker, collate_fn=collate_fn, pin_memory=True,
                                 sampler=DistributedSampler(test_dataset, num_replicas=world, rank=rank, shuffle=True),
                                 drop_last=False, generator=g)
        return D.DeviceLoader(train_loader, device), D.DeviceLoader(test_loader, device)

    def get_model(self):
        return None

    def get_criterion(self):
        return None

    def get_optimizer(self, model):
        """engine.py:89-102.  The returned FlatAdamW owns the flat parameter / gradient buffers (`optimizer.flat`)."""
        _, opt = E.get_optimizer(model, lr=self.args.lr)
        return opt

    def _val_iters(self):
        return max(5, int(4 * len(self.args.tasks) / self.args.batch_size_val))

    def main(self, collate_fn=default_collate):
        """engine.py:104-181"""
        train_loader, test_loader = self.get_loaders(collate_fn)
        model = self.get_model()
        criterion = self.get_criterion()
        model = model.to(torch.device("cuda", getattr(self.args, "local_rank", 0)))
        optimizer = self.get_optimizer(model)
        if get_world_size() > 1:
            self.ddp = E.FlatDataParallel(optimizer.flat, overlap=True, model=model)
            self.ddp.broadcast_parameters()

        start_iter, best_loss = 0, None
        if getattr(self.args, "checkpoint", None):
            assert os.path.isfile(self.args.checkpoint)
            start_iter, best_loss = self.load_checkpoint(model, optimizer)

        if bool(getattr(self.args, "eval_only", 0)):
            print("Test evaluation.......")
            model.eval()
            self.evaluate_nsteps(model, criterion, test_loader, step_id=-1, val_iters=self._val_iters())
            return model

        iter_loader = iter(train_loader)
        model.train()
        for step_id in range(start_iter, self.args.train_iters):
            try:
                sample = next(iter_loader)
            except StopIteration:
                iter_loader = iter(train_loader)
                sample = next(iter_loader)
            self.train_one_step(model, criterion, optimizer, step_id, sample)
            if (step_id + 1) % self.args.val_freq == 0:
                print("Train evaluation.......")
                model.eval()
                self.evaluate_nsteps(model, criterion, train_loader, step_id, val_iters=self._val_iters(), split='train')
                print("Test evaluation.......")
                model.eval()
                new_loss = self.evaluate_nsteps(model, criterion, test_loader, step_id, val_iters=self._val_iters())
                if get_rank() == 0:
                    best_loss = self.save_checkpoint(model, optimizer, step_id, new_loss, best_loss)
                model.train()
        return model

    def train_one_step(self, model, criterion, optimizer, step_id, sample):
        pass

    @torch.no_grad()
    def evaluate_nsteps(self, model, criterion, loader, step_id, val_iters, split='val'):
        return None

    def load_checkpoint(self, model, optimizer):
        """engine.py:195-212"""
        print("=> loading checkpoint '{}'".format(self.args.checkpoint))
        start_iter, best_loss = E.load_checkpoint(self.args.checkpoint, model, optimizer)
        optimizer.lr = self.args.lr
        print("=> loaded successfully '{}' (step {})".format(self.args.checkpoint, start_iter))
        return start_iter, best_loss

    def save_checkpoint(self, model, optimizer, step_id, new_loss, best_loss):
        """engine.py:214-230: last.pth always, best.pth when the validation loss did not get worse."""
        if new_loss is None or best_loss is None or new_loss <= best_loss:
            best_loss = new_loss
            E.save_checkpoint(os.path.join(str(self.args.log_dir), "best.pth"), model, optimizer, step_id, best_loss)
        E.save_checkpoint(os.path.join(str(self.args.log_dir), "last.pth"), model, optimizer, step_id, best_loss)
        return best_loss

    def synchronize_between_processes(self, a_dict):
        """engine.py:232-245: concatenate every rank's per-key tensors on rank 0."""
        all_dicts = all_gather(a_dict)
        if not is_dist_avail_and_initialized() or dist.get_rank() == 0:
            merged = {}
            for key in all_dicts[0].keys():
                device = all_dicts[0][key].device
                merged[key] = torch.cat([p[key].to(device) for p in all_dicts if key in p])
            a_dict = merged
        return a_dict

    def _log(self, values, step_id, always=True):
        if get_rank() == 0:
            if always or step_id > -1:
                for key, val in values.items():
                    self.writer.add_scalar(key, val, step_id)
            print(f"Step {step_id}:")
            for key, value in values.items():
                print(f"{key}: {value:.03f}")


def _append(values, key, item, device):
    if key not in values:
        values[key] = torch.empty((0,), device=device)
    values[key] = torch.cat([values[key], item.reshape(1).to(device)])


class KeyposeTrainTester(BaseTrainTester):
    """main_keypose.py:97-281 (`TrainTester`)"""

    def get_datasets(self):
        a = self.args
        instruction = load_instructions(a.instructions, tasks=a.tasks, variations=a.variations)
        if instruction is None:
            raise NotImplementedError()
        taskvar = [(task, var) for task, var_instr in instruction.items() for var in var_instr.keys()]
        common = dict(instructions=instruction, taskvar=taskvar, max_episode_length=a.max_episode_length,
                      max_episodes_per_task=a.max_episodes_per_task, cameras=a.cameras,
                      gripper_loc_bounds=a.gripper_loc_bounds,
                      image_rescale=tuple(float(x) for x in a.image_rescale.split(",")),
                      point_cloud_rotate_yaw_range=a.point_cloud_rotate_yaw_range, return_low_lvl_trajectory=False,
                      dense_interpolation=False, interpolation_length=0, action_dim=8, predict_short=False)
        train = D.RLBenchDataset(root=a.dataset, cache_size=a.cache_size, num_iters=a.train_iters, training=True, **common)
        test = D.RLBenchDataset(root=a.valset, cache_size=a.cache_size_val, training=False, **common)
        return train, test

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
                     regress_position_offset=bool(a.regress_position_offset), use_instruction=bool(a.use_instruction))

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
        """main_keypose.py:207-234"""
        loss = E.train_one_step(model, criterion, optimizer, step_id, sample, ddp=self.ddp,
                                accumulate_grad_batches=self.args.accumulate_grad_batches,
                                use_ground_truth_position_for_sampling_train=bool(
                                    self.args.use_ground_truth_position_for_sampling_train))
        if get_rank() == 0 and (step_id + 1) % self.args.val_freq == 0:
            self.writer.add_scalar("lr", self.args.lr, step_id)
            self.writer.add_scalar("train-loss/noise_mse", loss, step_id)
        return loss

    @torch.no_grad()
    def evaluate_nsteps(self, model, criterion, loader, step_id, val_iters, split='val'):
        """main_keypose.py:236-281: free-running forward (no ground-truth anchor for the ghost points), metrics per batch,
        mean over batches.  Returns values.get('val-losses/action_mse') -- a key compute_metrics never produces, so (as in
        the reference) None: every validation round overwrites best.pth."""
        values = {}
        device = next(model.parameters()).device
        model.eval()
        for i, sample in enumerate(loader):
            if i == val_iters:
                break
            action = model(sample["rgbs"], sample["pcds"], sample["instr"], sample["curr_gripper"], gt_action=None)
            for n, l in criterion.compute_metrics(action, sample).items():
                _append(values, f"{split}-losses/{n}", l, device)
        values = {k: torch.as_tensor(v).mean().item() for k, v in values.items()}
        self._log(values, step_id)
        return values.get('val-losses/action_mse', None)


class TrajectoryTrainTester(BaseTrainTester):
    """main_trajectory.py:86-274 (`TrainTester`)"""

    def get_datasets(self):
        a = self.args
        instruction = load_instructions(a.instructions, tasks=a.tasks, variations=a.variations)
        if instruction is None:
            raise NotImplementedError()
        taskvar = [(task, var) for task, var_instr in instruction.items() for var in var_instr.keys()]
        common = dict(instructions=instruction, taskvar=taskvar, max_episode_length=a.max_episode_length,
                      max_episodes_per_task=a.max_episodes_per_task, cameras=a.cameras,
                      gripper_loc_bounds=a.gripper_loc_bounds,
                      image_rescale=tuple(float(x) for x in a.image_rescale.split(",")),
                      point_cloud_rotate_yaw_range=a.point_cloud_rotate_yaw_range, return_low_lvl_trajectory=True,
                      dense_interpolation=bool(a.dense_interpolation), interpolation_length=a.interpolation_length,
                      action_dim=a.action_dim, predict_short=False)
        train = D.RLBenchDataset(root=a.dataset, cache_size=a.cache_size, num_iters=a.train_iters, training=True, **common)
        test = D.RLBenchDataset(root=a.valset, cache_size=a.cache_size_val, training=False, **common)
        return train, test

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
                                diffusion_timesteps=a.diffusion_timesteps)

    @staticmethod
    def get_criterion():
        return TrajectoryCriterion()

    def train_one_step(self, model, criterion, optimizer, step_id, sample):
        """main_trajectory.py:177-204"""
        loss = E.train_one_step_trajectory(model, criterion, optimizer, step_id, sample, ddp=self.ddp,
                                           accumulate_grad_batches=self.args.accumulate_grad_batches)
        if get_rank() == 0 and (step_id + 1) % self.args.val_freq == 0:
            self.writer.add_scalar("lr", self.args.lr, step_id)
            self.writer.add_scalar("train-loss/noise_mse", loss, step_id)
        return loss

    @torch.no_grad()
    def evaluate_nsteps(self, model, criterion, loader, step_id, val_iters, split='val'):
        """main_trajectory.py:206-274: 100-step sampling per batch (run_inference=True), summary metrics and per-task means,
        gathered over ranks.  Returns the mean 'val-losses/traj_action_mse'.  (The tensorboard trajectory plots of
        generate_visualizations need matplotlib and are out of scope.)"""
        values = {}
        device = next(model.parameters()).device
        model.eval()
        for i, sample in enumerate(loader):
            if i == val_iters:
                break
            s = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in sample.items()}
            action = model(s["trajectory"], s["trajectory_mask"], s["rgbs"], s["pcds"], s["instr"], s["curr_gripper"],
                           s["action"], run_inference=True)
            losses, losses_B = criterion.compute_metrics(action, s["trajectory"], s["trajectory_mask"])
            for n, l in losses.items():
                _append(values, f"{split}-losses/{n}", l, device)
            tasks = np.array(sample["task"])
            for n, l in losses_B.items():
                for task in np.unique(tasks):
                    sel = torch.from_numpy(tasks == task).to(device)
                    _append(values, f"{split}-loss/{task}/{n}", l[sel].mean(), device)
        values = self.synchronize_between_processes(values)
        values = {k: v.mean().item() for k, v in values.items()}
        self._log(values, step_id, always=False)
        return values.get('val-losses/traj_action_mse', None)
