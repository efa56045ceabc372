"""Loss / metric objects with the reference's interface (main_keypose.py:295-482, main_trajectory.py:295-343); the
losses are fused HIP kernels (soft cross-entropy over ghost points, MSE), the metrics are tiny no-grad reductions."""
import numpy as np
import torch

from . import ops as O


class LossAndMetrics:
    def __init__(self, position_loss, rotation_parametrization, ground_truth_gaussian_spread,
                 position_prediction_only=False, compute_loss_at_all_layers=False, label_smoothing=0.0,
                 position_loss_coeff=1.0, position_offset_loss_coeff=10000.0, rotation_loss_coeff=10.0,
                 gripper_loss_coeff=1.0, symmetric_rotation_loss=False):
        assert position_loss in ["mse", "ce", "ce+mse"]
        if position_loss != "ce" or symmetric_rotation_loss or "quat" not in rotation_parametrization:
            raise NotImplementedError("only position_loss='ce', quaternion rotation and symmetric_rotation_loss=False "
                                      "(the shipped training configuration) are implemented")
        self.position_loss = position_loss
        self.rotation_parametrization = rotation_parametrization
        self.position_prediction_only = position_prediction_only
        self.compute_loss_at_all_layers = compute_loss_at_all_layers
        self.ground_truth_gaussian_spread = ground_truth_gaussian_spread
        self.label_smoothing = label_smoothing
        self.position_loss_coeff = position_loss_coeff
        self.position_offset_loss_coeff = position_offset_loss_coeff
        self.rotation_loss_coeff = rotation_loss_coeff
        self.gripper_loss_coeff = gripper_loss_coeff
        self.symmetric_rotation_loss = symmetric_rotation_loss

    def compute_loss(self, pred, sample):
        device = pred["position"].device
        gt_action = sample["action"].to(device).float()
        losses = {}
        levels = len(pred["ghost_pcd_masks_pyramid"])
        gt_pos = gt_action[:, :3].contiguous()
        # soft cross-entropy with a Gaussian label around the ground truth (main_keypose.py:382-405); the reference
        # re-assigns the same dictionary key for every layer, so only the last decoder layer is supervised
        for i, (ghost, masks) in enumerate(zip(pred["ghost_pcd_pyramid"], pred["ghost_pcd_masks_pyramid"])):
            losses[f"position_ce_level{i}"] = O.SoftCEFn.apply(
                masks[-1], ghost.transpose(1, 2), gt_pos, float(self.ground_truth_gaussian_spread),
                float(self.label_smoothing), float(self.position_loss_coeff) / levels)
        pred["position"] = pred["position"].detach()
        losses["rotation"] = O.ElemLossFn.apply(pred["rotation"], gt_action[:, 3:7], 0, float(self.rotation_loss_coeff))
        losses["gripper"] = O.ElemLossFn.apply(pred["gripper"], gt_action[:, 7:8], 0, float(self.gripper_loss_coeff))
        return losses

    @torch.no_grad()
    def compute_metrics(self, pred, sample):
        device = pred["position"].device
        dtype = pred["position"].dtype
        outputs = sample["action"].to(device).float()
        metrics = {}
        tasks = np.array(sample["task"])
        final_pos_l2 = ((pred["position"] - outputs[:, :3]) ** 2).sum(1).sqrt()
        metrics["mean/pos_l2_final"] = final_pos_l2.to(dtype).mean()
        metrics["mean/pos_l2_final<0.01"] = (final_pos_l2 < 0.01).to(dtype).mean()
        for i in range(len(pred["position_pyramid"])):
            pos_l2_i = ((pred["position_pyramid"][i].squeeze(1) - outputs[:, :3]) ** 2).sum(1).sqrt()
            metrics[f"mean/pos_l2_level{i}"] = pos_l2_i.to(dtype).mean()
        for task in np.unique(tasks):
            sel = torch.from_numpy(tasks == task).to(device)
            metrics[f"{task}/pos_l2_final"] = final_pos_l2[sel].to(dtype).mean()
            metrics[f"{task}/pos_l2_final<0.01"] = (final_pos_l2[sel] < 0.01).to(dtype).mean()
        acc = (pred["gripper"] > 0.5).squeeze(-1) == outputs[:, 7].bool()
        metrics["gripper"] = acc.to(dtype).mean()
        l1 = (pred["rotation"] - outputs[:, 3:7]).abs().sum(1)
        metrics["mean/rot_l1"] = l1.to(dtype).mean()
        metrics["mean/rot_l1<0.05"] = (l1 < 0.05).to(dtype).mean()
        metrics["mean/rot_l1<0.025"] = (l1 < 0.025).to(dtype).mean()
        for task in np.unique(tasks):
            sel = torch.from_numpy(tasks == task).to(device)
            metrics[f"{task}/rot_l1"] = l1[sel].to(dtype).mean()
            metrics[f"{task}/rot_l1<0.05"] = (l1[sel] < 0.05).to(dtype).mean()
            metrics[f"{task}/rot_l1<0.025"] = (l1[sel] < 0.025).to(dtype).mean()
        return metrics


class TrajectoryCriterion:
    """main_trajectory.py:295-343: the training loss is computed inside DiffusionPlanner.forward."""

    def compute_loss(self, pred, gt=None, mask=None, is_loss=True):
        if not is_loss:
            assert gt is not None and mask is not None
            return self.compute_metrics(pred, gt, mask)[0]['action_mse']
        return pred

    @staticmethod
    @torch.no_grad()
    def compute_metrics(pred, gt, mask):
        pos_l2 = ((pred[..., :3] - gt[..., :3]) ** 2).sum(-1).sqrt()
        quat_l1 = (pred[..., 3:7] - gt[..., 3:7]).abs().sum(-1)
        quat_l1_ = (pred[..., 3:7] + gt[..., 3:7]).abs().sum(-1)
        sel = (quat_l1 < quat_l1_).float()
        quat_l1 = sel * quat_l1 + (1 - sel) * quat_l1_
        tr = 'traj_'
        ret_1 = {tr + 'action_mse': ((pred - gt) ** 2).mean(), tr + 'pos_l2': pos_l2.mean(),
                 tr + 'pos_acc_001': (pos_l2 < 0.01).float().mean(), tr + 'rot_l1': quat_l1.mean(),
                 tr + 'rot_acc_0025': (quat_l1 < 0.025).float().mean()}
        ret_2 = {tr + 'pos_l2': pos_l2.mean(-1), tr + 'pos_acc_001': (pos_l2 < 0.01).float().mean(-1),
                 tr + 'rot_l1': quat_l1.mean(-1), tr + 'rot_acc_0025': (quat_l1 < 0.025).float().mean(-1)}
        pos_l2 = ((pred[:, -1, :3] - gt[:, -1, :3]) ** 2).sum(-1).sqrt()
        quat_l1 = (pred[:, -1, 3:7] - gt[:, -1, 3:7]).abs().sum(-1)
        quat_l1_ = (pred[:, -1, 3:7] + gt[:, -1, 3:7]).abs().sum(-1)
        sel = (quat_l1 < quat_l1_).float()
        quat_l1 = sel * quat_l1 + (1 - sel) * quat_l1_
        ret_1.update({'pos_l2': pos_l2.mean(), 'pos_acc_001': (pos_l2 < 0.01).float().mean(), 'rot_l1': quat_l1.mean(),
                      'rot_acc_0025': (quat_l1 < 0.025).float().mean()})
        return ret_1, ret_2
