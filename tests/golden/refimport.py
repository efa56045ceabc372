"""Imports the read-only reference (/root/reference) in the BUILD container with stubs for its missing third-party
packages.  Used only by make_goldens.py; never on the GPU box (the reference does not travel).

Stubs (their arithmetic is third-party and therefore "parity unpinned"):
  torchvision.ops.FeaturePyramidNetwork -- restated from torchvision 0.14 (1x1 lateral convs, nearest top-down,
                                           3x3 output convs, Conv2dNormActivation naming `inner_blocks.i.0.weight`)
  diffusers DDPMScheduler               -- thin adapter over oracle.diffusion.DDPMSchedules
  clip / torchvision.models / tap / cv2 / datasets / engine -- inert placeholders
"""
import os
import sys
import types
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _FPN(nn.Module):
    def __init__(self, in_channels_list, out_channels):
        super().__init__()
        self.inner_blocks = nn.ModuleList([nn.Sequential(nn.Conv2d(c, out_channels, 1)) for c in in_channels_list])
        self.layer_blocks = nn.ModuleList(
            [nn.Sequential(nn.Conv2d(out_channels, out_channels, 3, padding=1)) for _ in in_channels_list])

    def forward(self, x):
        names, feats = list(x.keys()), list(x.values())
        last = self.inner_blocks[-1](feats[-1])
        outs = [self.layer_blocks[-1](last)]
        for i in range(len(feats) - 2, -1, -1):
            lat = self.inner_blocks[i](feats[i])
            last = lat + F.interpolate(last, size=lat.shape[-2:], mode="nearest")
            outs.insert(0, self.layer_blocks[i](last))
        return OrderedDict(zip(names, outs))


class _Normalize(nn.Module):
    def __init__(self, mean, std):
        super().__init__()
        self.mean, self.std = torch.tensor(mean).view(1, -1, 1, 1), torch.tensor(std).view(1, -1, 1, 1)

    def forward(self, x):
        return (x - self.mean) / self.std


class TinyBackbone(nn.Module):
    """Deterministic stand-in for CLIP-RN50: res1..res5 with the right channels and strides."""

    def __init__(self):
        super().__init__()
        chans, prev = [64, 256, 512, 1024, 2048], 3
        self.convs = nn.ModuleList()
        for c in chans:
            self.convs.append(nn.Conv2d(prev, c, 3, stride=2, padding=1))
            prev = c

    def forward(self, x):
        out = {}
        for i, conv in enumerate(self.convs):
            x = torch.tanh(conv(x))
            out[f"res{i + 1}"] = x
        return out


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle.diffusion import DDPMSchedules

    class DDPMScheduler:
        """diffusers.schedulers.scheduling_ddpm.DDPMScheduler surface used by diffusion_model.py:51-117,291-303."""

        def __init__(self, num_train_timesteps=100, beta_schedule="scaled_linear", prediction_type="sample"):
            assert prediction_type == "sample"
            self.config = types.SimpleNamespace(num_train_timesteps=num_train_timesteps)
            s = DDPMSchedules(num_train_timesteps)
            self.kind = beta_schedule
            self.acp = s.acp_pos if beta_schedule == "scaled_linear" else s.acp_rot
            self.coef = s.coef_pos if beta_schedule == "scaled_linear" else s.coef_rot
            self.timesteps = None
            self.injected_noise = None      # set by the golden generator: dict t -> tensor

        def set_timesteps(self, n):
            self.timesteps = torch.arange(n - 1, -1, -1)

        def add_noise(self, x, noise, t):
            sa = (self.acp[t] ** 0.5).view(-1, *([1] * (x.dim() - 1)))
            sb = ((1 - self.acp[t]) ** 0.5).view(-1, *([1] * (x.dim() - 1)))
            return sa * x + sb * noise

        def step(self, model_output, t, sample):
            t = int(t)
            x0 = model_output.clamp(-1.0, 1.0)
            prev = self.coef[t, 0] * x0 + self.coef[t, 1] * sample
            if t > 0:
                noise = self.injected_noise[t] if self.injected_noise is not None else torch.randn_like(sample)
                prev = prev + self.coef[t, 2] * noise
            return types.SimpleNamespace(prev_sample=prev)

    tv = _mod("torchvision")
    tv.ops = _mod("torchvision.ops", FeaturePyramidNetwork=_FPN)
    tv.transforms = _mod("torchvision.transforms", Normalize=_Normalize)
    tv.models = _mod("torchvision.models")
    tv.models.resnet = _mod("torchvision.models.resnet", _resnet=None, BasicBlock=nn.Module, Bottleneck=nn.Module,
                            ResNet=nn.Module)
    cl = _mod("clip", load=lambda *a, **k: None)
    cl.model = _mod("clip.model", ModifiedResNet=nn.Module)
    df = _mod("diffusers")
    df.schedulers = _mod("diffusers.schedulers")
    df.schedulers.scheduling_ddpm = _mod("diffusers.schedulers.scheduling_ddpm", DDPMScheduler=DDPMScheduler)
    _mod("tap", Tap=object)
    _mod("cv2")
    _mod("datasets", RLBenchDataset=object)
    _mod("engine", BaseTrainTester=object)


def import_reference():
    """Returns a namespace with the reference classes, backbone loaders patched to TinyBackbone."""
    if not os.path.isdir(REF):
        raise RuntimeError("the reference is only available in the build container")
    sys.dont_write_bytecode = True
    install_stubs()
    if REF not in sys.path:
        sys.path.insert(1, REF)
    import model.keypose_optimization.act3d as ref_act3d
    import model.utils.encoder as ref_encoder
    import model.utils.layers as ref_layers
    import model.utils.multihead_custom_attention as ref_mha
    import model.utils.position_encodings as ref_pe
    import model.utils.utils as ref_utils
    import model.trajectory_optimization.diffusion_model as ref_dm
    import model.trajectory_optimization.diffusion_head as ref_dh
    import utils.pytorch3d_transforms as ref_p3d
    import main_keypose as ref_main_keypose
    import main_trajectory as ref_main_trajectory

    def fake_clip():
        return TinyBackbone(), _Normalize([0.481, 0.457, 0.408], [0.268, 0.261, 0.275])

    ref_act3d.load_clip = fake_clip
    ref_encoder.load_clip = fake_clip
    return types.SimpleNamespace(act3d=ref_act3d, encoder=ref_encoder, layers=ref_layers, mha=ref_mha, pe=ref_pe,
                                 utils=ref_utils, dm=ref_dm, dh=ref_dh, p3d=ref_p3d, main_keypose=ref_main_keypose,
                                 main_trajectory=ref_main_trajectory)
