"""act3d-chained-diffuser_amd: MI355X-native hot path of Act3D / ChainedDiffuser.

Import with ``importlib.import_module("act3d-chained-diffuser_amd")`` (the directory name carries a hyphen).
Nothing here computes on the CPU: every op needs libact3d_hip.so and a gfx950 device.
"""
from . import lib  # noqa: F401
from .build import build  # noqa: F401
from . import ops  # noqa: F401,E402
from . import nn, act3d, losses  # noqa: F401,E402
from .act3d import Act3D  # noqa: F401,E402
from .losses import LossAndMetrics, TrajectoryCriterion  # noqa: F401,E402
from . import engine  # noqa: F401,E402
from . import diffusion  # noqa: F401,E402
from .diffusion import DiffusionPlanner, DiffusionHead  # noqa: F401,E402
from . import data, trainers  # noqa: F401,E402
from .trainers import BaseTrainTester, KeyposeTrainTester, TrajectoryTrainTester  # noqa: F401,E402
