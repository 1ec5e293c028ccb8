"""MI355X-native vectorised skid-steer torque-control environment.

Drop-in for the reset()/step() hot path of meesjansen/Options_In_Simulation
(src/my_envs/KAMMA.py and variants): one fused HIP kernel per policy step for N arenas,
reached through the C ABI of include/ois.h. See DESIGN.md.
"""
from . import _abi
from .config import TASK_CFG, VARIANTS, build_config, default_task_cfg

__all__ = ["TASK_CFG", "VARIANTS", "build_config", "default_task_cfg", "VecEnv", "TorqueDistributionTask",
           "NativeEnv", "DeviceFIFOMemory", "compute_gae", "compute_advantages_and_returns", "_abi"]


def __getattr__(name):  # torch / the native library are only needed for the env classes
    if name in ("VecEnv", "TorqueDistributionTask", "NativeEnv"):
        from . import vecenv
        return getattr(vecenv, name)
    if name == "DeviceFIFOMemory":
        from .replay import DeviceFIFOMemory
        return DeviceFIFOMemory
    if name in ("compute_gae", "compute_advantages_and_returns"):
        from . import gae
        return getattr(gae, name)
    raise AttributeError(name)
