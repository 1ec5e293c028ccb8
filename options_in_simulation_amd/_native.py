"""ctypes binding of the product library (lib/libois_hip.so, C ABI of include/ois.h).

There is no CPU fallback: if the HIP library is missing or does not load, importing the
environment fails loudly with instructions to build it.
"""
from __future__ import annotations

import ctypes as C
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OIS_LIB_OVERRIDE") or os.path.join(_HERE, "lib", "libois_hip.so")  # override: kernel experiments only

# every symbol include/ois.h declares (tests/test_abi.py checks the exports against the header)
SYMBOLS = (
    "ois_abi_version", "ois_last_error", "ois_config_default", "ois_workspace_bytes", "ois_create",
    "ois_destroy", "ois_reset", "ois_step", "ois_rollout", "ois_buffer", "ois_randomize_params",
    "ois_episode_info", "ois_stats_fetch", "ois_env0_diag", "ois_state_bytes", "ois_save_state",
    "ois_load_state", "ois_replay_bytes", "ois_replay_attach", "ois_replay_buffer", "ois_replay_cursor",
    "ois_gae_scratch_bytes", "ois_gae", "ois_terrain_attach", "ois_height_scan", "ois_replay_sample",
    "ois_profile_begin", "ois_profile_end", "ois_minfield_bytes", "ois_push_robots", "ois_kernel_info",
    "ois_logs_enable", "ois_logs_fetch", "ois_outputs_copy", "ois_params_dirty",
)

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libois_hip.so (once). torch must be imported first by the caller when the env is
    used from PyTorch so that both share one HIP runtime (same libamdhip64 SONAME)."""
    global _lib
    if _lib is not None:
        return _lib
    # Load order matters: torch ships its own libamdhip64 (SONAME libamdhip64.so.7, found via
    # its RPATH). Importing torch first makes the dynamic linker resolve this library's
    # NEEDED libamdhip64.so.7 to that already-loaded runtime, so streams and device pointers
    # handed over by PyTorch belong to the same HIP runtime instance.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C {os.path.join(_HERE, 'csrc')}`. There is no CPU fallback.")
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {exc}") from exc
    if os.environ.get("OIS_LIB_OVERRIDE"):
        # kernel A/Bs against the library of an earlier commit (tools/ab_build.sh): entry points it does not have yet become
        # stubs that say so when called, instead of failing the load
        class _Tolerant:
            def __init__(self, lib): object.__setattr__(self, "_lib", lib)
            def __getattr__(self, name):
                try:
                    return getattr(self._lib, name)
                except AttributeError:
                    def missing(*_a, **_k):
                        raise NativeLibraryError(f"{LIB_PATH} (OIS_LIB_OVERRIDE) has no {name}")
                    stub = type("Stub", (), {"__call__": staticmethod(missing), "argtypes": None, "restype": None})()
                    object.__setattr__(self, name, stub)
                    return stub
        L = _Tolerant(L)
    vp, fp = C.c_void_p, C.POINTER(C.c_float)
    cfgp = C.POINTER(_abi.OisConfig)
    L.ois_abi_version.restype = C.c_int
    L.ois_last_error.restype = C.c_char_p
    L.ois_config_default.argtypes = [cfgp]
    L.ois_workspace_bytes.argtypes = [cfgp]
    L.ois_workspace_bytes.restype = C.c_size_t
    L.ois_create.argtypes = [cfgp, C.c_int, vp, C.c_size_t, C.POINTER(vp)]
    L.ois_destroy.argtypes = [vp]
    L.ois_destroy.restype = None
    L.ois_reset.argtypes = [vp, C.c_uint64, vp]
    L.ois_step.argtypes = [vp, vp, vp]
    L.ois_rollout.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    L.ois_buffer.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ois_randomize_params.argtypes = [vp, C.c_float, C.c_float, fp, C.c_int, vp]
    L.ois_episode_info.argtypes = [vp, vp]
    L.ois_push_robots.argtypes = [vp, vp]
    L.ois_stats_fetch.argtypes = [vp, C.POINTER(_abi.OisStats), C.c_int, vp]
    L.ois_env0_diag.argtypes = [vp, fp, vp]
    L.ois_logs_enable.argtypes = [vp, C.c_int]
    L.ois_params_dirty.argtypes = [vp]
    L.ois_logs_fetch.argtypes = [vp, vp, vp, vp]
    L.ois_outputs_copy.argtypes = [vp, vp, vp, vp, vp, vp]
    L.ois_state_bytes.argtypes = [vp]
    L.ois_state_bytes.restype = C.c_size_t
    L.ois_save_state.argtypes = [vp, vp, C.c_size_t, vp]
    L.ois_load_state.argtypes = [vp, vp, C.c_size_t, vp]
    L.ois_replay_bytes.argtypes = [vp, C.c_int]
    L.ois_replay_bytes.restype = C.c_size_t
    L.ois_replay_attach.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.ois_replay_buffer.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ois_replay_cursor.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ois_gae_scratch_bytes.argtypes = [C.c_int]
    L.ois_gae_scratch_bytes.restype = C.c_size_t
    L.ois_gae.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_double, C.c_double, C.c_int, vp, vp, vp,
                          C.c_size_t, vp]
    L.ois_replay_sample.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.ois_terrain_attach.argtypes = [vp, C.POINTER(_abi.OisTerrain), vp, vp, vp, C.POINTER(C.c_int32), vp]
    L.ois_height_scan.argtypes = [vp, vp, vp]
    L.ois_minfield_bytes.argtypes = [C.c_int, C.c_int]
    L.ois_minfield_bytes.restype = C.c_size_t
    L.ois_kernel_info.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.ois_profile_begin.argtypes = [vp]
    L.ois_profile_end.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
    if L.ois_abi_version() != _abi.ABI_VERSION:
        raise NativeLibraryError(
            f"ABI mismatch: library {L.ois_abi_version()} vs python {_abi.ABI_VERSION}; rebuild the library")
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(f"libois_hip call failed ({rc}): {load().ois_last_error().decode()}")
