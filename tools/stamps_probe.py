"""In-kernel timeline of one ois_step launch (diagnostic; needs a variant library built from an instrumented
copy of the kernel, OIS_LIB_OVERRIDE=<variant .so>): s_memrealtime (100 MHz) at wave entry, before and after
every substep, before the state stores. Medians over the 1024 waves, us."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from options_in_simulation_amd import build_config, _native
from options_in_simulation_amd.vecenv import NativeEnv
N = 65536
env = NativeEnv(build_config(variant="KAMMA_medium", num_envs=N, seed=42), "cuda:0")
L = _native.load()
st = torch.zeros((N // 64, 32), dtype=torch.int64, device="cuda")
L.ois_debug_set_stamps.argtypes = [C.c_void_p]
assert L.ois_debug_set_stamps(C.c_void_p(st.data_ptr())) == 0
env.reset(42)
a = torch.rand((N, 4), device="cuda")
sweep_src = torch.empty(96 << 20, dtype=torch.float32, device="cuda"); sweep_dst = torch.empty_like(sweep_src)
for label, sweep, pipelined in (("isolated, warm", False, False), ("isolated, after a cache sweep", True, False), ("third of three back-to-back", False, True)):
    rows = []
    for rep in range(12):
        if sweep:
            sweep_dst.copy_(sweep_src)
        if pipelined:
            env.step(a); env.step(a)
        env.profile_begin()
        env.step(a)
        ms, _ = env.profile_end()
        s = st.cpu().numpy().astype(np.float64) * 0.01  # ticks of 10 ns -> us
        t0 = s[:, 0].min()
        sub = np.median(np.diff(s[:, 1:22], axis=1), axis=0)  # 20 substeps
        rows.append([ms * 1e3, s[:, 0].max() - t0, np.median(s[:, 1] - s[:, 0]), *sub, np.median(s[:, 30] - s[:, 21]), s[:, 30].max() - t0])
    r = np.median(np.array(rows[2:]), axis=0)
    print(f"{label:30s} kernel {r[0]:.2f} | last wave starts +{r[1]:.2f} | entry->physics {r[2]:.2f} | substeps " + " ".join(f"{x:.2f}" for x in r[3:23])
          + f" | post-physics {r[23]:.2f} | last wave at its stores +{r[24]:.2f}")
