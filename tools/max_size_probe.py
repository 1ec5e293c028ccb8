"""The largest environments ois_create accepts (include/ois.h: up to 2^27 arenas per GPU): a few steps at 2^24 and 2^27 arenas
(4.9 GB / 39 GB of workspace), checked for index arithmetic that would wrap at 32 bits — a strided sample of arenas against
1-arena oracle environments at the same global ids, counters, finiteness, the logging reductions (two launches above 131 072
arenas) and the episode statistics (GPU box)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from options_in_simulation_amd import _abi, build_config  # noqa: E402
from options_in_simulation_amd.vecenv import NativeEnv  # noqa: E402
from oracle.binding import OracleEnv  # noqa: E402

for logN in (24, 27):
    N = 1 << logN
    K = 3
    kw = dict(variant="KAMMA_medium", seed=21, max_sim_steps=60.0)
    cfg = build_config(num_envs=N, **kw)
    cfg.max_episode_length = 3   # every arena times out at step 2: the reset path and its records at full size
    t0 = time.time()
    env = NativeEnv(cfg, "cuda:0")
    env.reset(21)
    env.rollout(K, _abi.POLICY_RANDOM)
    torch.cuda.synchronize()
    t1 = time.time()
    cnt = env.buffer(_abi.BUF_COUNTERS)
    assert int(cnt[:, 1].min()) == K and int(cnt[:, 1].max()) == K, "sim_steps"
    for bid in (_abi.BUF_OBS, _abi.BUF_REW, _abi.BUF_POSE, _abi.BUF_QUAT, _abi.BUF_LINVEL):
        assert bool(torch.isfinite(env.buffer(bid)).all()), bid
    ids = sorted(set([0, 1, 63, 64, N // 3, N // 2 - 1, N // 2, N // 2 + 1, (1 << 23) + 5, N - 65, N - 2, N - 1]))
    pose = env.buffer(_abi.BUF_POSE)[ids].cpu().numpy()
    obs = env.buffer(_abi.BUF_OBS)[ids].cpu().numpy()
    cts = cnt[ids].cpu().numpy()
    for j, i in enumerate(ids):
        c1 = build_config(num_envs=1, arena_offset=i, **kw)
        c1.max_episode_length = 3
        o = OracleEnv(c1)
        o.reset(21)
        o.rollout(K, _abi.POLICY_RANDOM)
        assert np.array_equal(pose[j], o.buffer(_abi.BUF_POSE)[0]) and np.array_equal(obs[j], o.obs[0]), f"arena {i}"
        assert np.array_equal(cts[j], o.buffer(_abi.BUF_COUNTERS)[0]), f"counters of arena {i}"
    ei = env.episode_info().cpu().numpy()
    st = env.stats()
    assert float(ei[_abi.EI_COUNT]) == N, ei[_abi.EI_COUNT]
    assert st.episodes == float(N) and st.steps == float(N) * K, (st.episodes, st.steps)
    logs = env.host_logs()[0]
    assert np.array_equal(logs, ei, equal_nan=True)
    scan_ok = ""
    print(f"2^{logN} = {N} arenas: {K} steps incl. one reset of every arena, {len(ids)} sampled arenas equal their 1-arena oracle runs bit for bit, "
          f"extras count {int(ei[_abi.EI_COUNT])}, statistics {int(st.episodes)} episodes; create + reset + rollout {t1 - t0:.2f} s, "
          f"workspace {env.workspace.numel() / 2**30:.1f} GiB", flush=True)
    env.close()
    del env
    torch.cuda.empty_cache()
