import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from options_in_simulation_amd import _abi, build_config
from options_in_simulation_amd import terrain as tg
from options_in_simulation_amd.vecenv import NativeEnv
N = 65536
# terrain: 0 = plane, 1 / 2 = height field with that footprint stride (ois_terrain.footprint_stride)
for variant, terrain in (("KA_DDPG_4D", 0), ("KAMMA_eval", 0), ("KAMMA_eval", 1), ("KA_DDPG_4D", 1), ("KAMMA_eval", 2), ("KA_DDPG_4D", 2)):
    cfg = build_config(variant=variant, num_envs=N, seed=3)
    env = NativeEnv(cfg, "cuda:0")
    if terrain:
        T = tg.Terrain({"numLevels": 6, "numTerrains": 2}, N, seed=3)
        env.attach_terrain(T, T.sample_levels_types(N, rng=np.random.RandomState(3)), footprint_stride=terrain)
        env.randomize_params(0.5, 1.0)
    env.reset(3)
    bad = 0
    for chunk in range(30):
        env.rollout(100, _abi.POLICY_RANDOM)
        for bid in (_abi.BUF_POSE, _abi.BUF_QUAT, _abi.BUF_LINVEL, _abi.BUF_ANGVEL, _abi.BUF_WHEEL_VEL, _abi.BUF_OBS, _abi.BUF_REW,
                    _abi.BUF_SUMS_A, _abi.BUF_SUMS_B, _abi.BUF_LAMBDA_N):
            if not torch.isfinite(env.buffer(bid)).all():
                bad += 1
    st = env.stats()
    p = env.buffer(_abi.BUF_POSE)
    print(variant, ("plane", "terrain", "terrain, footprint stride 2")[terrain], "non-finite checks failed:", bad, "episodes", st.episodes, "mean len %.1f" % (st.length_sum / st.episodes),
          "mean return %.1f" % (st.return_sum / st.episodes), "max |xy| %.1f" % float(p[:, :2].abs().max()), "z range %.2f..%.2f" % (float(p[:, 2].min()), float(p[:, 2].max())))
    env.close()
