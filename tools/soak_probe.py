import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from options_in_simulation_amd import _abi, build_config
from options_in_simulation_amd import terrain as tg
from options_in_simulation_amd.vecenv import NativeEnv
N = 65536
# terrain: 0 = plane, -1 = plane with per-arena friction U(0.5, 1) and slopes {-0.3 ... 0.3} (BASELINE.json configs[3]), 1 / 2 = height
# field with that footprint stride (ois_terrain.footprint_stride). `tilt` = the largest angle between body z and the ground normal seen
# at the end of a chunk (a robot on its side, 90 degrees, is the pose SPEC 3.9 is about)
for variant, terrain in (("KA_DDPG_4D", 0), ("KAMMA_eval", 0), ("KAMMA_eval", -1), ("KAMMA_eval", 1), ("KA_DDPG_4D", 1), ("KAMMA_eval", 2), ("KA_DDPG_4D", 2)):
    cfg = build_config(variant=variant, num_envs=N, seed=3)
    env = NativeEnv(cfg, "cuda:0")
    if terrain < 0:
        env.randomize_params(0.5, 1.0, [-0.3, -0.2, 0.0, 0.2, 0.3])
    if terrain > 0:
        T = tg.Terrain({"numLevels": 6, "numTerrains": 2}, N, seed=3)
        env.attach_terrain(T, T.sample_levels_types(N, rng=np.random.RandomState(3)), footprint_stride=terrain)
        env.randomize_params(0.5, 1.0)
    env.reset(3)
    bad, tilt, fastest = 0, 0.0, 0.0
    for chunk in range(30):
        env.rollout(100, _abi.POLICY_RANDOM)
        q, prm = env.buffer(_abi.BUF_QUAT), env.buffer(_abi.BUF_PARAMS)
        # body z axis in the world (third column of R) against the plane normal of `params` (the field's own relief aside)
        zx = 2 * (q[:, 1] * q[:, 3] + q[:, 0] * q[:, 2]); zy = 2 * (q[:, 2] * q[:, 3] - q[:, 0] * q[:, 1]); zz = 1 - 2 * (q[:, 1] ** 2 + q[:, 2] ** 2)
        c = (zx * prm[:, 1] + zy * prm[:, 2] + zz * prm[:, 3]).clamp(-1, 1)
        tilt = max(tilt, float(torch.rad2deg(torch.acos(c)).max()))
        fastest = max(fastest, float(env.buffer(_abi.BUF_LINVEL)[:, :3].abs().max()))
        for bid in (_abi.BUF_POSE, _abi.BUF_QUAT, _abi.BUF_LINVEL, _abi.BUF_ANGVEL, _abi.BUF_WHEEL_VEL, _abi.BUF_OBS, _abi.BUF_REW,
                    _abi.BUF_SUMS_A, _abi.BUF_SUMS_B, _abi.BUF_LAMBDA_N):
            if not torch.isfinite(env.buffer(bid)).all():
                bad += 1
    st = env.stats()
    p = env.buffer(_abi.BUF_POSE)
    print(variant, {0: "plane", -1: "plane, friction + slopes", 1: "terrain", 2: "terrain, footprint stride 2"}[terrain], "non-finite checks failed:", bad, "episodes", st.episodes, "mean len %.1f" % (st.length_sum / st.episodes),
          "mean return %.1f" % (st.return_sum / st.episodes), "max |xy| %.1f" % float(p[:, :2].abs().max()), "z range %.2f..%.2f" % (float(p[:, 2].min()), float(p[:, 2].max())), "largest tilt %.1f deg" % tilt, "fastest %.1f m/s" % fastest)
    env.close()
