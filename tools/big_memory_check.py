import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from options_in_simulation_amd import _abi, build_config
from options_in_simulation_amd.vecenv import NativeEnv
from options_in_simulation_amd.replay import DeviceFIFOMemory
N, M = 65536, 6000
env = NativeEnv(build_config(variant="KA_DDPG_4D", num_envs=N, seed=1), "cuda:0")
env.reset(1)
mem = DeviceFIFOMemory(env, M)
print("memory bytes %.1f GB, states elements %.3e (> 2^31: %s)" % (env._replay_mem.numel() / 1e9, mem.tensors["states"].numel(), mem.tensors["states"].numel() > 2**31))
t0 = time.perf_counter()
for _ in range(6): env.rollout(1000, _abi.POLICY_RANDOM)
torch.cuda.synchronize(); print("6000 steps in %.2f s, cursor" % (time.perf_counter() - t0), mem.memory_index, mem.filled)
s, ns, d = mem.tensors["states"], mem.tensors["next_states"], mem.tensors["terminated"]
ok_chain = bool(torch.equal(s[5999], ns[5998])) and bool(torch.equal(s[3000], ns[2999]))
ok_last = bool(torch.equal(ns[5999], env.buffer(_abi.BUF_OBS)))
idx = torch.tensor([M * N - 1, M * N - N, 5461 * N + 7, 0], device="cuda")
got = env.replay_sample(idx)
ok_sample = bool(torch.equal(got["states"], s.reshape(-1, 6)[idx])) and bool(torch.equal(got["rewards"], mem.tensors["rewards"].reshape(-1, 1)[idx]))
(b,) = mem.sample(["states", "actions"], 4096)
print("chain", ok_chain, "last row", ok_last, "sample across 2^31", ok_sample, "batch", b[0].shape, "terminated rows", int(d.sum()), "finite", bool(torch.isfinite(s).all()))
