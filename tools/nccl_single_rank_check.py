import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from options_in_simulation_amd import build_config, _abi
from options_in_simulation_amd.vecenv import NativeEnv
from options_in_simulation_amd.dist import gather_stats, merge_stats
env = NativeEnv(build_config(variant="KAMMA_medium", num_envs=4096, seed=1), dev)
env.reset(1); env.rollout(120, _abi.POLICY_RANDOM)
st = env.stats()
g = gather_stats(st, dev)
dist.barrier()
print("nccl world=1 gather ok:", len(g), g[0].episodes, st.episodes, merge_stats(g)["episodes"] if isinstance(merge_stats(g), dict) else merge_stats(g))
dist.destroy_process_group()
