import os, sys, torch, time
sys.path.insert(0, "stvo-pl_amd/python")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"; os.environ["LOCAL_RANK"] = "0"
import torch.distributed as dist
from stvo_amd import shard
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
torch.cuda.set_device(0)
dist.barrier(); torch.cuda.synchronize()
print(shard.aggregate(dist, 512 * 20, 0.0337, device="cuda:0"))
print([p.shape for p in shard.gather_poses(dist, torch.eye(4).reshape(1, 16).numpy(), device="cuda:0")])
dist.barrier(); dist.destroy_process_group(); print("nccl single-rank path ok")
