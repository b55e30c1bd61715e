import os, torch, torch.distributed as dist, numpy as np, sys
sys.path.insert(0, os.getcwd())
from bulletproofs_amd import sharded, ffi
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
ffi.init(0)
part = np.zeros(9, dtype=np.uint64); part[8] = 1
out = sharded.combine_partials(0, part, torch.device("cuda", 0))
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
mine = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64, device="cuda"); allr = [torch.empty_like(mine)]; dist.all_gather(allr, mine)
dist.barrier(); dist.destroy_process_group(); print("nccl single-rank ok", out.tolist(), t.item())
