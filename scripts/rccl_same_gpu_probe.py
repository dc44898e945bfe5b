"""can two RCCL ranks share ONE GPU on this box? (test-infrastructure probe)"""
import os, sys, torch, torch.distributed as dist
rank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group('nccl')
x = torch.full((4,), float(rank + 1), device='cuda', dtype=torch.float64)
dist.all_reduce(x)
torch.cuda.synchronize()
print('rank', rank, 'allreduce ->', x.tolist(), flush=True)
dist.destroy_process_group()
