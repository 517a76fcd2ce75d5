"""Child script for tests/test_launch_cpu.py: every rank joins a gloo group, all-reduces (rank + 1) and rank 0 writes
the sum and the world size to the file named on the command line."""
import os
import sys
import torch
import torch.distributed as dist

dist.init_process_group("gloo")
t = torch.tensor([float(dist.get_rank() + 1)])
dist.all_reduce(t)
if dist.get_rank() == 0:
    with open(sys.argv[1], "w") as f:
        f.write(f"{int(t.item())} {dist.get_world_size()} {os.environ['MASTER_ADDR']}")
dist.barrier()
dist.destroy_process_group()
