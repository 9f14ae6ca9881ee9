"""Can two ranks of the library's own RCCL communicator (maed_comm_*) share ONE GPU?  (a world-2 test on the 1-GPU box.)
Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 scripts/rccl_two_ranks_one_gpu.py
Both ranks use cuda:0; gloo carries the unique id.  Prints what RCCL says; exit code 0 either way (it documents a property of RCCL, not of the library)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from maed_amd.ddp import RcclComm
dist.init_process_group("gloo")
rank = dist.get_rank()
torch.cuda.set_device(0)
try:
    comm = RcclComm()
    buf = torch.full((1 << 20,), float(rank + 1), device="cuda")
    comm.allreduce_async(buf); comm.wait(); torch.cuda.synchronize()
    print(f"rank {rank}: all-reduce over 2 ranks on one GPU -> {buf[0].item()} (want 3.0)", flush=True)
    comm.destroy()
except Exception as e:
    print(f"rank {rank}: {type(e).__name__}: {e}", flush=True)
dist.destroy_process_group()
