"""The library's own communicator (csrc/comm.hip: maed_comm_load / unique_id / init / allreduce_async / wait / destroy -- train.py:113,182's DistributedDataParallel
all-reduce) driven from TWO processes on a box without GPUs.  RCCL itself cannot do that (it refuses two ranks on one device and has no CPU
transport), and comm.hip binds the NCCL API by dlsym from whatever library the host names -- so the test names tests/hostsim/libfakerccl.so, a shared-memory
implementation of the five entry points, and runs comm.hip itself (x86 build of the same source, streams / events as no-ops) through the product's ctypes table,
ddp.RcclComm and ddp.GradBucketer: the reduced gradient / world must equal the single-process gradient of the concatenated batch."""
import os
import sys

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


class _Toy(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = nn.Linear(16, 32)
        self.b = nn.Linear(32, 32)
        self.c = nn.Linear(32, 4)

    def forward(self, x):
        return self.c(torch.tanh(self.b(torch.tanh(self.a(x)))))


def _batch():
    return torch.randn(8, 16, generator=torch.Generator().manual_seed(42)), torch.randn(8, 4, generator=torch.Generator().manual_seed(43))


def _worker(rank, world, port, fake, out):
    import torch.distributed as dist
    from _hostsim import patched
    from maed_amd import _lib as L
    from maed_amd.ddp import GradBucketer, ParamArena, RcclComm
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)          # carries the 128-byte unique id and the parameter broadcast only
    try:
        with patched() as sim:
            comm = RcclComm(lib_path=fake)                                # maed_comm_load(fake) -> unique_id on rank 0 -> broadcast -> maed_comm_init
            assert sim.maed_comm_world() == world
            model = _Toy()
            if rank == 1:
                with torch.no_grad():
                    for p in model.parameters():
                        p.add_(1.0)
            arena = ParamArena(model, device=torch.device("cpu"))
            bucketer = GradBucketer(arena, model, bucket_bytes=2048, comm=comm)     # several buckets -> several maed_comm_allreduce_async calls per backward
            bucketer.broadcast_parameters(0)
            assert len(bucketer.buckets) > 1 and bucketer.collectives
            x, y = _batch()
            per = 8 // world
            shard = slice(rank * per, rank * per + per)
            for _ in range(2):
                arena.zero_grad()
                loss = ((model(x[shard]) - y[shard]) ** 2).mean() + 0.5 * ((model(x[shard] * 2) - y[shard]) ** 2).mean()
                loss.backward()
                bucketer.finish()                                          # maed_comm_wait
            # a large bf16 buffer too: several 1-MiB chunks of the stand-in, the dtype code path of maed_comm_allreduce_async
            big = (torch.arange(700_000, dtype=torch.float32) % 251 * (rank + 1)).bfloat16()
            comm.allreduce_async(big)
            comm.wait()
            out[rank] = ((arena.grad / world).clone(), arena.flat.clone(), big.float().sum().item())
            comm.destroy()
            assert sim.maed_comm_world() == 0
            # a second communicator in the same process (new unique id): destroy really released the first
            comm2 = RcclComm(lib_path=fake)
            t = torch.full((5,), float(rank + 1))
            comm2.allreduce_async(t)
            comm2.wait()
            assert torch.equal(t, torch.full((5,), world * (world + 1) / 2.0))
            comm2.destroy()
    finally:
        dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("world", [2, 4])
def test_library_communicator_two_processes_matches_single_process_gradient(world):
    """world = 4 (round 5): four ranks through the same communicator code -- rank-ordered reduction, several buckets per backward, bf16 payload"""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(HERE, "hostsim"))
    import build_sim
    build_sim.build()
    fake = build_sim.build_fakerccl()
    port = 29633 + (os.getpid() + 7 * world) % 1000
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, fake, out), nprocs=world, join=True)
    (g0, p0, s0) = out[0]
    for r in range(1, world):
        g1, p1, s1 = out[r]
        assert torch.equal(p0, p1), "parameters must be identical after the broadcast"
        assert torch.equal(g0, g1), "every rank must hold the same reduced gradient"
        assert s1 == s0
    from maed_amd.ddp import ParamArena
    model = _Toy()
    arena = ParamArena(model, device=torch.device("cpu"))
    x, y = _batch()
    (((model(x) - y) ** 2).mean() + 0.5 * ((model(x * 2) - y) ** 2).mean()).backward()
    torch.testing.assert_close(g0, arena.grad, rtol=1e-5, atol=1e-6)
    base = torch.arange(700_000, dtype=torch.float32) % 251
    acc = torch.zeros_like(base)
    for r in range(world):                      # the stand-in sums the ranks' bf16 values in rank order in fp32 and rounds once
        acc += (base * (r + 1)).bfloat16().float()
    assert s0 == acc.bfloat16().float().sum().item()


def test_comm_load_refuses_a_library_without_the_nccl_api(tmp_path):
    from _hostsim import patched
    with patched() as sim:
        if sim.maed_comm_world() != 0:
            return
        rc = sim.maed_comm_load(b"libm.so.6")
        # (once a library is bound in this process maed_comm_load is a no-op: only assert on a fresh handle)
        assert rc in (0, -5)
        if rc != 0:
            assert b"NCCL API" in sim.maed_last_error()
