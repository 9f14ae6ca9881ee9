"""Upper bounds for next-round work (NOT product paths; gradients are wrong while a probe is on): the cfg3 bf16 train step with a kernel family switched off.
    python scripts/bound_probe.py [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from maed_amd import ops  # noqa: E402
from maed_amd.ddp import FusedAdam, GradBucketer, ParamArena  # noqa: E402
from maed_amd.loss import LossVideo  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda", 0)
    model = bench.build_model(torch.bfloat16, dev).train()
    arena = ParamArena(model)
    opt = FusedAdam(arena, lr=1e-4, weight_decay=1e-5, bucketer=GradBucketer(arena, model))
    crit = LossVideo(**bench.LOSS_W)
    gen = torch.Generator().manual_seed(0)
    C = bench.CFG
    clip = torch.randn(C["clips"], C["T"], 3, C["img"], C["img"], generator=gen).to(dev)
    tgt = bench.make_targets(C["clips"], C["T"], dev, gen)

    def step():
        opt.zero_grad()
        loss, _ = crit(model(clip), tgt, None)
        loss.backward()
        opt.step()

    def timed(tag):
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        ev[0].record()
        for i in range(steps):
            step()
            ev[i + 1].record()
        torch.cuda.synchronize()
        per = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
        print(f"{tag:70s} median {per[len(per) // 2]:7.3f} ms/step", flush=True)

    timed("baseline")
    real_tn, real_c3 = ops.gemm_tn_wgrad, ops.conv3x3_wgrad
    ops.gemm_tn_wgrad = lambda Y, X, dW=None, dbias=None, prec=None: dW if dW is not None else real_tn(Y, X, dW, dbias, prec)
    ops.conv3x3_wgrad = lambda dy, x, out=None, prec=None: out if out is not None else real_c3(dy, x, out, prec)
    timed("backbone weight-gradient GEMMs (1x1 + 3x3, 67 launches) switched off")
    ops.gemm_tn_wgrad, ops.conv3x3_wgrad = real_tn, real_c3
    side = ops._SIDE_ON
    ops._SIDE_ON = False
    timed("backbone weight gradients on the caller's stream (no side stream)")
    ops._SIDE_ON = side
    timed("baseline again")


if __name__ == "__main__":
    main()
