"""Where the host's time goes while it enqueues one train step (cProfile over a few steps of the bench's own step function).
    python scripts/host_profile.py [steps]      -> top functions by own time and by cumulative time"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda", 0)
    from maed_amd.ddp import FusedAdam, GradBucketer, ParamArena
    from maed_amd.loss import LossVideo
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
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    torch.autograd.set_multithreading_enabled(False)      # the backward's Python functions in THIS thread, so that cProfile sees them
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
        torch.cuda.synchronize()        # (so that enqueue time is not hidden behind a full queue)
    pr.disable()
    for key in ("tottime", "cumtime"):
        print(f"==== top by {key} ({steps} steps) ====")
        st = pstats.Stats(pr)
        st.sort_stats(key).print_stats(45)


if __name__ == "__main__":
    main()
