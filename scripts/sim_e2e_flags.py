"""Two whole train steps of a tiny MAED on the HOST SIMULATOR (no GPU): run it once plainly and once with the opt-in switches in the
environment; the printed loss trajectory must agree (first step identical, second within bf16 / summation-order noise).

    python scripts/sim_e2e_flags.py
    MAED_WS_PER_STAGE=1 MAED_GN_FUSE_STATS=0 MAED_GN_LAZY_DRES=0 MAED_CONV3X3=miopen python scripts/sim_e2e_flags.py

(switches that still exist: README.md's table; the round-2 in-kernel switches MAED_GN_DEFER_AFFINE / MAED_LN_DEFER_AFFINE / MAED_TAIL_PARALLEL are gone with the
variants that lost -- setting them changes nothing)

Round-1 tree: [171.142578125, 162.39 +- 0.01] either way (atomics make the second step vary in the 5th digit from run to run)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ["MAED_SLOW_TESTS"]="1"
import test_hostsim_e2e as E
from _hostsim import patched
from maed_amd.ddp import FusedAdam, GradBucketer, ParamArena
from maed_amd.loss import Loss
from maed_amd.trainer import TrainStep
torch.manual_seed(0)
g = torch.Generator().manual_seed(1)
model = E.TinyMAED(torch.bfloat16).train()
model.decoder.drop1.p = model.decoder.drop2.p = 0.0      # deterministic comparison between flag sets
N, T = 2, 2
clip = torch.randn(N, T, 3, 32, 32, generator=g)
tgt = dict(images=clip, kp_2d=torch.cat([torch.randn(N, T, 49, 2, generator=g) * 0.3, torch.rand(N, T, 49, 1, generator=g)], -1),
           kp_3d=torch.cat([torch.randn(N, T, 49, 3, generator=g) * 0.3, torch.ones(N, T, 49, 1)], -1),
           theta=torch.cat([torch.randn(N, T, 3, generator=g) * 0.1, torch.randn(N, T, 72, generator=g) * 0.2, torch.randn(N, T, 10, generator=g)], -1),
           w_smpl=torch.ones(N, T))
with patched():
    arena = ParamArena(model, device=torch.device("cpu"))
    bucketer = GradBucketer(arena, model, bucket_bytes=256 << 10)
    opt = FusedAdam(arena, lr=1e-3, bucketer=bucketer, model=model)
    step = TrainStep(model, Loss(300.0, 600.0, 60.0, 0.06, 1.0, 0.0, device="cpu"), opt)
    tot=[]
    for _ in range(2):
        total, terms = step(target_3d=tgt)
        tot.append(float(total.detach()))
print("TOTALS", tot, "param checksum", float(arena.flat.double().abs().sum()))
