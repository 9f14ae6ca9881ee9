"""Where does a cfg3 train step spend its time?  Phase timings (wall, synchronised) + in-situ kernel events.
Diagnostic only: python scripts/diag_step.py [--dtype bf16]"""
import argparse, ctypes, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maed_amd import _lib as L
from maed_amd.ddp import FusedAdam, GradBucketer, ParamArena

ap = argparse.ArgumentParser(); ap.add_argument("--dtype", default="bf16"); ap.add_argument("--clips", type=int, default=8)
ap.add_argument("--miopen-benchmark", action="store_true")
args = ap.parse_args()
if args.miopen_benchmark:
    torch.backends.cudnn.benchmark = True
dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
dev = torch.device("cuda", 0)
bench.CFG["clips"] = args.clips
T0 = time.perf_counter()
def P(msg): print(f"[diag +{time.perf_counter()-T0:7.1f}s] {msg}", flush=True)
def timed(name, fn, n=3):
    for i in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        P(f"{name} #{i}: {1e3*(time.perf_counter()-t):9.2f} ms")
    return r
P("build model"); model = bench.build_model(dtype, dev).train()
arena = ParamArena(model); buck = GradBucketer(arena, model); opt = FusedAdam(arena, lr=1e-4, weight_decay=1e-5, bucketer=buck)
gen = torch.Generator().manual_seed(0)
clip = torch.randn(bench.CFG["clips"], 16, 3, 224, 224, generator=gen).to(dev)
img = clip.reshape(-1, 3, 224, 224)
tgt = bench.make_targets(bench.CFG["clips"], 16, dev, gen)
bb = model.encoder.patch_embed.backbone
with torch.no_grad():
    timed("backbone fwd (no grad)", lambda: bb(img))
    timed("patch_embed fwd (no grad)", lambda: model.encoder.patch_embed(img))
    tok = timed("encoder.forward_tokens (no grad)", lambda: model.encoder.forward_tokens(img, 16))
    timed("6 blocks fwd (no grad)", lambda: [blk(tok, 16) for blk in model.encoder.blocks][-1])
def bb_train():
    arena.zero_grad(); y = bb(img); y.float().square().mean().backward(); return y
timed("backbone fwd+bwd", bb_train)
tokg = tok.detach().clone().requires_grad_(True)
def ste_train():
    arena.zero_grad(); x = tokg
    for blk in model.encoder.blocks: x = blk(x, 16)
    x.square().mean().backward()
timed("6 blocks fwd+bwd", ste_train)
from maed_amd.loss import LossVideo
criterion = LossVideo(**bench.LOSS_W)
feat = model.encoder(img, seqlen=16).detach().requires_grad_(True)
def tail_train():
    arena.zero_grad(); out = model.decoder(feat, seqlen=16)
    out = {k: v.reshape(bench.CFG["clips"], 16, *v.shape[1:]) for k, v in out.items()}
    loss, _ = criterion(out, tgt, None); loss.backward(); return loss
timed("decoder tail + loss fwd+bwd", tail_train)
def step():
    opt.zero_grad(); loss, _ = criterion(model(clip), tgt, None); loss.backward(); opt.step(); return loss
timed("full train step", step, n=4)
# host enqueue time vs GPU time: if enqueueing a step takes as long as the GPU needs to run it, the step is host-bound
torch.cuda.synchronize(); t0 = time.perf_counter(); enq = []
for _ in range(10):
    t1 = time.perf_counter(); step(); enq.append(1e3 * (time.perf_counter() - t1))
t_enq = time.perf_counter() - t0; torch.cuda.synchronize(); t_all = time.perf_counter() - t0
P(f"10 back-to-back steps: host enqueue {1e3 * t_enq / 10:.2f} ms/step (per step: {' '.join(f'{e:.1f}' for e in enq)}), wall incl. drain {1e3 * t_all / 10:.2f} ms/step")
lib = L.lib(); lib.maed_prof_enable(1); step(); torch.cuda.synchronize()
nt = lib.maed_prof_ntags(); ms = (ctypes.c_double * nt)(); cnt = (ctypes.c_int * nt)(); lib.maed_prof_collect(ms, cnt); lib.maed_prof_enable(0)
names = ["attn_sp_fwd", "attn_tm_fwd", "gemm_qkv", "gemm_fc1", "gemm_fc2", "attn_sp_bwd", "attn_tm_bwd", "gemm_wgrad"]
P("in-situ per-launch us: " + json.dumps({n: round(1e3 * ms[i] / max(cnt[i], 1), 1) for i, n in enumerate(names)}))
P("in-situ total ms/step: " + json.dumps({n: round(ms[i], 2) for i, n in enumerate(names)}))
model.eval()
with torch.no_grad():
    timed("inference forward (cfg2)", lambda: model(clip))
