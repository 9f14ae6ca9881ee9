"""Repeat the round-5 LDS-DMA weight-gradient kernel (csrc/gemm_tn2.hip: four-stage ring, counted waits, one barrier per tile) at STE and backbone shapes -- incl. a
column tile that ends inside a 128-block and strided operands -- and one twin-mode STE block (fp32 forward + cast pass on a side stream + bf16 backward) many times with
a second stream keeping the GPU busy, comparing every result with an fp32 reference / the first run: a missing wait or barrier in the copy pipeline shows up as an
occasional outlier, not in a single parity run.   usage: stress_r5_kernels.py [rounds]"""
import os, sys, torch
from functools import partial
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maed_amd
from maed_amd import ops, _lib as L
from maed_amd.vision_transformer import Block
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = "cuda"
torch.manual_seed(0)
shapes = [(25216, 1536, 512), (25216, 512, 2048), (100352, 512, 128), (401408, 64, 256), (25088, 1000, 264), (6400, 520, 72)]
ops_ = []
for m, n, k in shapes:
    Y, X = torch.randn(m, n, device=dev).bfloat16(), torch.randn(m, k, device=dev).bfloat16()
    ops_.append((Y, X, Y.float().t() @ X.float(), Y.float().sum(0)))
Yw, Xw = torch.randn(25216, 3 * 512, device=dev).bfloat16(), torch.randn(25216, 2048, device=dev).bfloat16()
ops_.append((Yw[:, 512:1024], Xw[:, 256:1280], Yw[:, 512:1024].float().t() @ Xw[:, 256:1280].float(), Yw[:, 512:1024].float().sum(0)))
# one STE block in the twin mode
maed_amd.set_float32_matmul_precision("bf16x3"); maed_amd.set_float32_backward_precision("bf16")
blk = Block(512, 8, mlp_ratio=4, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), st_mode="parallel", compute_dtype=torch.float32).to(dev)
xb = torch.randn(32, 197, 512, device=dev); dyb = torch.randn(32, 197, 512, device=dev)
busy_a = torch.randn(4096, 4096, device=dev).bfloat16(); busy_b = torch.randn(4096, 4096, device=dev).bfloat16()
side = torch.cuda.Stream()


def run_block():
    for p in blk.parameters():
        p.grad = None
    xg = xb.clone().requires_grad_(True)
    y = blk(xg, 16)
    y.backward(dyb)
    torch.cuda.synchronize()
    return dict(y=y.detach().clone(), dx=xg.grad.clone(), dwq=blk.attn.qkv.weight.grad.clone(), dwf=blk.mlp.fc2.weight.grad.clone())


ref_blk = run_block()
worst, worst_b = 0.0, {k: 0.0 for k in ref_blk}
for it in range(rounds):
    with torch.cuda.stream(side):
        for _ in range(6):
            busy_a @ busy_b
    for Y, X, ref, refb in ops_:
        dW = torch.zeros(ref.shape, device=dev); db = torch.zeros(ref.shape[0], device=dev)
        ops.gemm_tn_wgrad(Y, X, dW=dW, dbias=db)
        e = max(float((dW - ref).abs().max() / ref.abs().max()), float((db - refb).abs().max() / refb.abs().max()))
        worst = max(worst, e)
        assert e < 1e-4, (it, tuple(ref.shape), e)
    got = run_block()
    for k in ref_blk:
        d = float((got[k] - ref_blk[k]).abs().max() / (ref_blk[k].abs().max() + 1e-30))
        worst_b[k] = max(worst_b[k], d)
        assert torch.isfinite(got[k]).all() and d < 2e-3, (it, k, d)
torch.cuda.synchronize()
print(f"{rounds} rounds under a busy second stream: weight-gradient GEMM worst relative error vs the fp32 product {worst:.2e} (7 shapes);")
print("twin-mode STE block, worst relative deviation from the first run: " + ", ".join(f"{k} {v:.2e}" for k, v in worst_b.items()))
print("device faults:", L.device_faults())
print("OK")
