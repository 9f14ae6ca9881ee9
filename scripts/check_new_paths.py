"""Parity of everything that was added after the round-1 GPU budget was spent, on the REAL library (run on the GPU box):
st_modes series/vanilla/temporal/coupling, decoder='iterative', the evaluation kernels + Evaluator, and the long-sequence
attention kernels.  The same comparisons run in the `-m "not gpu"` suite with the kernels on the host simulator
(tests/test_hostsim_{modes,iterative,eval,attention}.py); this script repeats them with cuda tensors.

    python scripts/check_new_paths.py            # on a GPU box
    python scripts/check_new_paths.py --sim      # self-test of this script on the host simulator (no GPU)

Every comparison is recorded (RESULTS) and printed; a mismatch does not stop the run -- the script exits non-zero at the end if any check failed, and
tests/test_gpu_paths.py reports each failing comparison by name."""
import contextlib
import os
import sys
from functools import partial

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
SIM = "--sim" in sys.argv
DEV = "cpu" if SIM else "cuda"
if SIM:
    from _hostsim import patched
else:
    patched = contextlib.nullcontext

from oracle import maed_ref as R                                              # noqa: E402
from oracle.make_golden_eval import StubModel                                 # noqa: E402
from maed_amd import _lib as L, eval_utils as EU, ops, ste_modes, tail        # noqa: E402
from maed_amd.evaluate import Evaluator                                       # noqa: E402
from maed_amd.smpl import SMPL                                                # noqa: E402
from maed_amd.spin import Regressor                                           # noqa: E402
from maed_amd.vision_transformer import Block                                # noqa: E402

LN = partial(nn.LayerNorm, eps=1e-6)
MODES = ["series", "vanilla", "temporal", "coupling"]


def golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))


def t(a):
    return torch.from_numpy(np.asarray(a)).to(DEV)


def rel(a, b):
    b = b.detach() if torch.is_tensor(b) else torch.from_numpy(np.asarray(b))
    a, b = a.detach().double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


RESULTS = []      # (name, rel err, tolerance, ok) of every comparison made so far


def check(name, err, tol):
    ok = err < tol
    RESULTS.append((name, err, tol, ok))
    print(f"{'ok  ' if ok else 'FAIL'} {name:70s} rel err {err:.3e} (tol {tol:.0e})", flush=True)


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def modes():
    fx, g2 = golden("g13_st_modes"), golden("g2_block")
    H, T, step = int(g2["heads"]), int(g2["seqlen"]), int(fx["row_step"])
    sd = {k[3:]: t(g2[k]) for k in g2.files if k.startswith("sd.") and "ts_attn" not in k}
    for mode in MODES:
        blk = Block(128, H, mlp_ratio=4, qkv_bias=True, norm_layer=LN, st_mode=mode, compute_dtype=torch.float32).to(DEV)
        blk.load_state_dict(sd)
        x = t(g2["x"]).clone().requires_grad_(True)
        with patched():
            out = blk(x, T)
            (out * t(fx["cot_tok"])).sum().backward()
        check(f"Block[{mode}] f32 forward vs reference", rel(out, fx[f"{mode}.blk.out"]), 2e-5)
        check(f"Block[{mode}] f32 dx vs reference", rel(x.grad, fx[f"{mode}.blk.dx"]), 1e-4)
        worst = max(rel(p.grad[::step] if p.dim() == 2 else p.grad, fx[f"{mode}.blk.grad.{n}"]) for n, p in blk.named_parameters())
        check(f"Block[{mode}] f32 parameter gradients vs reference", worst, 2e-4)
    # bf16 against fp64 autograd through the oracle
    N, T, P, H = 2, 2, 9, 2
    C, Fr = 64 * H, N * T
    p = {k[len("encoder.blocks.0."):]: v * (3.0 if k.endswith("weight") and v.dim() == 2 else 1.0)
         for k, v in R.make_params(embed_dim=C, depth=1, hidden_dim=64, layers=(1, 1, 1), n_tokens=P, seed=3).items()
         if k.startswith("encoder.blocks.0.") and "ts_attn" not in k}
    x0, dy = rnd(Fr, P, C, seed=1), rnd(Fr, P, C, seed=2)
    for mode in MODES:
        pd = {k: v.double().requires_grad_(True) for k, v in p.items()}
        xr = x0.double().requires_grad_(True)
        yref = R.block(xr, pd, "", H, T, mode)
        yref.backward(dy.double())
        blk = Block(C, H, mlp_ratio=4, qkv_bias=True, norm_layer=LN, st_mode=mode, compute_dtype=torch.bfloat16)
        blk.load_state_dict(p)
        blk = blk.to(DEV)
        xg = x0.clone().to(DEV).requires_grad_(True)
        with patched():
            y = blk(xg, T)
            y.backward(dy.to(DEV))
        check(f"Block[{mode}] bf16 forward vs fp64 oracle", rel(y, yref), 3e-2)
        check(f"Block[{mode}] bf16 dx vs fp64 oracle", rel(xg.grad, xr.grad), 3e-2)
        check(f"Block[{mode}] bf16 parameter gradients vs fp64 oracle", max(rel(q.grad, pd[n].grad) for n, q in blk.named_parameters()), 6e-2)


def iterative():
    fx = golden("g12_iterative")
    reg = Regressor(smpl_mean_params=dict(pose=fx["mean_pose"], shape=fx["mean_shape"], cam=fx["mean_cam"]), feat_dim=128, hidden_dim=64)
    reg.load_state_dict({k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("sd.")}, strict=False)
    reg = reg.to(DEV).eval()
    x = t(fx["x"])
    with patched(), torch.no_grad():
        if SIM:
            pose, shape, cam = reg._regress_hip(x, *reg._init(x.shape[0], None, None, None), 3)
            out = reg.get_output(pose, shape, cam, None, hip=True)
        else:
            out = reg(x, seqlen=3)                              # eval + no_grad on a cuda tensor = the HIP inference path
    for k in ("theta", "kp_2d", "kp_3d", "rotmat"):
        check(f"Regressor inference {k} vs reference", rel(out[k], fx[k]), 1e-4)
    xg = t(fx["x"]).clone().requires_grad_(True)
    with patched():
        if SIM:
            pose, shape, cam = reg.iterative_regress(xg)
            theta, verts, kp2d, kp3d, rotmat = tail.SmplTailFn.apply(pose, shape, cam, reg.smpl)
            out = dict(theta=theta, kp_2d=kp2d, kp_3d=kp3d)
        else:
            out = reg(xg, seqlen=3)                             # grad enabled: ATen head + tail.SmplTailFn
        sum((out[k] * t(fx["cot_" + k])).sum() for k in ("theta", "kp_2d", "kp_3d")).backward()
    check("Regressor training dx vs reference", rel(xg.grad, fx["gx"]), 5e-4)
    check("Regressor training parameter gradients vs reference", max(rel(p.grad, fx["grad." + n]) for n, p in reg.named_parameters()), 5e-4)


def evaluation():
    fx = golden("g14_eval")
    with patched():
        check("similarity transform vs reference", rel(EU.batch_compute_similarity_transform_torch(t(fx["S1"]), t(fx["S2"])), fx["S1_hat"]), 3e-4)
        check("compute_accel vs reference", rel(EU.compute_accel(t(fx["acc_pred"])), fx["accel"]), 1e-5)
        check("compute_error_accel vs reference", rel(EU.compute_error_accel(t(fx["acc_gt"]), t(fx["acc_pred"])), fx["accel_err"]), 1e-5)
        check("compute_error_verts vs reference", rel(EU.compute_error_verts(pred_verts=t(fx["verts_a"].astype(np.float32)),
                                                                             target_verts=t(fx["verts_b"].astype(np.float32))), fx["verts_err"]), 1e-5)
        b = {k[len("batch."):]: fx[k] for k in fx.files if k.startswith("batch.")}
        batch = {k: torch.from_numpy(v) for k, v in b.items() if v.dtype.kind != "U"}
        batch["instance_id"], batch["paths"] = [list(r) for r in b["instance_id"]], [list(r) for r in b["paths"]]

        class DS:
            dataset_name = "mpii3d"

        class Loader(list):
            dataset = DS()

        class OnDevice(StubModel):                              # the stub computes on the CPU; hand the Evaluator device tensors
            def forward(self, inp, J_regressor=None):
                return {k: v.to(DEV) for k, v in super().forward(inp, J_regressor).items()}

        model = OnDevice(R.make_synthetic_smpl(int(fx["smpl_seed"])))
        model.decoder = nn.Module()
        model.decoder.smpl = SMPL().to(DEV)
        ev = Evaluator()
        ev.inference(model, Loader([batch]), seqlen=3, interp=2, device=DEV, verbose=False)
        eval_dict, num_pred = ev.evaluate()
    assert num_pred == int(fx["num_pred"])
    for k, v in eval_dict.items():
        check(f"Evaluator eval_dict[{k}] vs reference", abs(v - float(fx["eval." + k])) / abs(float(fx["eval." + k])), 2e-4)


def long_attention():
    cases = [(2, 5, 2, L.IMPL_MFMA_LONG), (1, 197, 2, L.IMPL_MFMA_LONG), (1, 530, 1, L.IMPL_AUTO)]
    if not SIM:
        cases += [(1, 16 * 197, 2, L.IMPL_AUTO)]               # the real coupling sequence (cfg3): 3152 tokens
    for Fr, L_, H, impl in cases:
        qkv, do = rnd(Fr, L_, 3 * 64 * H, seed=L_).bfloat16(), rnd(Fr, L_, 64 * H, seed=4).bfloat16()
        x = qkv.double().requires_grad_(True)
        qq, kk, vv = R.split_qkv(x, H)
        oref = R.attention_spatial(qq, kk, vv, 64 ** -0.5)
        oref.backward(do.double())
        with patched():
            o, lse = ops.attn_spatial_fwd(qkv.to(DEV), H, impl)
            g = ops.attn_spatial_bwd(qkv.to(DEV), o, do.to(DEV), lse, H, impl=impl)
        check(f"long attention F{Fr} L{L_} H{H} impl{impl} forward vs fp64 oracle", rel(o, oref), 2e-2)
        check(f"long attention F{Fr} L{L_} H{H} impl{impl} dqkv vs fp64 oracle", rel(g, x.grad), 3e-2)
    qkv, do = rnd(1, 330, 3 * 64, seed=330), rnd(1, 330, 64, seed=4)          # f32 parity mode past the whole-head LDS limit: exact tiled kernels
    x = qkv.double().requires_grad_(True)
    oref = R.attention_spatial(*R.split_qkv(x, 1), 64 ** -0.5)
    oref.backward(do.double())
    with patched():
        o, lse = ops.attn_spatial_fwd(qkv.to(DEV), 1, L.IMPL_AUTO)
        g = ops.attn_spatial_bwd(qkv.to(DEV), o, do.to(DEV), lse, 1)
    check("long attention f32 (exact VALU tiles) L330 forward vs fp64 oracle", rel(o, oref), 2e-5)
    check("long attention f32 (exact VALU tiles) L330 dqkv vs fp64 oracle", rel(g, x.grad), 5e-5)
    N, T, P, H = 1, 8, 70, 1
    qkv, do = rnd(N * T, P, 3 * 64 * H, seed=21).bfloat16(), rnd(N * T, P, 64 * H, seed=22).bfloat16()
    x = qkv.double().requires_grad_(True)
    oref = R.attention_coupling(*R.split_qkv(x, H), T, 64 ** -0.5)
    oref.backward(do.double())
    xg = qkv.clone().to(DEV).requires_grad_(True)
    with patched():
        o = ste_modes.SpatialAttnFn.apply(xg.view(N, T * P, 3 * 64 * H), H, L.IMPL_AUTO).view(N * T, P, 64 * H)
        o.backward(do.to(DEV))
    check("coupling (8 x 70 tokens) forward vs fp64 oracle", rel(o, oref), 2e-2)
    check("coupling (8 x 70 tokens) dqkv vs fp64 oracle", rel(xg.grad, x.grad), 3e-2)


def conv3x3():
    import torch.nn.functional as F
    from maed_amd.resnetv2 import _same_pad
    bf = torch.bfloat16
    for N, I, O, H, W, stride in [(2, 64, 64, 6, 5, 1), (2, 64, 136, 8, 8, 2), (1, 64, 64, 7, 7, 2), (2, 128, 128, 28, 28, 1),
                                       (4, 64, 64, 14, 14, 1), (1, 64, 136, 8, 16, 1)]:     # last two: F*H*W % 64 == 0 -> own weight-gradient kernel
        x = rnd(N, I, H, W, seed=H).to(bf).float()
        w = (rnd(O, I, 3, 3, seed=W) * (1.0 / (3 * I ** 0.5))).to(bf).float()
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        ref = F.conv2d(_same_pad(xr, 3, stride), wr, None, stride)
        dy = rnd(*ref.shape, seed=3).to(bf).float()
        ref.backward(dy)
        xs = x.to(bf).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        ws = w.to(bf).to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).requires_grad_(True)
        with patched():
            y = ops.Conv3x3Fn.apply(xs, ws, stride)
            y.backward(dy.to(bf).to(DEV).contiguous(memory_format=torch.channels_last))
        tag = f"conv3x3 N{N} I{I} O{O} {H}x{W} stride {stride}"
        check(tag + " forward vs fp32 conv", rel(y, ref), 1e-2)
        check(tag + " input gradient", rel(xs.grad, xr.grad), 1e-2)
        check(tag + " weight gradient", rel(ws.grad, wr.grad), 2e-2)
    # round 4: the stride-2 convolutions' backward on the library (maed_conv3x3_s2_dgrad: one implicit GEMM per parity class of the input pixel;
    # maed_conv3x3_s2_wgrad: the TN kernel through per-output-pixel gather tables) -- taken when WeightStdFn hands over the transposed image and the fp32 dW slice
    for N, I, O, H, W in ([(4, 64, 128, 8, 8), (1, 128, 64, 7, 9)] if SIM else [(4, 64, 128, 16, 16), (1, 128, 128, 15, 17), (16, 256, 256, 28, 28), (4, 128, 128, 56, 56)]):
        x = rnd(N, I, H, W, seed=H + 1).to(bf).float()
        w = (rnd(O, I, 3, 3, seed=W + 1) * (1.0 / (3 * I ** 0.5))).to(bf).float()
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        ref = F.conv2d(_same_pad(xr, 3, 2), wr, None, 2)
        dy = rnd(*ref.shape, seed=4).to(bf).float()
        ref.backward(dy)
        xs = x.to(bf).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        ws = w.to(bf).to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        wt = w.to(bf).to(DEV).permute(2, 3, 1, 0).contiguous()                 # (3, 3, I, O): the transposed image maed_weight_std_fwd writes
        dw = torch.zeros(O, 9 * I, device=DEV)
        with patched():
            y = ops.Conv3x3Fn.apply(xs, ws, 2, wt, dw)
            y.backward(dy.to(bf).to(DEV).contiguous(memory_format=torch.channels_last))
            if not SIM:
                torch.cuda.synchronize()
                ops.side_stream_join(xs.device)
                torch.cuda.synchronize()
        tag = f"conv3x3 N{N} I{I} O{O} {H}x{W} stride 2, library backward"
        check(tag + " input gradient (parity classes)", rel(xs.grad, xr.grad), 1e-2)
        check(tag + " weight gradient (gather tables)" + ("" if (N * ref.shape[-2] * ref.shape[-1]) % 64 == 0 else " [ragged: framework fallback]"),
              rel(dw.view(O, 3, 3, I).permute(0, 3, 1, 2), wr.grad), 2e-2)


if __name__ == "__main__":
    for part in (conv3x3, long_attention, modes, iterative, evaluation):
        print(f"--- {part.__name__} ({'host simulator' if SIM else 'GPU'}) ---", flush=True)
        part()
    bad = [r for r in RESULTS if not r[3]]
    if bad:
        print(f"{len(bad)} of {len(RESULTS)} checks FAILED: " + "; ".join(r[0] for r in bad))
        sys.exit(1)
    print(f"ALL NEW PATHS OK ({len(RESULTS)} checks)")
