"""CPU estimate (oracle arithmetic, no GPU) of the forward error of 16-bit ENGINES between "bf16 everywhere" and the split-bf16 products, at full cfg3 module
size (one 16-frame 224^2 clip per seed; clips are independent, so one clip is the whole story) -- VERDICT r4 "next" item 1(a).

Emulation of what the product's 16-bit mode does: every matrix product (convolution, Linear, QK^T, PV) rounds BOTH operands to the storage format and accumulates in
fp32 (a product of two 11-bit or 8-bit significands is exact in fp32: this IS the MFMA's arithmetic up to summation order); convolution outputs and GroupNorm /
LayerNorm outputs are stored rounded; statistics, softmax, GELU and the STE's residual stream stay fp32; the decoder head stays fp32 (split products in the product).

Engines:  bf16 (the headline mode), fp16 (v_mfma_f32_32x32x16_f16: same rate, 3 more significand bits), bf16 activations hi+lo x bf16 weight (2 products),
fp16 backbone + bf16 STE and the reverse (which half carries the error).

    python scripts/half_precision_probe_cpu.py [--seeds 7 8 9] [--engines bf16 fp16 ...]
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import maed_ref as R  # noqa: E402

_conv2d, _linear, _gn, _ln = F.conv2d, F.linear, F.group_norm, F.layer_norm


def rnd(x, fmt):
    if fmt is None:
        return x
    if fmt == "bf16":
        return x.bfloat16().float()
    if fmt == "fp16":
        return x.half().float()
    if fmt == "bf16x2":      # hi + lo planes: 16 significand bits
        hi = x.bfloat16().float()
        return hi + (x - hi).bfloat16().float()
    raise ValueError(fmt)


class Engine:
    """act = format activations are stored / multiplied in, wgt = format of the weight operand; separately for the backbone (convolutions) and the STE"""

    def __init__(self, bb_act, bb_wgt, ste_act, ste_wgt):
        self.bb_act, self.bb_wgt, self.ste_act, self.ste_wgt = bb_act, bb_wgt, ste_act, ste_wgt
        self.max_abs = 0.0

    def conv2d(self, x, w, b=None, stride=1, padding=0, *a, **k):
        y = _conv2d(rnd(x, self.bb_act), rnd(w, self.bb_wgt), b, stride, padding, *a, **k)
        self.max_abs = max(self.max_abs, float(y.abs().max()))
        return rnd(y, "bf16" if self.bb_act == "bf16x2" else self.bb_act)

    def group_norm(self, x, *a, **k):
        return rnd(_gn(x, *a, **k), "bf16" if self.bb_act == "bf16x2" else self.bb_act)

    def linear(self, x, w, b=None):
        if self.in_head:
            return _linear(x, w, b)
        y = _linear(rnd(x, self.ste_act), rnd(w, self.ste_wgt), b)
        self.max_abs = max(self.max_abs, float(y.abs().max()))
        return y

    def mm(self, a, b):
        f = "bf16" if self.ste_act == "bf16x2" else self.ste_act
        return rnd(a, f) @ rnd(b, f)


def run(engine, x, params, sp, depth, H, T):
    if engine is None:
        feat = R.ste_forward_features(x, params, "encoder.", depth, H, T)
    else:
        e = engine
        e.in_head = False

        def attention_spatial(q, k, v, scale):
            Fr, Hh, P, d = q.shape
            attn = (e.mm(q, k.transpose(-2, -1)) * scale).softmax(dim=-1)
            return e.mm(attn, v).transpose(1, 2).reshape(Fr, P, Hh * d)

        def attention_temporal(q, k, v, T, scale):
            Fr, Hh, P, d = q.shape
            qt = q.reshape(-1, T, Hh, P, d).permute(0, 2, 3, 1, 4)
            kt = k.reshape(-1, T, Hh, P, d).permute(0, 2, 3, 1, 4)
            vt = v.reshape(-1, T, Hh, P, d).permute(0, 2, 3, 1, 4)
            attn = (e.mm(qt, kt.transpose(-2, -1)) * scale).softmax(dim=-1)
            return e.mm(attn, vt).permute(0, 3, 2, 1, 4).reshape(Fr, P, Hh * d)

        saved = (F.conv2d, F.linear, F.group_norm, R.attention_spatial, R.attention_temporal)
        F.conv2d, F.linear, F.group_norm = e.conv2d, e.linear, e.group_norm
        R.attention_spatial, R.attention_temporal = attention_spatial, attention_temporal
        try:
            feat = R.ste_forward_features(x, params, "encoder.", depth, H, T)
        finally:
            F.conv2d, F.linear, F.group_norm, R.attention_spatial, R.attention_temporal = saved
    pose, shape, cam = R.ktd_head(feat, params, "decoder.")
    return R.ktd_get_output(pose, shape, cam, sp), feat


ENGINES = {
    "bf16": ("bf16", "bf16", "bf16", "bf16"),
    "fp16": ("fp16", "fp16", "fp16", "fp16"),
    "fp16_backbone+bf16_ste": ("fp16", "fp16", "bf16", "bf16"),
    "bf16_backbone+fp16_ste": ("bf16", "bf16", "fp16", "fp16"),
    "act_hi+lo_x_bf16_wgt": ("bf16x2", "bf16", "bf16x2", "bf16"),
    "act_hi+lo_x_fp16_wgt": ("bf16x2", "fp16", "bf16x2", "fp16"),
    "fp32_backbone+fp16_ste": (None, None, "fp16", "fp16"),
    "fp16_backbone+fp32_ste": ("fp16", "fp16", None, None),
}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[7, 8, 9])
    ap.add_argument("--engines", nargs="+", default=["bf16", "fp16", "act_hi+lo_x_bf16_wgt", "fp16_backbone+bf16_ste", "bf16_backbone+fp16_ste"])
    ap.add_argument("--cfg", default="cfg3", choices=["cfg3", "cfg5"])
    ap.add_argument("--frames", type=int, default=0)
    a = ap.parse_args()
    if a.cfg == "cfg3":
        depth, H, img, hidden, T = 6, 8, 224, 1024, 16
    else:
        depth, H, img, hidden, T = 12, 12, 256, 1024, 64
    C, P = 64 * H, (img // 16) ** 2 + 1
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    rel = lambda x, y: float((x.double() - y.double()).abs().max() / y.double().abs().max())
    rms = lambda x, y: float((x.double() - y.double()).pow(2).mean().sqrt() / y.double().std())
    for seed in a.seeds:
        params = R.make_params(embed_dim=C, depth=depth, hidden_dim=hidden, n_tokens=P, seed=seed, **({"max_seqlen": T} if a.cfg == "cfg5" else {}))
        sp = R.make_synthetic_smpl(0)
        x = torch.randn(1, T, 3, img, img, generator=torch.Generator().manual_seed(21 + seed)).reshape(-1, 3, img, img)
        with torch.no_grad():
            t0 = time.time()
            o32, f32 = run(None, x, params, sp, depth, H, T)
            print(f"seed {seed}: fp32 oracle {time.time() - t0:.0f}s", flush=True)
            for name in a.engines:
                e = Engine(*ENGINES[name])
                o, f = run(e, x, params, sp, depth, H, T)
                print(f"  {name:28s} theta max-rel {rel(o['theta'], o32['theta']):.2e}  rms/std {rms(o['theta'], o32['theta']):.2e} | feature {rel(f, f32):.2e} | "
                      f"kp_3d {rel(o['kp_3d'], o32['kp_3d']):.2e} kp_2d {rel(o['kp_2d'], o32['kp_2d']):.2e} verts {rel(o['verts'], o32['verts']):.2e} | max |product output| {e.max_abs:.0f}", flush=True)
