"""Full-size (cfg3 module dimensions, ONE 16-frame 224^2 clip) forward + backward of the product in every fp32 matmul mode against the CPU oracle:
outputs vs the fp32 oracle (north_star: 1e-3 on SMPL parameters), EVERY parameter gradient vs fp64 autograd through the oracle -- and, as the
yardstick for the gradients, the fp32 oracle's own distance from fp64 (the reference's arithmetic is fp32: vision_transformer.py:146-228,
resnetv2.py:74-93).  Prints one line per (mode, parameter group).   python scripts/x3_probe.py [modes...]   (default: exact bf16x3 bf16x6)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MAED_SYNTHETIC_SMPL_OK", "1")
from oracle import maed_ref as R  # noqa: E402

CFG = dict(depth=6, H=8, img=224, hidden=1024, T=16)
WTS = {"theta": 1.0, "kp_3d": 1.0, "kp_2d": 0.01}


def group_of(name):
    if "backbone" in name:
        s = name.split("backbone.")[1]
        return "backbone." + (s.split(".")[0] if s.startswith("stem") else ".".join(s.split(".")[:2]))
    if name.startswith("encoder.blocks"):
        return "ste.blocks"
    if name.startswith("encoder"):
        return "ste.embed"
    return "decoder"


def oracle_grads(params, clip, sp, dtype):
    pd = {k: v.clone().to(dtype).requires_grad_(True) for k, v in params.items()}
    spd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sp.items()}
    out = R.maed_forward(clip.to(dtype), pd, spd, depth=CFG["depth"], H=CFG["H"])
    loss = sum(w * (out[k] ** 2).mean() for k, w in WTS.items())
    loss.backward()
    return {k: v.detach() for k, v in out.items()}, {k: v.grad for k, v in pd.items() if v.grad is not None}, loss.item()


def main():
    import maed_amd
    modes = sys.argv[1:] or ["exact", "bf16x3", "bf16x6"]
    dev = "cuda"
    C, P = 64 * CFG["H"], (CFG["img"] // 16) ** 2 + 1
    params = R.make_params(embed_dim=C, depth=CFG["depth"], hidden_dim=CFG["hidden"], n_tokens=P, seed=7)
    sp = R.make_synthetic_smpl(0)
    clip = torch.randn(1, CFG["T"], 3, CFG["img"], CFG["img"], generator=torch.Generator().manual_seed(21))
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    t0 = time.time()
    o64, g64, l64 = oracle_grads(params, clip, sp, torch.float64)
    t1 = time.time()
    o32, g32, l32 = oracle_grads(params, clip, sp, torch.float32)
    print(f"oracle: fp64 fwd+bwd {t1 - t0:.1f}s, fp32 {time.time() - t1:.1f}s; loss fp64 {l64:.8f} fp32 {l32:.8f}", flush=True)
    rel = lambda a, b: float((a.double().cpu() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))

    def summarize(tag, grads):
        by = {}
        for n, gr in g64.items():
            if n not in grads or grads[n] is None:
                continue
            by.setdefault(group_of(n), []).append((rel(grads[n], gr), n))
        for grp in sorted(by):
            v = sorted(by[grp])
            print(f"  {tag:22s} grad rel-to-max vs fp64  {grp:22s} n={len(v):3d} median {v[len(v) // 2][0]:.2e} worst {v[-1][0]:.2e} ({v[-1][1]})", flush=True)

    print("fp32 ORACLE vs fp64 oracle (the reference arithmetic's own distance):")
    for k in ("theta", "kp_3d", "kp_2d", "rotmat", "verts"):
        print(f"  fp32 oracle            out {k:7s} rel-to-max vs fp64 {rel(o32[k], o64[k]):.2e}")
    summarize("fp32 oracle", g32)
    for mode in modes:
        dtype = torch.bfloat16 if mode == "bf16" else torch.float32
        glob, _, bb = mode.partition("+bb:")          # "bf16x3+bb:bf16x6" = process-wide bf16x3, the backbone on its own bf16x6 engine
        if mode != "bf16":
            maed_amd.set_float32_matmul_precision(glob)
        m = maed_amd.MAED(num_blocks=CFG["depth"], num_heads=CFG["H"], embed_dim=C, hidden_dim=CFG["hidden"], img_size=CFG["img"], compute_dtype=dtype,
                          backbone_f32_matmul=bb or None)
        m.load_state_dict(params, strict=False)
        m = m.to(dev).train()
        m.decoder.drop1.p = 0.0
        m.decoder.drop2.p = 0.0
        for it in range(2):         # second pass timed (first: MIOpen search / code objects)
            for p in m.parameters():
                p.grad = None
            torch.cuda.synchronize(); t0 = time.time()
            out = m(clip.to(dev))
            loss = sum(w * (out[k] ** 2).mean() for k, w in WTS.items())
            loss.backward()
            torch.cuda.synchronize(); dt = time.time() - t0
        print(f"mode {mode}: fwd+bwd of one clip {dt * 1e3:.1f} ms; loss {loss.item():.8f}", flush=True)
        for k in ("theta", "kp_3d", "kp_2d", "rotmat", "verts"):
            print(f"  {mode:22s} out {k:7s} rel-to-max vs fp32 oracle {rel(out[k].detach().float(), o32[k]):.2e}   vs fp64 {rel(out[k].detach().float(), o64[k]):.2e}")
        summarize(mode, {n: p.grad for n, p in m.named_parameters()})
        del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
