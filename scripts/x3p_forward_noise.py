"""Is the twin forward with plane storage (MAED_OPT_X3_PLANES = 6) the same forward as without (0)?  Run-to-run the accurate mode's outputs move by ~1e-5 of their
maximum (order of fp32 atomics in the split-K head GEMMs and token means, amplified by the backbone); three forwards per setting at full cfg3 module size, one clip:
differences WITHIN a setting against differences ACROSS the settings."""
import os, sys, itertools
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import maed_amd
from maed_amd import ops, _lib as L
from oracle import maed_ref as R
CFG = dict(depth=6, H=8, img=224, hidden=1024, T=16)
C, P = 64 * CFG["H"], (CFG["img"] // 16) ** 2 + 1
params = R.make_params(embed_dim=C, depth=CFG["depth"], hidden_dim=CFG["hidden"], n_tokens=P, seed=7)
clip = torch.randn(1, CFG["T"], 3, CFG["img"], CFG["img"], generator=torch.Generator().manual_seed(21))
ops.set_float32_matmul_precision("bf16x3")
ops.set_float32_backward_precision("bf16")
m = maed_amd.MAED(num_blocks=CFG["depth"], num_heads=CFG["H"], embed_dim=C, hidden_dim=CFG["hidden"], img_size=CFG["img"], compute_dtype=torch.float32)
m.load_state_dict(params, strict=False)
m = m.to("cuda").train()
m.decoder.drop1.p = 0.0; m.decoder.drop2.p = 0.0
runs = {}
for rep in range(3):
    for v in (0, 6):
        L.set_option(L.OPT_X3_PLANES, v)
        out = m(clip.to("cuda"))
        torch.cuda.synchronize()
        runs[v, rep] = {k: out[k].detach().float().cpu() for k in ("theta", "verts", "kp_3d")}
        del out
        ops.shadow_clear()
for k in ("theta", "verts", "kp_3d"):
    scale = max(float(r[k].abs().max()) for r in runs.values())
    within = max(float((runs[a][k] - runs[b][k]).abs().max()) for a, b in itertools.combinations(runs, 2) if a[0] == b[0]) / scale
    across = max(float((runs[a][k] - runs[b][k]).abs().max()) for a, b in itertools.combinations(runs, 2) if a[0] != b[0]) / scale
    print(f"{k:6s}: max difference within a setting {within:.2e}, across the settings {across:.2e}   (of the output's maximum)")
