"""TEST INFRASTRUCTURE, build container only: time the REFERENCE's own CPU path (lib/models MAED + lib/core/loss.py LossVideo + torch Adam,
shim-imported from /root/reference) next to the oracle port (oracle/maed_ref.py + oracle/loss_ref.py) on the same inputs, weights and
thread count -- BASELINE.md B1/B2: what bench.py's cpu_baseline (kind "port", the only CPU code that can travel to the GPU box) stands for.

    python -m oracle.time_reference_cpu [--clips 2] [--frames 8] [--steps 3] [--threads N]     (cfg1: 2 x 8 x 224^2, C=768, H=12, depth 6)

Prints and writes profiles/<tag>_reference_vs_port_cpu.json: seconds per forward and per train step (forward + loss + backward + Adam)
of both, their ratio, and the largest output difference.  Step semantics follow lib/core/trainer.py:240-257 (forward, criterion,
zero_grad, backward, optimizer.step)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import loss_ref, ref_shims  # noqa: E402
from oracle import maed_ref as R  # noqa: E402


def median(v):
    v = sorted(v)
    return v[len(v) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=2)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--heads", type=int, default=12)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_reference_vs_port_cpu.json"))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    ref_shims.install(smpl_seed=0)
    from lib.core.loss import LossVideo
    from lib.models import MAED
    if a.dim != 768:      # the reference hard-codes embed_dim 768 in its factory (vision_transformer.py:560-576)
        raise SystemExit("the reference's factory only builds embed_dim 768: time cfg1's model (--dim 768 --heads 12)")
    torch.manual_seed(0)
    model = MAED(encoder="ste", num_blocks=6, num_heads=a.heads, st_mode="parallel", decoder="ktd", hidden_dim=1024)
    params = R.make_params(embed_dim=a.dim, depth=6, hidden_dim=1024, seed=7)
    missing, unexpected = model.load_state_dict(params, strict=False)
    assert not unexpected and all(".smpl." in k for k in missing)
    for m in model.modules():      # KTD's Dropout(0.5) draws from the RNG: switch it off on both sides so that the two steps are the same arithmetic
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.train()
    g = torch.Generator().manual_seed(1234)
    N, T = a.clips, a.frames
    clip = torch.randn(N, T, 3, 224, 224, generator=g)
    r = lambda *s: torch.randn(*s, generator=g)
    tgt = dict(kp_2d=torch.cat([r(N, T, 49, 2) * 0.3, torch.rand(N, T, 49, 1, generator=g)], -1), kp_3d=torch.cat([r(N, T, 49, 3) * 0.3, torch.ones(N, T, 49, 1)], -1),
               theta=torch.cat([r(N, T, 3) * 0.1, r(N, T, 72) * 0.2, r(N, T, 10)], -1), w_smpl=(torch.rand(N, T, generator=g) > 0.2).float())
    W = dict(e_loss_weight=300.0, e_3d_loss_weight=600.0, e_pose_loss_weight=60.0, e_shape_loss_weight=0.06, e_smpl_norm_loss=1.0, e_smpl_accl_loss=0.0)

    # ---- the reference
    crit = LossVideo(device="cpu", **W)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-5)

    def ref_fwd():
        with torch.no_grad():
            return model(clip)

    def ref_step():
        preds = model(clip)
        loss, _ = crit(preds, tgt, None)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss.item()

    # ---- the port (functional; parameters are leaves of the same values)
    pp = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    sp = R.make_synthetic_smpl(0)
    popt = torch.optim.Adam(list(pp.values()), lr=1e-4, weight_decay=1e-5)

    def port_fwd():
        with torch.no_grad():
            return R.maed_forward(clip, pp, sp, 6, a.heads)

    def port_step():
        loss, _ = loss_ref.loss_video(R.maed_forward(clip, pp, sp, 6, a.heads), tgt, None, W["e_loss_weight"], W["e_3d_loss_weight"], W["e_pose_loss_weight"],
                                      W["e_shape_loss_weight"], W["e_smpl_norm_loss"])
        popt.zero_grad()
        loss.backward()
        popt.step()
        return loss.item()

    def bench2(fa, fb, n):
        """interleaved (a, b, a, b, ...): the two sides see the same machine state -- timing one after the other in this container moved
        the ratio between 0.6 and 1.5 from run to run"""
        fa(); fb()                              # warm-up
        ta, tb = [], []
        for _ in range(n):
            t0 = time.perf_counter(); oa = fa(); ta.append(time.perf_counter() - t0)
            t0 = time.perf_counter(); ob = fb(); tb.append(time.perf_counter() - t0)
        return median(ta), median(tb), oa, ob

    res = {"threads": a.threads, "logical_cpus": os.cpu_count(), "workload": f"cfg1: {N} clips x {T} frames x 224^2, STE dim {a.dim} heads {a.heads} depth 6, KTD hidden 1024, fp32",
           "torch": torch.__version__, "timing": f"median of {a.steps} interleaved (reference, port) pairs after 1 warm-up each"}
    tf_ref, tf_port, o_ref, o_port = bench2(ref_fwd, port_fwd, a.steps)
    res["forward_s"] = {"reference": round(tf_ref, 3), "port": round(tf_port, 3), "port_over_reference": round(tf_port / tf_ref, 3)}
    res["forward_max_abs_diff"] = {k: float((o_ref[k] - o_port[k]).abs().max()) for k in ("theta", "kp_3d", "kp_2d")}
    ts_ref, ts_port, l_ref, l_port = bench2(ref_step, port_step, a.steps)
    res["train_step_s"] = {"reference": round(ts_ref, 3), "port": round(ts_port, 3), "port_over_reference": round(ts_port / ts_ref, 3)}
    res["clips_per_s_train"] = {"reference": round(N / ts_ref, 3), "port": round(N / ts_port, 3)}
    res["last_loss"] = {"reference": l_ref, "port": l_port}
    print(json.dumps(res, indent=1))
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
