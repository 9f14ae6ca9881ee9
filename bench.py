#!/usr/bin/env python
"""bench.py -- MAED hot-path throughput on MI355X (contract: see the task statement / DESIGN.md §5).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one data-parallel TRAIN step (forward + backward + gradient all-reduce + Adam) of MAED on one
batch of synthetic clips resident in HBM: BASELINE.json cfg3/cfg4 -- per GPU 8 clips x 16 frames x 3x224x224,
STE depth 6 / heads 8 / dim 512, KTD hidden 1024, bf16 compute with fp32 master weights and fp32
residual stream.  Weak scaling: per-GPU batch fixed, value = (N * 8 clips) / max-over-ranks step time.
Rank 0 prints ONE JSON line.  It also carries
  roofline     : the STE spatial-attention forward kernel (the kernel north_star names), timed in situ with
                 hipEvents on the launch stream (maed_prof_*), against its HBM roofline; the MFMA view and the
                 other instrumented kernels are under "kernels".
  cpu_baseline : the CPU oracle (restatement of the reference's PyTorch CPU path, pinned to the reference by
                 tests/golden) timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("MAED_SYNTHETIC_SMPL_OK", "1")     # the licensed SMPL model file is unavailable: synthetic stand-in, stated in the JSON

CFG = dict(clips=8, T=16, img=224, depth=6, heads=8, dim=512, hidden=1024)
_T0 = time.perf_counter()


def log(msg):
    """progress on stderr (stdout carries only the JSON line)"""
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak


def build_model(dtype, device):
    import maed_amd
    torch.manual_seed(0)
    m = maed_amd.MAED(num_blocks=CFG["depth"], num_heads=CFG["heads"], embed_dim=CFG["dim"], hidden_dim=CFG["hidden"],
                      img_size=CFG["img"], max_seqlen=max(16, CFG["T"]), compute_dtype=dtype)
    return m.to(device)


# configs/config_stage2.yaml:34-39 (LOSS: KP_2D_W 300, KP_3D_W 600, SHAPE_W 0.06, POSE_W 60; SMPL_NORM/ACCL_W = config.py defaults)
LOSS_W = dict(e_loss_weight=300.0, e_3d_loss_weight=600.0, e_pose_loss_weight=60.0, e_shape_loss_weight=0.06, e_smpl_norm_loss=1.0, e_smpl_accl_loss=0.0)


def make_targets(n, T, device, gen):
    """synthetic labels in lib/core/loss.py's LossVideo layout (data_3d only: every clip carries 3D + SMPL labels)"""
    r = lambda *s: torch.randn(*s, generator=gen)
    return {k: v.to(device) for k, v in dict(
        kp_2d=torch.cat([r(n, T, 49, 2) * 0.3, torch.rand(n, T, 49, 1, generator=gen)], -1),
        kp_3d=torch.cat([r(n, T, 49, 3) * 0.3, torch.ones(n, T, 49, 1)], -1),
        theta=torch.cat([r(n, T, 3) * 0.1, r(n, T, 72) * 0.2, r(n, T, 10)], -1),
        w_smpl=(torch.rand(n, T, generator=gen) > 0.2).float()).items()}


def usable_cores(cap=32):
    """host cores this process may really use: affinity mask and cgroup quota, capped (torch CPU ops on the
    small per-frame tensors of this workload stop scaling -- and collapse -- far below 256 threads)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


def cpu_baseline(budget_s=25.0):
    """oracle train step (fwd + autograd bwd + torch Adam) on host cores; bounded sample: 1 clip x 16 frames."""
    from oracle import loss_ref
    from oracle import maed_ref as R
    cores = usable_cores()
    torch.set_num_threads(cores)
    P = (CFG["img"] // 16) ** 2 + 1
    params = R.make_params(embed_dim=CFG["dim"], depth=CFG["depth"], hidden_dim=CFG["hidden"], n_tokens=P, seed=0)
    params = {k: v.requires_grad_(True) for k, v in params.items()}
    sp = R.make_synthetic_smpl(0)
    opt = torch.optim.Adam(list(params.values()), lr=1e-4, weight_decay=1e-5)
    gen = torch.Generator().manual_seed(1)
    n_clips = 1
    clip = torch.randn(n_clips, CFG["T"], 3, CFG["img"], CFG["img"], generator=gen)
    tgt = make_targets(n_clips, CFG["T"], "cpu", gen)

    def step():
        opt.zero_grad()
        loss, _ = loss_ref.loss_video(R.maed_forward(clip, params, sp, CFG["depth"], CFG["heads"]), tgt, None, LOSS_W["e_loss_weight"],
                                      LOSS_W["e_3d_loss_weight"], LOSS_W["e_pose_loss_weight"], LOSS_W["e_shape_loss_weight"], LOSS_W["e_smpl_norm_loss"])
        loss.backward()
        opt.step()

    t0 = time.perf_counter()
    step()  # warm-up (also bounds the sample: if one step is slow we time fewer)
    warm = time.perf_counter() - t0
    log(f"cpu_baseline: warm-up step {warm:.1f}s on {cores} threads")
    if warm > budget_s:  # already over budget: the warm-up step IS the sample
        n, dt = 1, warm
    else:
        n = max(1, min(5, int(budget_s / max(warm, 1e-3)) - 1))
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        dt = (time.perf_counter() - t0) / n
    return dict(value=n_clips / dt, unit="video-clips/sec", cores=cores, kind="port",
                sample=f"{n} timed train steps (fwd+bwd+Adam, fp32) of {n_clips} clip x {CFG['T']} frames x {CFG['img']}^2 after 1 warm-up; "
                       f"oracle/maed_ref.py on torch CPU ops, {cores} threads (of {os.cpu_count()} logical CPUs)", s_per_step=dt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--forward-only", action="store_true", help="cfg2: inference forward instead of the train step")
    ap.add_argument("--workload", default="cfg3", choices=["cfg3", "cfg5"], help="cfg3 = BASELINE's metric workload (default); cfg5 = long-clip stress")
    args = ap.parse_args()

    if args.workload == "cfg5":   # BASELINE.json configs[4]: long-clip stress (per-GPU clips stated in config.workload)
        CFG.update(clips=2, T=64, img=256, depth=12, heads=12, dim=768)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    from maed_amd import _lib as L
    from maed_amd.ddp import FusedAdam, GradBucketer, ParamArena
    from maed_amd.loss import LossVideo
    lib = L.lib()  # raises if libmaed_hip.so is missing: no fallback

    log(f"building model ({args.dtype}) on {dev}")
    model = build_model(dtype, dev)
    gen = torch.Generator().manual_seed(1000 + rank)
    clip = torch.randn(CFG["clips"], CFG["T"], 3, CFG["img"], CFG["img"], generator=gen).to(dev)
    tgt = make_targets(CFG["clips"], CFG["T"], dev, gen)

    if args.forward_only:
        model.eval()

        def step():
            with torch.no_grad():
                model(clip)
    else:
        model.train()
        arena = ParamArena(model)
        comm = None
        if os.environ.get("MAED_COMM", "torch") == "direct":   # the library's own RCCL communicator + side stream (maed_comm_*)
            from maed_amd.ddp import RcclComm
            comm = RcclComm()
        bucketer = GradBucketer(arena, model, comm=comm)
        bucketer.broadcast_parameters(0)
        opt = FusedAdam(arena, lr=1e-4, weight_decay=1e-5, bucketer=bucketer)  # configs/config_stage2.yaml:63-66
        criterion = LossVideo(**LOSS_W)                                        # lib/core/loss.py via maed_loss_fwd_bwd

        def step():
            opt.zero_grad()
            loss, _ = criterion(model(clip), tgt, None)
            loss.backward()
            opt.step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # setup, not a benchmark step: the first pass through the model makes MIOpen search its convolution algorithms (~20 s) and
    # loads every code object; it is kept out of the W warm-up steps so that a small --warmup cannot put it next to the timed region
    t1 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    log(f"setup pass (MIOpen algorithm search, code-object loading): {time.perf_counter() - t1:.3f}s")
    for i in range(args.warmup):
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        log(f"warm-up step {i}: {time.perf_counter() - t1:.3f}s")
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = tt.item()
    ms_per_step = 1e3 * dt / args.steps
    log(f"timed {args.steps} steps: {ms_per_step:.2f} ms/step")

    # ---- in-situ kernel timing for the roofline (extra steps, events on the launch stream) ----------------
    kernels = {}
    roofline = None
    roofline_dominant = None
    # EVERY rank runs the extra steps (a train step contains the gradient all-reduce: a rank stepping alone would wait for
    # collectives nobody else issues); only rank 0 brackets its launches with events and reports
    nprof = 3
    if rank == 0:
        lib.maed_prof_enable(1)
    for _ in range(nprof):
        step()
    fence()
    if rank == 0:
        ms = (ctypes.c_double * 8)()
        cnt = (ctypes.c_int * 8)()
        lib.maed_prof_collect(ms, cnt)
        lib.maed_prof_enable(0)
        Fr, P, C_, T = CFG["clips"] * CFG["T"], (CFG["img"] // 16) ** 2 + 1, CFG["dim"], CFG["T"]
        M, es, Hd = Fr * P, (2 if dtype == torch.bfloat16 else 4), 4 * CFG["dim"]
        # algorithmic work per launch: flops, HBM bytes
        work = {
            "attn_spatial_fwd": (4.0 * P * P * C_ * Fr, (4.0 * M * C_) * es + 4.0 * Fr * CFG["heads"] * P),
            "attn_temporal_fwd": (4.0 * P * T * C_ * Fr, (4.0 * M * C_) * es + 4.0 * Fr * CFG["heads"] * P),
            "gemm_qkv": (2.0 * M * 3 * C_ * C_, (M * C_ + 3 * C_ * C_ + 3 * M * C_) * es),
            "gemm_fc1_gelu": (2.0 * M * Hd * C_, (M * C_ + Hd * C_ + 2 * M * Hd) * es),
            "gemm_fc2_resid": (2.0 * M * Hd * C_, (M * Hd + Hd * C_) * es + 8.0 * M * C_),
            "attn_spatial_bwd": (14.0 * P * P * C_ * Fr, (3 + 1 + 1 + 3) * M * C_ * es),
            "attn_temporal_bwd": (10.0 * P * T * C_ * Fr, (3 + 1 + 1 + 3) * M * C_ * es),
            "gemm_wgrad(all)": (None, None),
        }
        names = list(work)
        for i, nm in enumerate(names):
            if cnt[i] == 0:
                continue
            us = 1e3 * ms[i] / cnt[i]
            fl, by = work[nm]
            ent = dict(launches=cnt[i], avg_us=round(us, 2))
            if fl:
                ent.update(tflops=round(fl / us / 1e6, 2), frac_mfma_peak=round(fl / us / 1e6 / MFMA_BF16_PEAK_TF, 4),
                           algorithmic_gbs=round(by / us / 1e3, 1), frac_hbm_peak=round(by / us / 1e3 / HBM_PEAK_GBS, 4))
            kernels[nm] = ent
        traffic, traffic_note = None, ""
        try:  # HBM bytes per launch from the committed PMC passes (scripts/gpu_pmc.sh), same kernel and shape
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc", "attn_traffic.json")))
            if dtype == torch.bfloat16:
                traffic = tj["hbm_bytes_per_launch"]
                traffic_note = "; traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 B from separate rocprofv3 --pmc passes (profiles/r01_pmc/attn_traffic.json)"
        except Exception:
            pass
        if "attn_spatial_fwd" in kernels:
            k = kernels["attn_spatial_fwd"]
            roofline = dict(kernel="attn_sp_fwd_mfma (STE spatial attention forward)", bound="hbm", achieved=k["algorithmic_gbs"], peak=HBM_PEAK_GBS,
                            unit="GB/s", frac=round(k["algorithmic_gbs"] / HBM_PEAK_GBS, 4), traffic=traffic, avg_us=k["avg_us"],
                            mfma_view=dict(achieved=k["tflops"], peak=MFMA_BF16_PEAK_TF, unit="TFLOP/s", frac=k["frac_mfma_peak"]),
                            note="algorithmic bytes = q,k,v read + o written once (8*P*C*F B bf16) + lse; in-situ hipEvent timing over "
                                 f"{cnt[0]} launches inside {nprof} extra steps" + traffic_note)

        if "gemm_qkv" in kernels:   # by time the step's dominant own kernel family is the bf16 GEMM (MFMA-bound): report it beside the named one
            k = kernels["gemm_qkv"]
            roofline_dominant = dict(kernel="gemm_nt_glds_bf16_kernel (STE qkv projection; same kernel runs fc1/fc2/proj and the backbone's 1x1 convolutions)",
                                     bound="mfma", achieved=k["tflops"], peak=MFMA_BF16_PEAK_TF, unit="TFLOP/s", frac=k["frac_mfma_peak"], traffic=None,
                                     avg_us=k["avg_us"], note="2*M*N*K FLOP per launch (M = frames*tokens, N = 3C, K = C); in-situ hipEvent timing")

    log(f"kernel timing done: {json.dumps(kernels)}")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline()
        except Exception as e:  # the baseline must never take the bench line down
            cpu = dict(value=None, unit="video-clips/sec", cores=os.cpu_count(), kind="port", sample=f"failed: {e!r}")

    if rank == 0:
        clips = CFG["clips"] * world
        out = {
            "metric": "video-clips/sec (BxT frames) train step" if not args.forward_only else "video-clips/sec (BxT frames) forward",
            "value": round(clips * args.steps / dt, 3), "unit": "video-clips/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "frames_per_sec": round(clips * CFG["T"] * args.steps / dt, 1),
            "config": {"workload": (args.workload if args.workload != "cfg3" else "cfg3/cfg4" if not args.forward_only else "cfg2") + f": {CFG['clips']} clips x {CFG['T']} frames x 3x{CFG['img']}x{CFG['img']} per GPU, "
                       f"hybrid R50(3,4,9)+STE depth{CFG['depth']} heads{CFG['heads']} dim{CFG['dim']}+KTD hidden{CFG['hidden']}, "
                       + ("train step fwd+bwd+allreduce+Adam" if not args.forward_only else "inference forward"),
                       "global_batch_clips": clips, "frames_per_clip": CFG["T"], "parallelism": f"dp{world}",
                       "loss": "lib/core/loss.py LossVideo (config_stage2 weights) on synthetic labels, fused fwd+bwd kernel",
                       "smpl": "synthetic SMPL-shaped parameters (licensed model file unavailable)"},
            "roofline": roofline, "roofline_dominant": roofline_dominant, "kernels": kernels, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
