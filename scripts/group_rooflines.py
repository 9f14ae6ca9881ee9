"""Per-group time and (where the bytes are known in closed form) achieved bandwidth from a SINGLE-STREAM steady-state kernel summary
(scripts/summarize_trace.py output, taken with MAED_WGRAD_SIDE_STREAM=0 so that kernel durations are not inflated by concurrency):
    python scripts/group_rooflines.py steady_state_kernels.csv [traffic.json]
With a traffic.json the result is stored under "__groups__" (bench.py then reports it as kernel_groups for the same source hash).
GroupNorm bytes per step at cfg3 (128 frames, bf16), from the layer list of the hybrid R50: forward apply reads x (+ residual) and writes y (+ 1 bit per element
when a residual precedes the ReLU); every layer's statistics come from the epilogue of the convolution in front (since round 4 the stem's too: csrc/stem.hip); backward (round 4: one pass, gn_bwd_onepass_kernel)
reads x, dy (+ bits) once and writes dx (+ the masked residual gradient in the four downsample blocks) -- the round-3 figure counted x and dy twice (reduce + apply)."""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def groupnorm_bytes(frames=128, E=2):
    # (channels, H) of every GroupNorm in forward order; residual = norm3 of each bottleneck; stats_pass = a separate statistics read of x (none since the stem runs
    # on the library: MAED_STEM_OWN=0 brings the gn_stats_kernel pass back)
    layers = [(64, 112, False, os.environ.get("MAED_STEM_OWN", "1") == "0")]     # stem norm
    chans, depth, H = (256, 512, 1024), (3, 4, 9), (56, 28, 14)
    for s in range(3):
        for b in range(depth[s]):
            mid = chans[s] // 4
            h_in = H[s] * 2 if (b == 0 and s > 0) else H[s]              # conv1 of a stage's first block still runs at the previous resolution
            if b == 0:
                layers.append((chans[s], H[s], False, False))            # downsample.norm
            layers.append((mid, h_in, False, False))                     # norm1
            layers.append((mid, H[s], False, b == 0 and s > 0))          # norm2 (behind the strided 3x3: own kernel forward -> fused stats; kept False)
            layers.append((chans[s], H[s], True if b == 0 else "lazy", False))   # norm3 (+ residual, ReLU); identity blocks do not write the masked residual gradient
    fwd = bwd = 0
    for C, h, res, stats_pass in layers:
        t = frames * h * h * C * E
        bits = t // (8 * E) if res else 0
        fwd += t * (2 + (1 if res else 0)) + bits + (t if stats_pass else 0)
        bwd += t * (3 + (1 if res is True else 0)) + bits                # round 4: ONE pass -- x, dy read once, dx written (+ the masked residual gradient in the downsample blocks)
    return len(layers), fwd, bwd


def main():
    rows = [r for r in csv.reader(open(sys.argv[1])) if len(r) == 5 and r[0] != "kernel"]
    grp = {}
    def which(n):
        if n.startswith("gn_"): return "groupnorm"
        if "igemm" in n or n.startswith("ck::") or "SubTensor" in n or "Cijk" in n: return "miopen_rocblas"
        if n.startswith("gemm_tn"): return "gemm_tn"
        if n.startswith("gemm_nt_glds") or n.startswith("gemm_nt_256") or n.startswith("gemm_nt_mfma"): return "gemm_nt"
        if n.startswith("conv3x3") or n.startswith("stem7x7s2") or n.startswith("stem_wimg") or n.startswith("wgrad_slots_reduce"): return "conv3x3"   # (+ the stem: 3 launches)
        if n.startswith("attn_"): return "attention"
        if n.startswith("ln_"): return "layernorm"
        if n.startswith("at::") or "elementwise" in n or "rocclr" in n or "reduce_kernel<" in n or "CatArray" in n: return "aten_runtime"
        return "other"
    for n, c, a, ms, p in rows:
        g = grp.setdefault(which(n), {"ms_per_step": 0.0, "launches": 0})
        g["ms_per_step"] += float(ms); g["launches"] += int(float(c))
    nl, fwd, bwd = groupnorm_bytes()
    g = grp.get("groupnorm")
    if g:
        g.update(layers=nl, algorithmic_bytes_per_step=fwd + bwd, gb_per_s=round((fwd + bwd) / g["ms_per_step"] / 1e6, 1),
                 frac_hbm_peak=round((fwd + bwd) / g["ms_per_step"] / 1e6 / 8000.0, 3))
    for v in grp.values():
        v["ms_per_step"] = round(v["ms_per_step"], 3)
    out = {"how": "single-stream steady-state rocprofv3 summary (MAED_WGRAD_SIDE_STREAM=0), scripts/group_rooflines.py", "groups": grp}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 2:
        t = json.load(open(sys.argv[2])); t["__groups__"] = out; json.dump(t, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
