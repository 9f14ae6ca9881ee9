#!/bin/bash
# HBM traffic per launch of the roofline kernels from PMC counters -> gpurun_out/pmc/traffic.json, stamped with the kernel-source hash
# (bench.py quotes it only for the build it was measured on).  Separate passes for FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md:
# FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2 -> not both in one pass; --kernel-trace only).  On gfx950 FETCH_SIZE reports 1/2 of a wide
# coalesced stream: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Copy the result to profiles/rNN_pmc/traffic.json.
set -u
cd "$(dirname "$0")/.."
ROOT=$(pwd); mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
pass() {   # tag counter command...
  tag=$1; c=$2; shift 2; rm -rf /tmp/pmc_${tag}_$c
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$c -o p -- "$@" > "$ROOT/gpurun_out/pmc/${tag}_$c.log" 2>&1)
  f=$(find /tmp/pmc_${tag}_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/pmc/${tag}_$c.csv || echo "$tag $c: no counter output"
}
for c in FETCH_SIZE WRITE_SIZE; do
  pass attn $c python "$ROOT/scripts/attn_micro.py" 8
  pass gemm $c python "$ROOT/scripts/gemm_micro.py" 3 ste 0
  WGRAD_STE=1 pass wgrad $c python "$ROOT/scripts/wgrad_micro.py" 1
  # the roofline kernel over EVERY launch of the train step (what bench.py's roofline.achieved averages over): the bench itself, a few steps
  MAED_WGRAD_SIDE_STREAM=0 pass step $c python "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline
done
python - <<'PY'
import csv, json, os, sys
sys.path.insert(0, os.getcwd())
from maed_amd.build import source_hash
def mean(tag, counter, pats):
    fn = f"gpurun_out/pmc/{tag}_{counter}.csv"
    if not os.path.exists(fn):
        return None, 0
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(fn)) if r.get("Counter_Name") == counter and any(p in r.get("Kernel_Name", "") for p in pats)]
    return (sum(v) / len(v) if v else None), len(v)
out = {"source_hash": source_hash(), "dtype": "bf16", "workload": "cfg3",
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (--kernel-trace only) over scripts/attn_micro.py, gemm_micro.py ste, "
                 "wgrad_micro.py (WGRAD_STE=1); bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024, averaged over the launches of the kernel family "
                 "(gfx950: FETCH_SIZE reports 1/2 of a wide coalesced stream, MI355X_MICROARCH.md HBM section)", "kernels": {}, "detail": {}}
for key, tag, pats in (("attn_spatial_fwd", "attn", ["attn_sp_fwd", "attn_long_fwd"]), ("gemm_nt", "gemm", ["gemm_nt_glds", "gemm_nt_256"]), ("gemm_tn_ste_shapes", "wgrad", ["gemm_tn_mfma", "gemm_tn_dma"]),
                       ("gemm_tn", "step", ["gemm_tn_mfma_bf16_kernel<false", "gemm_tn_mfma_bf16_kernelILb0", "gemm_tn_dma_bf16_kernel"])):   # gemm_tn: every launch inside the train step
    f, nf = mean(tag, "FETCH_SIZE", pats); w, nw = mean(tag, "WRITE_SIZE", pats)
    out["detail"][key] = {"FETCH_SIZE_KB_mean": f, "WRITE_SIZE_KB_mean": w, "launches": [nf, nw]}
    out["kernels"][key] = None if f is None or w is None else int((2.0 * f + w) * 1024)
json.dump(out, open("gpurun_out/pmc/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
