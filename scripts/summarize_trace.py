"""Steady-state per-kernel summary from a rocprofv3 kernel trace CSV: keeps only the kernels launched
between the (n_skip)-th and the last optimizer step (adam_kernel marks the end of a train step), so MIOpen's
first-iteration algorithm search / naive kernels do not pollute the numbers.
usage: python scripts/summarize_trace.py <kernel_trace.csv> <out.csv> [n_skip=3]"""
import csv, re, sys
from collections import defaultdict
src, dst = sys.argv[1], sys.argv[2]
n_skip = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["End_Timestamp"]) for r in rows if "adam_kernel" in r["Kernel_Name"]]
assert len(marks) > n_skip + 1, f"only {len(marks)} optimizer steps in the trace"
t0, t1, nsteps = marks[n_skip - 1], marks[-1], len(marks) - n_skip
agg = defaultdict(lambda: [0, 0])
for r in rows:
    s = int(r["Start_Timestamp"])
    if t0 < s <= t1:
        a = agg[r["Kernel_Name"]]; a[0] += 1; a[1] += int(r["End_Timestamp"]) - s
tot = sum(v[1] for v in agg.values())
with open(dst, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "calls_per_step", "avg_us", "ms_per_step", "pct"])
    w.writerow([f"# steady state: {nsteps} train steps, wall {(t1 - t0) / 1e6 / nsteps:.3f} ms/step, sum of kernel time {tot / 1e6 / nsteps:.3f} ms/step"])
    for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = k
        for pat, rep in ((r"void ", ""), (r"at::native::", ""), (r"\(anonymous namespace\)::", ""), (r"<unnamed>::", "")):
            short = re.sub(pat, rep, short)
        short = re.sub(r"\((?:[^()]|\([^()]*\))*\)\s*$", "", short)   # drop the trailing argument list only
        short = short[:200]
        w.writerow([short, round(c / nsteps, 2), round(d / c / 1e3, 2), round(d / 1e6 / nsteps, 4), round(100.0 * d / tot, 2)])
print(f"steady state: {nsteps} steps, wall {(t1 - t0) / 1e6 / nsteps:.3f} ms/step, kernel time {tot / 1e6 / nsteps:.3f} ms/step -> {dst}")
# ordered kernel sequence of the LAST step (name truncated, duration us, gap to the previous kernel us): where the launches go
seq = [r for r in rows if marks[-2] < int(r["Start_Timestamp"]) <= marks[-1]]
with open(dst.replace(".csv", "_last_step_sequence.txt"), "w") as fh:
    prev_end = None
    for r in seq:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = re.sub(r"^void ", "", r["Kernel_Name"])[:70]
        fh.write(f"{(e - s) / 1e3:9.2f} {((s - prev_end) / 1e3 if prev_end else 0.0):8.2f}  {name}\n")
        prev_end = e
