"""From a rocprofv3 kernel trace of `bench.py --graph-leg`: per train step (adam_kernel ends a step) the wall time, the sum of kernel durations and the time during which two
or more kernels ran at once -- for the eager steps and for the replayed ones (the last `steps` steps of the trace are replays).  usage: graph_overlap.py <kernel_trace.csv> <n_replays>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nrep = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["End_Timestamp"]) for r in rows if "adam_kernel" in r["Kernel_Name"]]
def step_stats(t0, t1):
    ev = []
    tot = 0
    n = 0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t0 < s <= t1:
            ev.append((s, 1)); ev.append((e, -1)); tot += e - s; n += 1
    ev.sort()
    depth, last, busy, multi = 0, None, 0, 0
    for t, d in ev:
        if last is not None and depth > 0:
            busy += t - last
            if depth > 1:
                multi += t - last
        depth += d; last = t
    return (t1 - t0) / 1e6, tot / 1e6, busy / 1e6, multi / 1e6, n
print(f"{len(marks)} optimizer steps in the trace; the last {nrep} are replays of the captured graph")
for name, idx in (("eager (same entry points)", range(len(marks) - nrep - 12, len(marks) - nrep - 6)), ("graph replay", range(len(marks) - 6, len(marks)))):
    st = [step_stats(marks[i - 1], marks[i]) for i in idx]
    m = lambda k: sum(s[k] for s in st) / len(st)
    print(f"{name:28s}: wall {m(0):7.3f} ms/step   sum of kernel durations {m(1):7.3f}   GPU busy {m(2):7.3f}   two or more kernels at once {m(3):6.3f} ms   launches {m(4):.0f}")
