"""Per-step wall time + allocator statistics over many train steps, as bench.py runs them: are slow steps allocator growth (device mallocs after
record_stream-delayed frees), MIOpen, or the box?   python scripts/diag_step_jitter.py [steps]"""
import os, sys, time, json, torch
os.environ.setdefault("MAED_SYNTHETIC_SMPL_OK", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maed_amd.ddp import ParamArena, GradBucketer, FusedAdam
from maed_amd.loss import LossVideo
dev = torch.device("cuda:0")
model = bench.build_model(torch.bfloat16, dev).train()
arena = ParamArena(model); buck = GradBucketer(arena, model); opt = FusedAdam(arena, lr=1e-4, weight_decay=1e-5, bucketer=buck)
gen = torch.Generator().manual_seed(0)
clip = torch.randn(bench.CFG["clips"], 16, 3, 224, 224, generator=gen).to(dev)
tgt = bench.make_targets(bench.CFG["clips"], 16, dev, gen)
criterion = LossVideo(**bench.LOSS_W)
def step():
    opt.zero_grad(); loss, _ = criterion(model(clip), tgt, None); loss.backward(); opt.step()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rows = []
for i in range(n):
    torch.cuda.synchronize(); s0 = torch.cuda.memory_stats(); t = time.perf_counter()
    step()
    torch.cuda.synchronize(); dt = 1e3 * (time.perf_counter() - t); s1 = torch.cuda.memory_stats()
    rows.append((i, dt, s1["reserved_bytes.all.current"] / 2**30, s1["num_device_alloc"] - s0["num_device_alloc"], s1["num_device_free"] - s0["num_device_free"],
                 s1["num_alloc_retries"] - s0["num_alloc_retries"]))
for r in rows:
    flag = "  <-- slow" if r[1] > 1.3 * sorted(x[1] for x in rows)[len(rows) // 2] else ""
    if r[0] < 6 or flag or r[3] or r[4]:
        print(f"step {r[0]:3d}: {r[1]:7.2f} ms  reserved {r[2]:6.2f} GiB  device mallocs {r[3]} frees {r[4]} retries {r[5]}{flag}")
med = sorted(x[1] for x in rows)[len(rows) // 2]
print(f"median {med:.2f} ms; slow steps (> 1.3 x median): {sum(1 for x in rows if x[1] > 1.3 * med)} of {n}; steps with a device malloc after step 5: {sum(1 for x in rows[6:] if x[3])}")
