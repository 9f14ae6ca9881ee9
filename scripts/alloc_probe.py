import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from maed_amd.ddp import FusedAdam, GradBucketer, ParamArena
from maed_amd.loss import LossVideo
from maed_amd import ops
dev = torch.device("cuda", 0)
model = bench.build_model(torch.bfloat16, dev).train()
arena = ParamArena(model); opt = FusedAdam(arena, lr=1e-4, weight_decay=1e-5, bucketer=GradBucketer(arena, model)); crit = LossVideo(**bench.LOSS_W)
gen = torch.Generator().manual_seed(0); C = bench.CFG
clip = torch.randn(C["clips"], C["T"], 3, C["img"], C["img"], generator=gen).to(dev); tgt = bench.make_targets(C["clips"], C["T"], dev, gen)
def step():
    opt.zero_grad(); loss, _ = crit(model(clip), tgt, None); loss.backward(); opt.step()
def stats(tag):
    s = torch.cuda.memory_stats(dev)
    print(tag, "dev_alloc", s["num_device_alloc"], "dev_free", s["num_device_free"], "reserved MB", s["reserved_bytes.all.current"] >> 20, "active MB", s["active_bytes.all.current"] >> 20,
          "small-pool segs", s["segment.small_pool.current"], "large-pool segs", s["segment.large_pool.current"], "retries", s["num_alloc_retries"], flush=True)
for i in range(12):
    step(); torch.cuda.synchronize(); stats(f"step {i}")
side_on = ops._SIDE_ON
ops._SIDE_ON = False
for i in range(4):
    step(); torch.cuda.synchronize(); stats(f"no-side step {i}")
