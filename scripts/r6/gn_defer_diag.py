import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from maed_amd import ops
dev = torch.device("cuda")
dtype = torch.bfloat16
def run(defer, N=5, C=256, H=14, W=14, twice=True):
    ops.GN_DEFER_AFFINE = defer
    torch.manual_seed(0)
    g = (1 + 0.2 * torch.randn(C)).to(dev).requires_grad_(True); b = (0.1 * torch.randn(C)).to(dev).requires_grad_(True)
    abs_ = []
    for li in range(2 if twice else 1):
        torch.manual_seed(10 + li)
        x = (torch.randn(N, C, H, W) * 1.5 + 0.2).to(dev).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        dy = torch.randn(N, C, H, W).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
        ab = torch.zeros(N, C, 2, dtype=torch.float32, device=dev); sums = torch.zeros(N, 32, 2, dtype=torch.float64, device=dev)
        sync = torch.zeros(N * ops.GN_SYNC_WORDS, dtype=torch.int32, device=dev)
        y = ops.GroupNormFn.apply(x, None, g, b, 1e-5, True, True, sums, ab, False, False, sync)
        y.backward(dy)
        abs_.append(ab)
    ops.gn_affine_flush(dev)
    torch.cuda.synchronize()
    return g.grad.clone(), b.grad.clone(), [a.clone() for a in abs_]
for twice in (False, True):
    r = [run(d, twice=twice) for d in (True, False, False, True)]
    for i in range(1, 4):
        print("twice", twice, "run", i, "vs 0: dgamma equal", torch.equal(r[0][0], r[i][0]), "dbeta equal", torch.equal(r[0][1], r[i][1]),
              "ab equal", [torch.equal(a, c) for a, c in zip(r[0][2], r[i][2])], "max diff", (r[0][0] - r[i][0]).abs().max().item(), (r[0][1] - r[i][1]).abs().max().item())
    print("manual colsum vs kernel (run 0):", (sum(a.sum(0)[:, 1] for a in r[0][2]) - r[0][0]).abs().max().item())
