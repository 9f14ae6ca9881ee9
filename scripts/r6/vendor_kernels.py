"""Which hipBLASLt / rocBLAS kernels does the vendor library pick at the GEMM shapes of the benchmarked step?  ("nobody has looked at what the
vendor kernel is".)  Run under `rocprofv3 --kernel-trace`: every shape is preceded by a marker fill of 4096 * (index + 1) elements, so the trace (dispatch order)
maps kernel names -- which spell macro tile, depthU, direct-to-LDS, prefetch depths, stream-K -- to shapes.  scripts/r6/vendor_kernels_parse.py reads the trace."""
import sys

import torch

M = 128 * 197
SHAPES = [("NT qkv", M, 1536, 512, "nt"), ("NT fc1", M, 2048, 512, "nt"), ("NT fc2", M, 512, 2048, "nt"), ("NT proj", M, 512, 512, "nt"),
          ("NT dqkv", M, 512, 1536, "nt"), ("NT dfc1", M, 512, 2048, "nt"), ("NT sq4k", 4096, 4096, 4096, "nt"), ("NT sq8k", 8192, 8192, 8192, "nt"),
          ("NT s3 1024>256", 25088, 256, 1024, "nt"), ("NT embed", 25088, 512, 1024, "nt"),
          ("TN w qkv", M, 1536, 512, "tn"), ("TN w fc1", M, 2048, 512, "tn"), ("TN w fc2", M, 512, 2048, "tn"), ("TN w proj", M, 512, 512, "tn")]
if __name__ == "__main__":
    torch.manual_seed(0)
    for i, (name, m, n, k, kind) in enumerate(SHAPES):
        if kind == "nt":
            A = torch.randn(m, k, device="cuda").bfloat16()
            B = torch.randn(n, k, device="cuda").bfloat16()
            out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
            fn = lambda: torch.mm(A, B.t(), out=out)
        else:
            Y = torch.randn(m, n, device="cuda").bfloat16()
            X = torch.randn(m, k, device="cuda").bfloat16()
            out = torch.empty(n, k, device="cuda", dtype=torch.bfloat16)
            fn = lambda: torch.mm(Y.t(), X, out=out)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        torch.empty(4096 * (i + 1), device="cuda").fill_(1.0)       # marker
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        print(i, name, m, n, k, flush=True)
