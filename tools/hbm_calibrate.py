#!/usr/bin/env python3
"""What a plain streaming kernel reaches on this GPU (torch elementwise / reduction kernels), to put the tile kernels'
TB/s next to a locally measured ceiling instead of the 8 TB/s spec figure."""
import torch
dev = torch.device("cuda:0")
for mb in (256, 600, 2048):
    n = mb * 1024 * 1024 // 8
    x = torch.randn(n, dtype=torch.float64, device=dev)
    y = torch.empty_like(x)
    def timed(fn, reps=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps
    t_sum = timed(lambda: x.sum())
    t_copy = timed(lambda: y.copy_(x))
    t_scale = timed(lambda: torch.mul(x, 1.0001, out=y))
    print("%5d MB: read (sum) %.2f TB/s   copy %.2f TB/s (read+write)   scale %.2f TB/s" % (mb, mb * 1.048576e-3 / t_sum, 2 * mb * 1.048576e-3 / t_copy, 2 * mb * 1.048576e-3 / t_scale))
