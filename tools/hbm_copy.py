#!/usr/bin/env python3
"""achievable HBM streaming rates on this GPU (report the measured copy bandwidth next to the 8 TB/s peak):
device-to-device copy (read + write), fill (write only), sum (read only) over buffers far larger than the caches"""
import time

import torch

dev = torch.device("cuda:0")
n = 1 << 31                                    # 8 GiB of int32 per buffer
a = torch.empty(n, dtype=torch.int32, device=dev).fill_(1)
b = torch.empty_like(a)


def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best


nbytes = n * 4
print(torch.cuda.get_device_name(0))
print(f"copy  (R+W) {2 * nbytes / timed(lambda: b.copy_(a)) / 1e12:5.2f} TB/s")
print(f"fill  (W)   {nbytes / timed(lambda: b.fill_(3)) / 1e12:5.2f} TB/s")
print(f"sum   (R)   {nbytes / timed(lambda: a.sum()) / 1e12:5.2f} TB/s   (torch's reduction, not a bandwidth test; bucket_fill_kernel reads 8.6 GB in 1.42 ms = 6.0 TB/s)")
