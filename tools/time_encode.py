#!/usr/bin/env python3
"""normalise + encode of N rows alone (one GPU): where the last millisecond of the step goes"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from pangaea_amd.data import Data  # noqa: E402
from pangaea_amd.models.VAENET import VAENET  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
dev = torch.device("cuda:0")
abd = torch.randint(0, 50, (n, 400), dtype=torch.int32, device=dev)
tnf = torch.randint(0, 500, (n, 136), dtype=torch.int32, device=dev)
names = np.arange(n)
torch.manual_seed(1)
vae = VAENET(400, 136, 32, 30, 1, True, 1, 0.005, 0.2, 0.1, 0.015, 0.0001)
vae.network.eval()


def timed(f, reps=20):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


d = Data(names, abd, tnf)
print(f"Data (normalise + weights): {timed(lambda: Data(names, abd, tnf)):.3f} ms")
print(f"encode: {timed(lambda: vae.encode(d)):.3f} ms")
with torch.no_grad():
    x = torch.cat([d.abd_dev, d.tnf_dev], dim=1)
    w = torch.randn(536, 512, device=dev)
    print(f"one 50k x 536 x 512 fp32 matmul: {timed(lambda: x @ w):.3f} ms")
