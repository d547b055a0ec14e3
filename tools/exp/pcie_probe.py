#!/usr/bin/env python3
"""Host <-> device copy rates of this box (pinned memory, one stream), the ceiling of every PCIe-inclusive number."""
import json
import time

import torch

out = {}
for mb in (2, 8, 64, 256):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for name, (src, dst) in {"h2d": (h, d), "d2h": (d, h)}.items():
        for _ in range(3):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        reps = max(4, 512 // mb)
        t0 = time.perf_counter()
        for _ in range(reps):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        out[f"{name}_{mb}MB_GBps"] = round(n * reps / (time.perf_counter() - t0) / 1e9, 2)
print(json.dumps(out))
