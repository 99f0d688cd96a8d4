#!/usr/bin/env python3
"""Known-byte-count streaming copy for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE reads 1/2 of a wide coalesced stream; WRITE_SIZE is
uncalibrated).  2 GiB -> 2 GiB device copy (well past the 256 MiB Infinity Cache), three times."""
import torch
n = 2 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda").fill_(7)
b = torch.empty_like(a)
torch.cuda.synchronize()
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
print("calib copy bytes", n)
