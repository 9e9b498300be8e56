#!/usr/bin/env python3
"""Loop a 1 GiB device copy (torch) for N seconds: the power / clock reference of plain streaming."""
import sys, time, torch
dev = torch.device("cuda:0"); secs = float(sys.argv[1])
a = torch.empty(1 << 28, device=dev).uniform_(); b = torch.empty_like(a)
torch.cuda.synchronize(); t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(50): b.copy_(a)
    torch.cuda.synchronize(); n += 50
dt = (time.time() - t0) / n
print("copy 1 GiB: %.1f us per launch, %.2f TB/s read + write" % (dt * 1e6, 2 * a.numel() * 4 / dt / 1e12))
