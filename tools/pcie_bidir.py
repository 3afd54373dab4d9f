#!/usr/bin/env python3
"""PCIe rate of pinned 25-MB copies, one direction at a time and both at once (what bounds the host-image entry points)."""
import torch, time
n = 25 * 1024 * 1024
reps = 40
h_in = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
h_out = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
d_in = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(4)]
d_out = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(4)]
s_up, s_dn = torch.cuda.Stream(), torch.cuda.Stream()
def run(up, dn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        if up:
            with torch.cuda.stream(s_up):
                d_in[r % 4].copy_(h_in[r % 4], non_blocking=True)
        if dn:
            with torch.cuda.stream(s_dn):
                h_out[r % 4].copy_(d_out[r % 4], non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return reps * n / dt / 1e9
for _ in range(2):
    print("up only %.1f GB/s  down only %.1f GB/s  both: %.1f GB/s each direction" % (run(True, False), run(False, True), run(True, True)))
