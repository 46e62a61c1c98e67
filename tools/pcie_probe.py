#!/usr/bin/env python3
"""What the host link gives: pinned H2D / D2H of 256 MiB alone and both at once (two streams), pageable for comparison."""
import time, torch
n = 256 << 20
dev = torch.device("cuda:0")
hp = torch.empty(n, dtype=torch.uint8).pin_memory(); hp2 = torch.empty(n, dtype=torch.uint8).pin_memory()
hg = torch.empty(n, dtype=torch.uint8)
d1 = torch.empty(n, dtype=torch.uint8, device=dev); d2 = torch.empty(n, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(f, reps=5):
    f(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
print("pinned H2D  %.1f GB/s" % (n / t(lambda: d1.copy_(hp, non_blocking=True)) / 1e9))
print("pinned D2H  %.1f GB/s" % (n / t(lambda: hp.copy_(d1, non_blocking=True)) / 1e9))
def both():
    with torch.cuda.stream(s1): d1.copy_(hp, non_blocking=True)
    with torch.cuda.stream(s2): hp2.copy_(d2, non_blocking=True)
print("both at once: %.1f GB/s each way" % (n / t(both) / 1e9))
for chunk in (4 << 20, 16 << 20, 64 << 20):
    def pieces():
        for a in range(0, n, chunk): d1[a:a + chunk].copy_(hp[a:a + chunk], non_blocking=True)
    print("pinned H2D in pieces of %d MiB: %.1f GB/s" % (chunk >> 20, n / t(pieces) / 1e9))
print("pageable H2D %.1f GB/s, D2H %.1f GB/s" % (n / t(lambda: d1.copy_(hg)) / 1e9, n / t(lambda: hg.copy_(d1)) / 1e9))
