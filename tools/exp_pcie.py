"""Raw PCIe rates of this box (pinned memory, 512 MB): H2D, D2H, and both at once on two streams."""
import time, torch
n = 512 << 20
h1 = torch.empty(n, dtype=torch.uint8).pin_memory(); h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
d1 = torch.empty(n, dtype=torch.uint8, device="cuda"); d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
a = t(lambda: d1.copy_(h1, non_blocking=True)); print("H2D %.1f GB/s" % (n / a / 1e9))
b = t(lambda: h2.copy_(d2, non_blocking=True)); print("D2H %.1f GB/s" % (n / b / 1e9))
def both():
    with torch.cuda.stream(s1): d1.copy_(h1, non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
c = t(both); print("both at once: %.1f GB/s each, %.1f GB/s aggregate" % (n / c / 1e9, 2 * n / c / 1e9))
