"""The link between host and GPU by itself (not a pytest; run on the GPU box): 64 MiB of pinned memory up, down, and both
ways at once on two torch streams.  (The library's own pipelines overlap the two directions better than the last line
suggests: see DESIGN.md §6.)"""
import torch, time
n = 64 << 20
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, fn in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
    for _ in range(3): fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("%s 64 MiB pinned: %.2f ms = %.1f GiB/s" % (name, dt * 1e3, n / dt / 2**30))
s1 = torch.cuda.Stream(); s2 = torch.cuda.Stream()
h2 = torch.empty(n, dtype=torch.uint8).pin_memory(); d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    with torch.cuda.stream(s1): d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("both ways at once: %.2f ms per 64+64 MiB" % (dt * 1e3))
