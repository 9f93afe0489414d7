"""H2D bandwidth of the GPU box: pinned vs pageable, one copy of 400 MB (context for DESIGN.md 6, PCIe-inclusive rate)."""
import time, torch
n = 400 << 20
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, h in (("pinned", torch.empty(n, dtype=torch.uint8).pin_memory()), ("pageable", torch.empty(n, dtype=torch.uint8))):
    h.fill_(1)
    for it in range(3):
        torch.cuda.synchronize(); t = time.perf_counter(); d.copy_(h, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"{name} H2D {n / dt / 1e9:.1f} GB/s ({dt * 1e3:.1f} ms)")
