#!/usr/bin/env python3
"""diagnostics: practical memory-bandwidth ceilings of the box (torch ops): copy, read, write, 16-byte gathers."""
import torch, time
dev = torch.device("cuda:0")
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.time() - t) / n
for mb in (40, 100, 1000, 5000):
    n = mb * 1000 * 1000 // 8
    a = torch.rand(n, device=dev, dtype=torch.float64); b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a)); print(f"copy   {mb:5d} MB: {2*n*8/t/1e9:8.1f} GB/s (read+write)")
    t = timeit(lambda: a.sum());    print(f"read   {mb:5d} MB: {n*8/t/1e9:8.1f} GB/s")
    t = timeit(lambda: b.fill_(1.0)); print(f"write  {mb:5d} MB: {n*8/t/1e9:8.1f} GB/s")
    t = timeit(lambda: torch.add(a, b, out=b)); print(f"triad  {mb:5d} MB: {3*n*8/t/1e9:8.1f} GB/s (2 reads + 1 write)")
    rows = a.view(-1, 2)
    idx = torch.randint(0, rows.shape[0], (50_000_000,), device=dev)
    out = torch.empty(idx.shape[0], 2, device=dev, dtype=torch.float64)
    t = timeit(lambda: torch.index_select(rows, 0, idx, out=out), 3)
    print(f"gather {mb:5d} MB table, 16-B rows: {idx.shape[0]*16/t/1e9:8.1f} GB/s useful (+{idx.shape[0]*24/t/1e9:.0f} GB/s of index reads and output writes)")
    del a, b, rows, idx, out
