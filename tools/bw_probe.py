#!/usr/bin/env python3
"""Calibrate what this MI355X delivers for plain streaming kernels (torch ops), to put the
kernels' algorithmic GB/s in context."""
import torch
dev = torch.device("cuda", 0)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for mb in (160, 480, 1280, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    ms = t(lambda: b.copy_(a))
    print(f"copy  {mb:5d} MB: {ms:.3f} ms  {2*mb/1024/ms*1e3/1e3:.2f} TB/s (r+w)")
    ms = t(lambda: a.sum())
    print(f"sum   {mb:5d} MB: {ms:.3f} ms  {mb/1024/ms*1e3/1e3:.2f} TB/s (read)")
    ms = t(lambda: b.fill_(1.0))
    print(f"fill  {mb:5d} MB: {ms:.3f} ms  {mb/1024/ms*1e3/1e3:.2f} TB/s (write)")
