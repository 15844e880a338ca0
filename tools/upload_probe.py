#!/usr/bin/env python3
"""GPU: what does getting a 347 MB float32 stream into HBM cost -- pageable copy, registered (pinned in place) copy, pinned staging?"""
import time, sys, os
import numpy as np
import torch
n = 86_640_000
x = np.random.default_rng(0).random(n, dtype=np.float32)
dev = torch.device("cuda", 0)
torch.cuda.synchronize()
def t(f, reps=3):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); out.append((time.perf_counter() - t0) * 1e3); del r
    return [round(v, 1) for v in out]
print("pageable .to():", t(lambda: torch.from_numpy(x).to(dev)))
rt = torch.cuda.cudart()
def registered():
    rc = rt.cudaHostRegister(x.ctypes.data, x.nbytes, 0)
    try:
        return torch.from_numpy(x).to(dev, non_blocking=True)
    finally:
        torch.cuda.synchronize(); rt.cudaHostUnregister(x.ctypes.data)
print("register + copy + unregister:", t(registered))
stage = torch.empty(32 << 20, dtype=torch.uint8).pin_memory()
def staged():
    out = torch.empty(n, dtype=torch.float32, device=dev)
    ob = out.view(torch.uint8); xb = torch.from_numpy(x).view(torch.uint8)
    step = stage.numel() // 2
    for k, o in enumerate(range(0, xb.numel(), step)):
        h = stage[(k & 1) * step:(k & 1) * step + min(step, xb.numel() - o)]
        if k >= 2: torch.cuda.current_stream().synchronize() if False else None
        h.copy_(xb[o:o + h.numel()])
        ob[o:o + h.numel()].copy_(h, non_blocking=True)
        torch.cuda.current_stream().synchronize()
    return out
print("16 MB pinned staging (sync each):", t(staged))
