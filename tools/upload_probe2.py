#!/usr/bin/env python3
"""GPU: WHAT is the 0.15 s the first large host-to-device copy of a process costs (tools/upload_probe.py)?  Each variant in a
process of its own: (a) a large pageable copy first; (b) a 1 MB copy first, then the large one, then another large buffer;
(c) a device allocation of the destination's size first (the caching allocator's first large block), then the copy;
(d) the large copy from a buffer whose pages another large copy's source never touched."""
import subprocess
import sys

BODY = r'''
import time, sys
import numpy as np, torch
n = 86_640_000
dev = torch.device("cuda", 0)
x = np.random.default_rng(0).random(n, dtype=np.float32)
y = np.random.default_rng(1).random(n, dtype=np.float32)
torch.zeros(1, device=dev); torch.cuda.synchronize()
def ms(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return round((time.perf_counter() - t0) * 1e3, 1), r
v = sys.argv[1]
out = []
if v == "a":
    out.append(("large X", ms(lambda: torch.from_numpy(x).to(dev))[0]))
    out.append(("large Y", ms(lambda: torch.from_numpy(y).to(dev))[0]))
elif v == "b":
    out.append(("1 MB", ms(lambda: torch.from_numpy(x[:262144]).to(dev))[0]))
    out.append(("large X", ms(lambda: torch.from_numpy(x).to(dev))[0]))
    out.append(("large Y", ms(lambda: torch.from_numpy(y).to(dev))[0]))
elif v == "c":
    t, buf = ms(lambda: torch.empty(n, dtype=torch.float32, device=dev))
    out.append(("device alloc 347 MB", t))
    out.append(("copy X into it", ms(lambda: buf.copy_(torch.from_numpy(x)))[0]))
    t, buf2 = ms(lambda: torch.empty(n, dtype=torch.float32, device=dev))
    out.append(("second alloc", t))
    out.append(("copy Y into it", ms(lambda: buf2.copy_(torch.from_numpy(y)))[0]))
elif v == "d":
    t, buf = ms(lambda: torch.empty(2 * n, dtype=torch.float32, device=dev))
    out.append(("device alloc 694 MB", t))
    out.append(("fill (first touch on the device)", ms(lambda: buf.zero_())[0]))
    out.append(("copy X into its first half", ms(lambda: buf[:n].copy_(torch.from_numpy(x)))[0]))
    out.append(("copy Y into its second half", ms(lambda: buf[n:].copy_(torch.from_numpy(y)))[0]))
print(v, out)
'''

for v in "abcd":
    subprocess.run([sys.executable, "-c", BODY, v], check=False)
