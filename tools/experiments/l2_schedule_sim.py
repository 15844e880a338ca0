#!/usr/bin/env python3
"""Lock-step model of mac_kernel's XCD schedule (no GPU): how many block-spectrum row pieces an XCD's L2 would have to
fetch from HBM if every workgroup walked one row per tick, for the BASELINE configs[2] request set.

One XCD, one chunk group: workgroups are dispatched in the kernel's order (for item: for chunk in chunk_group), `slots`
of them in flight, each reading the row piece (row, chunk) of its current position; an LRU of `cap` pieces (512 B each)
stands for the L2.  Prints L2 -> CU volume, HBM volume and the sharing factor per (cap, chunk_group), and the rows x class
work of the item grouping (request order against window-start order).
Measured on the MI355X (tools/gpu_cg_fetch.sh): 30.8 / 18.3 / 47.5 GB for chunk_group 1 / 8 / 32, of which 3.6 GB are the
pattern spectra: 27 / 15 / 44 GB of rows.  The model with 2048 pieces (1 MB of the 4 MB L2 for the rows; `Y` and the pattern
spectra pass through the same cache) gives 29 / 12 / 39 GB: the schedule behaves as modelled, and with these sizes
chunk_group 4 .. 8 is the best it can do -- a larger share of the L2 (4096 pieces: 4 GB) is what would cut the misses."""
import os
import sys
from collections import OrderedDict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sushi_amd import synth  # noqa: E402

RATE, SECONDS, WINDOW, N_EVENTS, OFFSET = 12000, 7200.0, 120.0, 3000, 7.25
SEED = 20260924 + 2
B, STEP = 4096, 6
PAD = 10 * RATE
L_DATA = int(20 * RATE + SECONDS * RATE)


def requests():
    events = synth.make_events(N_EVENTS, SECONDS, WINDOW + OFFSET, seed=SEED + 2)
    rng = np.random.default_rng(SEED + 3)
    out = []
    for (s, e) in events:
        c = s + OFFSET + float(rng.uniform(-WINDOW * 0.5, WINDOW * 0.5))
        m = int(RATE * e) - int(RATE * s)
        st = max(min(c - WINDOW, SECONDS), -10)
        en = max(min(c + WINDOW, SECONDS + 10), 0)
        a = int(RATE * st) + PAD
        b = int(RATE * en) + PAD + m
        out.append((a, min(b, L_DATA) - a - m + 1, m))
    return out


def layout(a, p, m):
    pair0 = a // (STEP * B)
    return pair0, (a + p - 1) // (STEP * B) - pair0 + 1, -(-m // B)


def group_range(smax, lo, hi):
    return (STEP * lo) // smax * smax, (STEP * (hi - 1) + smax - 1) // smax * smax


def items(reqs, by_start=True):
    its = []
    for c, smax in enumerate((6, 12, 18)):
        idx = [k for k, r in enumerate(reqs) if min(2, (layout(*r)[2] - 1) // 6) == c]
        if by_start:
            idx.sort(key=lambda k: reqs[k][0])
        for i in range(0, len(idx), 8):
            mem = idx[i:i + 8]
            lay = [layout(*reqs[k]) for k in mem]
            g0 = min(group_range(smax, l[0], l[0] + l[1])[0] for l in lay)
            g1 = max(group_range(smax, l[0], l[0] + l[1])[1] for l in lay)
            its.append((reqs[mem[0]][0], g0, g1 + smax, smax))
    its.sort(key=lambda t: t[0])
    return its


def simulate(its, cg, slots=96, cap=4096):
    wgs = [(i, c) for i in range(len(its)) for c in range(cg)]
    nxt, active, cache, miss, acc = 0, [], OrderedDict(), 0, 0
    while nxt < len(wgs) or active:
        while len(active) < slots and nxt < len(wgs):
            i, c = wgs[nxt]
            nxt += 1
            active.append([c, its[i][1], its[i][2]])
        alive = []
        for w in active:
            key = (w[1], w[0])
            acc += 1
            if key in cache:
                cache.move_to_end(key)
            else:
                miss += 1
                cache[key] = 1
                if len(cache) > cap:
                    cache.popitem(last=False)
            w[1] += 1
            if w[1] < w[2]:
                alive.append(w)
        active = alive
    return miss, acc


if __name__ == "__main__":
    reqs = requests()
    for by_start in (False, True):
        its = items(reqs, by_start)
        print("items %s: %d, rows x class = %d" % ("by window start" if by_start else "in request order", len(its),
                                                  sum((t[2] - t[1]) * t[3] for t in its)))
    its = items(reqs, True)
    for cap in (2048, 4096, 8192):
        for cg in (1, 2, 4, 8, 16, 32):
            m, a = simulate(its, cg, cap=cap)
            scale = 512 * (32 // cg) * 8 / 1e9
            print("L2 pieces %5d  chunk_group %2d   L2->CU %5.1f GB   HBM %5.1f GB   sharing %.2f" % (cap, cg, a * scale, m * scale, a / m))
