#!/usr/bin/env python3
"""Bank conflicts of the fbank kernel's FFT strip (256 complex values, 8-byte LDS accesses: two groups of 32 lanes over 32
bank pairs) under round 1's padding e + (e >> 2) and under the XOR swizzle of fbank.hip.h::pz, for every access pattern of the
four radix-4 stages and the real-FFT untangle.  Prints the extra LDS cycles per frame (sum over the 32 instructions).

Why 6 and 25: write p(e) = e ^ c0 [bit 5 of e] ^ c1 [bit 6 of e] with 5-bit constants.  q = 16 (two 16-runs 64 apart per group):
bit 4 of c1 must be set.  q = 4 (4-runs 16 apart): bits 2..3 of c0 and c1 must be independent.  q = 1 (stride 4) and the
digit-reversed stores: bits 0..1 of c0, c1 and c0 ^ c1 non-zero, bit 1 of c0 set.  c0 = 0b00110, c1 = 0b11001."""
import numpy as np

lane = np.arange(64)


def patterns():
    P = []
    for r in range(4): P.append(("stage 0 stores", lane + 64 * r))
    for r in range(4): P.append(("q = 16", (lane // 16) * 64 + lane % 16 + 16 * r))
    for r in range(4): P.append(("q = 4", (lane // 4) * 16 + lane % 4 + 4 * r))
    for r in range(4): P.append(("q = 1 loads", 4 * lane + r))
    rev3 = ((lane & 3) << 4) | (lane & 12) | (lane >> 4)
    for r in range(4): P.append(("last stores", rev3 + 64 * r))
    for m in range(4):
        P.append(("untangle k", lane + 64 * m))
        P.append(("untangle 256 - k", (256 - lane - 64 * m) & 255))
    return P


def cost(p):
    tot = 0
    for _, e in patterns():
        pos = p[e]
        for g in (slice(0, 32), slice(32, 64)):
            b = pos[g] % 32
            tot += max(len(np.unique(pos[g][b == bank])) for bank in np.unique(b)) - 1
    return tot


e = np.arange(256)
print("padding e + (e >> 2):", cost(e + (e >> 2)), "extra cycles per frame")
p = e ^ (((e >> 5) & 1) * 6) ^ (((e >> 6) & 1) * 25)
assert len(np.unique(p)) == 256
print("xor swizzle         :", cost(p), "extra cycles per frame")
