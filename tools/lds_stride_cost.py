#!/usr/bin/env python3
"""LDS cycles per ds_read_b128 lane group of the conv kernels' B-operand reads as a function of the pixel stride (CPU only).

ds_read_b128 is serviced in four NON-contiguous 16-lane groups (MI355X_MICROARCH.md, LDS table); bank = (addr / 4) mod 64, i.e.
16 slots of 16 B.  Lane (g, p) of the 3x3 stride-1 conv reads pixel p at the (tap, channel chunk) of k-slot block 4s + g, so half
of a group reads one block and the other half the next one.  This script enumerates every k-step of every channel width in
use and prints the average number of LDS cycles per group (1.0 = conflict free) for candidate strides of k slots."""
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
          [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59], [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63]]


def cost(cs, k, rw=34):
    ktot = 9 * cs
    tot = n = 0
    for s in range((ktot + 31) // 32):
        toff = []
        for g in range(4):
            kk0 = (s * 4 + g) * 8
            tap, cc0 = divmod(kk0, cs)
            toff.append(((tap // 3) * rw + tap % 3) * k + cc0 // 8 if kk0 < ktot else 0)
        for grp in GROUPS:
            cnt = {}
            for lane in grp:
                slot = (k * (lane & 15) + toff[lane >> 4]) % 16
                cnt[slot] = cnt.get(slot, 0) + 1
            tot += max(cnt.values()); n += 1
    return tot / n


if __name__ == "__main__":
    for cs in (16, 24, 40, 48, 64, 80):
        npb = cs // 8
        print(cs, {k: round(cost(cs, k), 2) for k in range(npb, npb + 7)})
