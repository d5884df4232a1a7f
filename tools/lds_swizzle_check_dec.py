#!/usr/bin/env python3
"""Bank-conflict model of cqt_decimate2_kernel's LDS traffic (cqt_dec.hip) and the search that found its swizzle.

Planes are stored in 16-byte granules; granule G lives at slot G ^ (((G >> 4) & 3) << 1).
  * Fragment reads (ds_read_b128, four non-contiguous 16-lane groups, slot = granule % 16): lane (q = lane & 15, g = lane >> 4) of the
    multiplying wave W reads granule 128 W + 8 q + 4 ks + g at step ks = 0 .. 11.
  * Plane stores of the producing waves (ds_write_b64, contiguous 16-lane groups, bank = (byte / 4) % 32): lane l writes the 8 bytes of piece
    idx = base + l, i.e. half (idx & 1) of granule idx >> 1.
Prints the worst multiplicity of both for the identity layout and for the swizzle, then every XOR-linear swizzle of the low four granule
bits by bits 4 .. 7 that makes the reads conflict-free (the kernel uses the first: columns (2, 4, 0, 0))."""
import itertools

G0 = list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28))
G1 = list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))
READ_GROUPS = [G0, G1, [l + 32 for l in G0], [l + 32 for l in G1]]


def reads(swz):
    worst = 0
    for wave in range(4):
        for ks in range(12):
            for grp in READ_GROUPS:
                slots = {}
                for l in grp:
                    p = swz(128 * wave + 8 * (l & 15) + 4 * ks + (l >> 4))
                    slots.setdefault(p % 16, set()).add(p)
                worst = max(worst, max(len(v) for v in slots.values()))
    return worst


def writes(swz):
    worst = 0
    for base in range(0, 1152, 16):                       # 16 consecutive pieces per lane group
        banks = {}
        for l in range(16):
            idx = base + l
            a = swz(idx >> 1) * 16 + (idx & 1) * 8
            for w in (0, 4):
                banks.setdefault(((a + w) // 4) % 32, set()).add(a + w)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def linear(cols):
    def swz(g):
        x = 0
        for b in range(4):
            if (g >> (4 + b)) & 1:
                x ^= cols[b]
        return g ^ x
    return swz


if __name__ == '__main__':
    ident = lambda g: g
    used = lambda g: g ^ (((g >> 4) & 3) << 1)
    print(f'identity layout: reads {reads(ident)}-way, stores {writes(ident)}-way')
    print(f'G ^ (((G >> 4) & 3) << 1): reads {reads(used)}-way, stores {writes(used)}-way')
    good = [cols for cols in itertools.product(range(16), repeat=4) if reads(linear(cols)) == 1]
    print(f'{len(good)} conflict-free XOR-linear swizzles; the first few: {good[:6]}')
