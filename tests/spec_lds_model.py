#!/usr/bin/env python3
"""Bank-conflict model of the log-mel kernel's two irregular LDS access patterns (spec.hip: the lane-per-row mel gather and the power-row
writes of the untangling pass), per MI355X_MICROARCH.md's LDS rules (ds_read_b32 / ds_write_b32: two 32-lane groups, bank = dword address
mod 32, one extra cycle per extra distinct address on a bank, 2-way free on ds_write_b32).  CPU only; prints LDS cycles per frame for a few
power-row layouts and for a permuted row -> lane assignment.  Used to decide (round 3) that neither a padded power-row layout nor a
permutation of the mel rows inside their rounds removes the gather's conflicts: they come from the mel rows' irregular first bins.
    python tests/spec_lds_model.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import frontend_np as fe          # (under tests/: only tests, smoke() and the CPU baseline may import the oracle)

fb = fe.mel_filterbank(22050, 2048, 229)
starts = [int(np.nonzero(fb[r])[0][0]) for r in range(229)] + [0] * (256 - 229)
SLOTS = [4, 8, 20, 32]          # tap slots of the four rounds of 64 rows (spec_power_ring_kernel<8, 4, 8, 20, 32>)


def group_cycles(addrs, free=1):
    banks = {}
    for a in addrs:
        banks.setdefault(a % 32, set()).add(a)
    return max(free, max(len(v) for v in banks.values()))


def gather(pidx, order=None):
    tot = 0
    for r in range(4):
        rows = list(range(64 * r, 64 * r + 64)) if order is None else order[r]
        for j in range(SLOTS[r]):
            for half in range(2):
                tot += group_cycles([pidx(starts[i] + j) for i in rows[32 * half:32 * half + 32]])
    return tot


def untangle_writes(pidx):
    M, tot = 1024, 0
    for it in range(2):
        for d in range(4):
            for which in range(2):
                for half in range(2):
                    addrs = []
                    for l in range(32):
                        q = it * 64 + half * 32 + l
                        if q < 112:
                            c = q // 7; k1 = 1 + q - 7 * c
                        elif q < 120:
                            k1, c = 8, q - 112
                        elif q < 127:
                            k1, c = 0, q - 119
                        else:
                            k1, c = 0, 0
                        ks = [k1 + 16 * c + 256 * d, [0, 256, 128, 384][d]][q == 127]
                        addrs.append(pidx(ks if which == 0 else M - ks))
                    tot += group_cycles(addrs, free=2)
    return tot


print('power-row layout        mel gather (ideal 128)   untangling writes (floor 32)   [LDS cycles per frame]')
for name, f in [('k (shipped)', lambda k: k), ('k + (k >> 4)', lambda k: k + (k >> 4)), ('k + (k >> 5)', lambda k: k + (k >> 5)),
                ('k + (k >> 3)', lambda k: k + (k >> 3)), ('k + 5 (k >> 4)', lambda k: k + 5 * (k >> 4)), ('k + (k >> 2)', lambda k: k + (k >> 2))]:
    print(f'{name:22s} {gather(f):12d} {untangle_writes(f):28d}')
# rows permuted inside their round so that the two half-waves see as few equal residues as possible
order = []
for r in range(4):
    res = {}
    for i in range(64 * r, 64 * r + 64):
        res.setdefault(starts[i] % 32, []).append(i)
    a, b = [], []
    for _, v in sorted(res.items(), key=lambda kv: -len(kv[1])):
        for n, i in enumerate(v):
            t, o = (a, b) if n % 2 == 0 else (b, a)
            (t if len(t) < 32 else o).append(i)
    order.append(a + b)
print(f'rows permuted inside their rounds, linear layout: mel gather {gather(lambda k: k, order)}')
