#!/usr/bin/env python3
"""Bank-conflict model of cqt_basis_kernel's operand reads (ds_read_b128 in four non-contiguous 16-lane groups, slot = (byte / 16) % 16):
lane (fl = lane & 15, g = lane >> 4) of k-step ks reads 16 bytes at sample fl * hop + 32 ks + 8 g of the staged copy, which carries `pad`
samples behind every `hop`.  Prints the worst multiplicity per hop for the old (8) and the new padding."""
g0 = list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28))
g1 = list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))
groups = [g0, g1, [l + 32 for l in g0], [l + 32 for l in g1]]


def worst(hop, pad, K=256):
    lh = hop.bit_length() - 1
    w = 0
    for ks in range(K // 32):
        for grp in groups:
            slots = {}
            for l in grp:
                fl, g = l & 15, l >> 4
                s = fl * hop + 32 * ks + 8 * g
                a = 2 * (s + pad * (s >> lh))
                slots.setdefault((a // 16) % 16, set()).add(a)
            w = max(w, max(len(v) for v in slots.values()))
    return w


for hop in (16, 32, 64, 128, 256):
    new = 16 if hop >= 32 else 0
    print(f'hop {hop:3d}: pad 8 -> {worst(hop, 8)}-way, pad {new} -> {worst(hop, new)}-way')
print('hop   8: pad 0 ->', worst(8, 0), '-way')
