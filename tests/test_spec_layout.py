"""The mel gather's slot layout (spec.hip: mel_assign_slots) without a GPU: amtx_spec_mel_layout is host-only.  The kernel reads
pb[slot_start + j] for tap j on all 64 lanes at once; a wave's LDS read is served in two 32-lane groups, one extra cycle per extra
distinct address on a bank (bank = dword address mod 32, MI355X_MICROARCH.md).  Dealt in row order the 229-row Slaney table cost 304
cycles per frame (44 % of the kernel's LDS cycles were bank conflicts); the matching must reach the conflict-free 128."""
import collections

import numpy as np
import pytest

from amt_tools_amd import _lib
from oracle import frontend_np as fe


def _layout(sr, n_fft, n_mels, htk):
    row = np.full(512, -9, np.int32)
    start = np.zeros(512, np.int32)
    rmax = np.zeros(8, np.int32)
    rounds = _lib.lib().amtx_spec_mel_layout(sr, n_fft, n_mels, int(htk), _lib.ptr(row), _lib.ptr(start), _lib.ptr(rmax))
    assert rounds == (n_mels + 63) // 64
    return rounds, row[:64 * rounds], start[:64 * rounds], rmax[:rounds]


def _gather_cycles(rounds, row, start, rmax):
    total = 0
    for r in range(rounds):
        for h in range(2):
            sl = slice(64 * r + 32 * h, 64 * r + 32 * h + 32)
            banks = collections.defaultdict(set)
            for s, i in zip(start[sl], row[sl]):
                if i >= 0:
                    banks[int(s) % 32].add(int(s))
            if banks:
                total += int(rmax[r]) * max(len(v) for v in banks.values())
    return total


@pytest.mark.parametrize('sr,n_fft,n_mels,htk', [(22050, 2048, 229, False), (16000, 2048, 229, True), (22050, 2048, 128, False),
                                                 (22050, 1024, 80, False), (22050, 2048, 512, False), (22050, 512, 40, False)])
def test_every_row_has_one_slot_that_fits_it(sr, n_fft, n_mels, htk):
    rounds, row, start, rmax = _layout(sr, n_fft, n_mels, htk)
    fb = fe.mel_filterbank(sr, n_fft, n_mels, htk=htk)
    assert sorted(row[row >= 0].tolist()) == list(range(n_mels))
    for s in range(64 * rounds):
        i = int(row[s])
        if i < 0:
            continue
        nz = np.nonzero(fb[i].astype(np.float32) > 1e-12)[0]      # (a Nyquist-bin weight of 1e-17 from a mel <-> Hz round trip is not a tap)
        lead = int(nz[0]) - int(start[s])
        assert lead >= 0 and lead % 2 == 0, (s, i, lead)               # even: the even / odd accumulator split of the gather is unchanged
        assert lead + (int(nz[-1]) - int(nz[0]) + 1) <= int(rmax[s // 64]), (s, i)


def test_baseline_table_is_conflict_free():
    rounds, row, start, rmax = _layout(22050, 2048, 229, False)
    assert rmax.tolist() == [4, 8, 20, 32]                              # the shape spec_power_ring_kernel<8, 4, 8, 20, 32> is built for
    assert _gather_cycles(rounds, row, start, rmax) == 2 * int(rmax.sum()) == 128
    # what row order cost (the layout of rounds 1 - 3)
    fb = fe.mel_filterbank(22050, 2048, 229)
    nat_start = np.array([int(np.nonzero(fb[i])[0][0]) for i in range(229)] + [0] * 27, np.int32)
    nat_row = np.array(list(range(229)) + [-1] * 27, np.int32)
    assert _gather_cycles(4, nat_row, nat_start, rmax) == 304
