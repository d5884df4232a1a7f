"""
Host-side helper processes for the one part of note decoding that has to stay on the host: the reference's row order.

`tools.multi_pitch_to_notes` and the two stacked-notes conversions behind `NoteTranscriber.estimate` each run `sort_notes`
(amt_tools/tools/utils.py:2713-2746 via :469, :745, :531) = `np.argsort` of the onset column with NumPy's default, UNSTABLE sort.  The
order among equal onsets (chords) is whatever that implementation -- an AVX-512 / AVX2 / scalar introsort, by CPU -- leaves behind, three
times over; it cannot be restated on the GPU, only repeated with the same NumPy.  Per clip that is three ~5 us argsorts plus call
overhead, ~25-40 us of single-threaded Python: at 8.5 ms of GPU time per 512 clips it bounds the batched transcription driver
(BASELINE config 5) at 13-15 M frames/s.  The argsorts of different clips are independent, so they are dealt to a few worker processes:

    parent:  onset column (E float64) + per-clip offsets  ->  one /dev/shm file (np.memmap)
    workers: `python -m amt_tools_amd._order_pool` children (started once, fed over pipes; they import numpy and nothing else -- no
             multiprocessing spawn, which would re-import the caller's __main__) write each clip's permutation, as global row indices
    parent:  ONE rows.take(perm) for the whole batch, then per-clip views

AMTX_NOTE_WORKERS=<n> sets the number of workers (default: min(4, cores // 4); 0 = order in-process); batches below 384 clips are ordered
in-process anyway (measured on the MI355X host: at 256 clips per batch the in-process loop already keeps up with the upload, 25.7 M
frames/s; at 512 - 1024 clips per batch four workers take the driver from 14.6 to 20.4 M frames/s).  A worker that dies, answers
anything but "ok" or does not answer within AMTX_NOTE_TIMEOUT seconds (default 30; replies are read with a deadline, never a bare
readline) gets the whole pool killed and turned off for the rest of the process, and the batch is ordered in-process -- results never
depend on the pool.  The shared file is unlinked as soon as every worker has mapped it: a crash of the parent leaves nothing in /dev/shm.
"""
import atexit
import json
import os
import select
import subprocess
import sys
import tempfile

import numpy as np

__all__ = ['reference_order', 'order_batch', 'pool_size']


def reference_order(onset_col):
    """Permutation of one clip's notes (np.nonzero order) into the reference's row order: three successive argsorts of the float64
    onset column with NumPy's default sort, composed as permutations."""
    p = onset_col.argsort()
    o = onset_col.take(p)
    for _ in range(2):
        q = o.argsort()
        p = p.take(q)
        o = o.take(q)
    return p


def _order_range(onset, offsets, perm, b0, b1):
    for b in range(b0, b1):
        lo, hi = int(offsets[b]), int(offsets[b + 1])
        if hi > lo:
            perm[lo:hi] = reference_order(onset[lo:hi]) + lo


def pool_size():
    env = os.environ.get('AMTX_NOTE_WORKERS')
    if env is not None:
        return max(0, int(env))
    return max(0, min(4, (os.cpu_count() or 1) // 4))


class _Pool(object):
    def __init__(self, n):
        self.n = n
        self.procs = []
        self.path = None
        self.capacity = 0
        self.failed = False
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''), OMP_NUM_THREADS='1')
        for _ in range(n):
            self.procs.append(subprocess.Popen([sys.executable, '-m', 'amt_tools_amd._order_pool'], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                               env=env, text=True, bufsize=1))
        atexit.register(self.close)

    def _reply(self, p):
        """One reply line of worker `p`, read with a deadline (a stopped / swapped-out worker must not hang the transcription driver)."""
        deadline = float(os.environ.get('AMTX_NOTE_TIMEOUT', '30'))
        ready, _, _ = select.select([p.stdout], [], [], deadline)
        if not ready:
            raise RuntimeError(f'note-order worker {p.pid} did not answer within {deadline:.0f} s')
        line = p.stdout.readline()
        if line.strip() != 'ok':
            raise RuntimeError(f'note-order worker answered {line!r}')

    def _buffer(self, E, B):
        need = 16 * E + 8 * (B + 1) + 64
        if self.path is None or need > self.capacity:
            self._drop_file()
            d = '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else tempfile.gettempdir()
            fd, self.path = tempfile.mkstemp(prefix=f'amtx_notes_{os.getpid()}_', dir=d)
            try:
                self.capacity = max(need, 1 << 22)
                os.ftruncate(fd, self.capacity)
                self._mm = np.memmap(self.path, dtype=np.uint8, mode='r+', shape=(self.capacity,))
                # every worker maps the new file now (an empty task), then the name goes away: the mappings keep the pages, and nothing
                # is left behind in /dev/shm if this process dies
                for p in self.procs:
                    p.stdin.write(json.dumps({'path': self.path, 'size': self.capacity, 'E': 0, 'B': 0, 'b0': 0, 'b1': 0}) + '\n')
                    p.stdin.flush()
                for p in self.procs:
                    self._reply(p)
            finally:
                os.close(fd)
                self._unlink()
        return np.asarray(self._mm)                  # plain ndarray view: slicing an np.memmap builds a subclass instance per slice (slow)

    def order(self, onset, offsets, B):
        E = int(offsets[B])
        mm = self._buffer(E, B)
        on = mm[:8 * E].view(np.float64)
        pm = mm[8 * E:16 * E].view(np.int64)
        of = mm[16 * E:16 * E + 8 * (B + 1)].view(np.int64)
        on[:] = onset[:E]
        of[:] = offsets[:B + 1]
        # clips dealt by note count: cut points at equal shares of E
        cuts = [0]
        for w in range(1, self.n):
            cuts.append(int(np.searchsorted(of, E * w // self.n, side='left')))
        cuts.append(B)
        cuts = [min(max(c, 0), B) for c in cuts]
        active = []
        for w, p in enumerate(self.procs):
            b0, b1 = cuts[w], max(cuts[w], cuts[w + 1])
            if b1 > b0:
                p.stdin.write(json.dumps({'path': self.path, 'size': self.capacity, 'E': E, 'B': B, 'b0': b0, 'b1': b1}) + '\n')
                p.stdin.flush()
                active.append(p)
        for p in active:
            self._reply(p)
        return np.array(pm, copy=True)

    def _unlink(self):
        if self.path is not None and os.path.exists(self.path):
            try:
                os.unlink(self.path)
            except OSError:
                pass

    def _drop_file(self):
        self._unlink()
        self.path = None
        self._mm = None

    def close(self, kill=False):
        for p in self.procs:
            try:
                p.stdin.close()
            except Exception:       # noqa: BLE001
                pass
        for p in self.procs:
            try:
                if kill:
                    p.kill()
                p.wait(timeout=2)
            except Exception:       # noqa: BLE001
                p.kill()
        self.procs = []
        self._drop_file()


_POOL = None


def order_batch(rows, onset, offsets, B, min_clips=384):
    """rows (E,3) float64 in np.nonzero order per clip, onset (E,) its first column (contiguous), offsets (B+1,) -> list of B (K,3) arrays
    in the reference's row order.  Batches of at least `min_clips` clips go through the worker pool when there is one."""
    global _POOL
    offsets = np.asarray(offsets, dtype=np.int64)
    E = int(offsets[B])
    perm = None
    n = pool_size()
    if n > 0 and B >= min_clips and E > 0 and not (_POOL is not None and _POOL.failed):
        try:
            if _POOL is None or _POOL.n != n or not _POOL.procs:
                if _POOL is not None:
                    _POOL.close()   # AMTX_NOTE_WORKERS changed: the old workers and their mapping go first
                _POOL = _Pool(n)
            perm = _POOL.order(onset, offsets, B)
        except Exception:           # noqa: BLE001  (a dead / wedged worker, a full /dev/shm ...): order in-process, and stay there
            if _POOL is not None:
                _POOL.failed = True
                _POOL.close(kill=True)
            perm = None
    if perm is None:
        perm = np.empty(E, dtype=np.int64)
        _order_range(onset, offsets, perm, 0, B)
    ordered = rows[:E].take(perm, axis=0) if E else rows[:0]
    off = offsets.tolist()
    return [ordered[off[b]:off[b + 1]] if off[b + 1] > off[b] else np.empty([0, 3]) for b in range(B)]


def _worker_main():
    maps = {}
    for line in sys.stdin:
        try:
            t = json.loads(line)
            key = t['path']
            mm = maps.get(key)
            if mm is None:          # a new file: map it now (the parent unlinks the name as soon as every worker has answered)
                maps.clear()
                maps[key + '#mm'] = np.memmap(key, dtype=np.uint8, mode='r+', shape=(int(t['size']),))
                mm = maps[key] = np.asarray(maps[key + '#mm'])      # plain ndarray view of the mapping
            E, B = t['E'], t['B']
            on = mm[:8 * E].view(np.float64)
            pm = mm[8 * E:16 * E].view(np.int64)
            of = mm[16 * E:16 * E + 8 * (B + 1)].view(np.int64)
            _order_range(on, of, pm, t['b0'], t['b1'])
            sys.stdout.write('ok\n')
        except Exception as e:      # noqa: BLE001
            sys.stdout.write(f'error {e!r}\n')
        sys.stdout.flush()


if __name__ == '__main__':
    _worker_main()
