"""
In-tree build of the HIP extension: every source under amt_tools_amd/csrc is compiled for gfx950 with
hipcc and linked into amt_tools_amd/csrc/libamtx.so (the C ABI of include/amtx.h).  hipcc cross-compiles
without a GPU, so this runs in the CPU-only build container; the .so travels to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(CSRC, 'libamtx.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function'] + os.environ.get('AMTX_EXTRA_FLAGS', '').split()


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp')))


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hdrs.append(os.path.join(ROOT, 'include', 'amtx.h'))
    return max(os.path.getmtime(h) for h in hdrs)


# Per-file flags.  spec.hip: without the SLP vectorizer hipcc emits scalar v_fma / v_add instead of v_pk_* pairs plus the v_mov
# shuffles that feed them (2641 -> 2555 vector instructions, no scratch): spec_power 2.25 -> 2.01 ms per 1024 clips.
# convg.hip: conv2 / conv3 at model_complexity 3 12.8 / 11.5 -> 11.4 / 10.0 ms per 512 clips.  The other files measure the same
# cqt_dec.hip: the decimator's producer waves split a raw tile into bf16 planes beside the matrix waves of the same SIMDs; packed f32 adds there
# (v_pk_add_f32) cost 2465 instead of 1422 cycles per tile.  The other files measure the same
# either way (conv.hip, gemm.hip, lstm.hip: within 1 %) and keep the default.
FILE_FLAGS = {'spec.hip': ['-fno-slp-vectorize'], 'convg.hip': ['-fno-slp-vectorize'], 'cqt_dec.hip': ['-fno-slp-vectorize']}


# OPTIONAL (AMTX_BUILD_F16=1): compiled a second time with -DAMTX_F16 (IEEE half operands instead of bf16, public functions suffixed _f16:
# csrc/amtx_f16_names.h) = the engine's precision 'f16'.  Off by default since round 6: it recompiles the six largest kernel files for a
# mode no BASELINE config names; without it amtx_has_f16() is 0 and amtx_of_model_create refuses AMTX_PREC_F16.
F16_TWINS = ('conv.hip', 'convf.hip', 'convg.hip', 'gemm.hip', 'lstm.hip', 'pack.hip')
WITH_F16 = os.environ.get('AMTX_BUILD_F16', '0') not in ('', '0')
if WITH_F16:
    FLAGS.append('-DAMTX_WITH_F16')
STAMP = os.path.join(CSRC, '.build_flags')


def _compile(src, hdr_mtime, verbose, f16=False):
    obj = os.path.join(CSRC, os.path.splitext(src)[0] + ('_f16.o' if f16 else '.o'))
    path = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_mtime):
        return obj
    cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(src, []) + (['-DAMTX_F16'] if f16 else []) + (['-x', 'hip'] if src.endswith('.hip') else []) + ['-c', path, '-o', obj]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return obj


def build(verbose=True, jobs=4):
    """Compile + link; rebuilds only what changed.  Returns the path of libamtx.so."""
    hdr_mtime = _deps()
    # objects are only as fresh as the flags they were compiled with: a changed flag set (AMTX_BUILD_F16, AMTX_EXTRA_FLAGS) rebuilds all
    flags_now = ' '.join(FLAGS)
    try:
        with open(STAMP) as f:
            flags_then = f.read()
    except OSError:
        flags_then = None
    relink = flags_then != flags_now
    if relink:
        for f in os.listdir(CSRC):
            if f.endswith('.o'):
                os.remove(os.path.join(CSRC, f))
        with open(STAMP, 'w') as f:
            f.write(flags_now)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        jobs_ = [(s, False) for s in _sources()] + ([(s, True) for s in F16_TWINS] if WITH_F16 else [])
        objs = list(ex.map(lambda j: _compile(j[0], hdr_mtime, verbose, j[1]), jobs_))
    if relink or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == '__main__':
    build(verbose=True)
    print(LIB)
    sys.exit(0)
