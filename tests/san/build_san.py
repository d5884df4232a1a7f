"""TEST INFRASTRUCTURE: AddressSanitizer + UBSan build of the HOST halves of amt_tools_amd/csrc (weight packers, amtx_of_model_finalize,
spectrogram / CQT plan builders, the C-ABI argument checks), CPU only: `hipcc --cuda-host-only -fsanitize=address,undefined`, linked
against tests/san/hip_host_shim.cpp instead of the HIP runtime.  Output: tests/san/_build/libamtx_san.so (git-ignored, never shipped).
SURVEY section 5 / VERDICT r02 item 8: sanitizers run on the CPU build only (the GPU pool refuses GPU ASan)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'amt_tools_amd', 'csrc')
OUT = os.path.join(HERE, '_build')
LIB = os.path.join(OUT, 'libamtx_san.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
SAN = ['-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-fno-omit-frame-pointer', '-g', '-O1']
F16_TWINS = ('conv.hip', 'convf.hip', 'convg.hip', 'gemm.hip', 'lstm.hip', 'pack.hip')


def asan_runtime():
    return subprocess.check_output([os.path.join(os.path.dirname(os.path.realpath(HIPCC)), '..', 'lib', 'llvm', 'bin', 'clang++'),
                                    '-print-file-name=libclang_rt.asan-x86_64.so'], text=True).strip()


def _newest_source():
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp', '.h'))]
    files += [os.path.join(ROOT, 'include', 'amtx.h'), os.path.join(HERE, 'hip_host_shim.cpp'), os.path.abspath(__file__)]
    return max(os.path.getmtime(f) for f in files)


def build(verbose=False):
    if os.path.exists(LIB) and os.path.getmtime(LIB) > _newest_source():
        return LIB
    os.makedirs(OUT, exist_ok=True)
    jobs = [(f, False) for f in sorted(os.listdir(CSRC)) if f.endswith(('.hip', '.cpp'))] + [(f, True) for f in F16_TWINS]

    def cc(job):
        src, f16 = job
        obj = os.path.join(OUT, os.path.splitext(src)[0] + ('_f16.o' if f16 else '.o'))
        cmd = [HIPCC, '--cuda-host-only', '-std=c++17', '-fPIC', '-Wno-unused-function', '-DAMTX_WITH_F16'] + SAN + (['-DAMTX_F16'] if f16 else []) + \
              ['-x', 'hip', '-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(cc, jobs))
    # the host-only objects reference one __hip_fatbin_<hash> symbol each (the device code they were not built with): define them empty
    und = subprocess.check_output(['nm', '-u'] + objs, text=True)
    fat = sorted({l.split()[-1] for l in und.splitlines() if '__hip_fatbin_' in l})
    stub = os.path.join(OUT, 'fatbin_stubs.c')
    with open(stub, 'w') as f:
        for name in fat:
            f.write(f'const char {name}[8] = {{0}};\n')
    shim = os.path.join(OUT, 'hip_host_shim.o')
    subprocess.run([HIPCC, '--cuda-host-only', '-std=c++17', '-fPIC'] + SAN + ['-x', 'hip', '-c', os.path.join(HERE, 'hip_host_shim.cpp'), '-o', shim], check=True)
    stubo = os.path.join(OUT, 'fatbin_stubs.o')
    subprocess.run(['gcc', '-fPIC', '-c', stub, '-o', stubo], check=True)
    # -Bsymbolic: the library's own hipMalloc / hipLaunchKernel ... (the shim) win over any HIP runtime already in the process
    cmd = [os.path.join(os.path.dirname(os.path.realpath(HIPCC)), '..', 'lib', 'llvm', 'bin', 'clang++'), '-shared', '-fPIC', '-Wl,-Bsymbolic'] + SAN + \
          ['-o', LIB] + objs + [shim, stubo]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == '__main__':
    print(build(verbose=True))
    sys.exit(0)
