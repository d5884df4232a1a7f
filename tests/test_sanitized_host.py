"""SURVEY section 5 / VERDICT r02 item 8: the HOST halves of the native library (weight packing, amtx_of_model_finalize, plan / basis
builders, C-ABI argument checks: hundreds of lines of index arithmetic) under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU
build only (the GPU pool refuses GPU ASan).  tests/san/build_san.py compiles them host-only with the sanitizers and links a host-memory
shim for the HIP runtime; tests/san/driver.py drives them in a child process that has clang's ASan runtime preloaded.  A report from
either sanitizer aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'san'))


def test_host_halves_of_the_library_are_clean_under_asan_and_ubsan():
    import build_san
    if not os.path.exists(build_san.HIPCC):
        pytest.skip('no hipcc: the sanitizer build needs the ROCm clang')
    lib = build_san.build()
    env = dict(os.environ, LD_PRELOAD=build_san.asan_runtime(), ASAN_OPTIONS='detect_leaks=0:abort_on_error=1:halt_on_error=1',
               UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'san', 'driver.py'), lib], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-6000:])
    assert 'no report' in p.stdout
    assert 'runtime error' not in p.stderr and 'AddressSanitizer' not in p.stderr, p.stderr[-6000:]
