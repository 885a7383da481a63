"""Sanitizer runs of the CPU-side native code (`make -C oracle san-run`; no GPU involved, GPU sanitizers are not available on this pool).

* ``csrc/host_util.cpp`` -- the host-only helpers behind the C-ABI's host-buffer entry points (float64 -> uint8 narrowing on host threads
  with its abort-on-first-bad-sample race, the pinned-ring copy loop of large downloads behind a memcpy shim of the transport: sizes that
  are not a multiple of the chunk, ring wrap-around, 1..4 slots, transport failures) -- under ASan + UBSan and under TSan
  (tests/native/test_host_util.cpp).
* ``oracle/slgc_oracle.c`` -- the C oracle under ASan + UBSan, driven through every golden vector (tests/test_oracle_golden.py,
  tests/test_oracle_cross.py run against the instrumented library)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

ORACLE = os.path.join(ROOT, "oracle")


def _have_sanitizers():
    if not shutil.which("gcc") or not shutil.which("g++") or not shutil.which("make"):
        return False
    for lib in ("libasan.so", "libubsan.so", "libtsan.so"):
        p = subprocess.run(["gcc", f"-print-file-name={lib}"], capture_output=True, text=True).stdout.strip()
        if not os.path.isabs(p):
            return False
    return True


pytestmark = pytest.mark.skipif(not _have_sanitizers(), reason="gcc sanitizer runtimes not installed")


@pytest.fixture(scope="module")
def built():
    r = subprocess.run(["make", "-s", "-C", ORACLE, "san"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return os.path.join(ORACLE, "_san")


def test_host_util_under_asan_ubsan(built):
    r = subprocess.run([os.path.join(built, "test_host_util_asan")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_host_util_under_tsan(built):
    r = subprocess.run([os.path.join(built, "test_host_util_tsan"), "light"], capture_output=True, text=True, timeout=600,      # reduced matrix: TSan is 10-20x slower
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_c_oracle_golden_vectors_under_asan_ubsan(built):
    def rt(name):
        return subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=rt("libasan.so") + ":" + rt("libubsan.so"), ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1",
               SLGC_ORACLE_SO=os.path.join(built, "libslgc_oracle_san.so"))
    r = subprocess.run(["python3", "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), os.path.join(ROOT, "tests", "test_oracle_cross.py"),
                        "-x", "-q", "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0 and " passed" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, \
        r.stdout[-1500:] + r.stderr[-3000:]
