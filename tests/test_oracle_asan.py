"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: GPU
sanitizers are not available on the pool). oracle/fuzz_main.c is self-checking."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_fuzz_under_asan_ubsan():
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", odir, "fuzz_asan"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(odir, "fuzz_asan")], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "oracle fuzz clean" in out.stdout
