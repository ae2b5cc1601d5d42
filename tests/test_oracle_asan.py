"""CPU: the oracle's restatement under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan`).
The golden vectors, the hand-derived unit cases and a synthetic pile go through the instrumented library in a child process (the sanitizer
runtime has to be the first library of the process: LD_PRELOAD); any report ends the child with a non-zero status."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.timeout(900)
def test_golden_set_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("no sanitizer runtime beside gcc")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan", f"OUT={tmp_path}"])
    env = dict(os.environ, LD_PRELOAD=f"{asan}:{ubsan}", CW_ORACLE_LIB=str(tmp_path / "liboracle_asan.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_golden.py"),
                          os.path.join(ROOT, "tests", "test_oracle_units.py"), "-k", "not baseline and not simd"],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=850)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]
