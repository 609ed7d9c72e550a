"""AddressSanitizer + UBSan runs of the CPU side ("sanitizers = ASan on the CPU oracle"; GPU ASan is not available on this pool).

Two sanitizer builds, both CPU-only and both loaded in CHILD processes (the sanitizer runtime has to be the first library of the process):
  * oracle/libzkoracle_asan.so  (make -C oracle asan): the checker itself -- its own known-answer tests and the multi-threaded context prover
  * zukelang_amd/libzkhost_asan.so (make -C zukelang_amd/csrc asan-host): the HOST half of the product library, i.e. the pairing / verify /
    decompress code of csrc/pairing_host.hip compiled as plain C++ -- the verify surface tests run against it.
A sanitizer report aborts the child (halt_on_error, -fno-sanitize-recover), so a green run means no report.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def _run_under_sanitizers(env_extra, pytest_args, timeout):
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan:
        pytest.skip("no libasan.so in this toolchain")
    env = dict(os.environ, LD_PRELOAD=":".join(x for x in (asan, ubsan) if x),
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", **env_extra)
    res = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + pytest_args, capture_output=True, text=True,
                         timeout=timeout, env=env, cwd=ROOT)
    tail = res.stdout[-3000:] + res.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    assert res.returncode == 0, tail
    assert " passed" in res.stdout, tail


@pytest.mark.slow
def test_cpu_oracle_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    so = os.path.join(ROOT, "oracle", "libzkoracle_asan.so")
    # the oracle's own pins (field / group / polynomial KATs, literal == MSM form == trapdoor for Groth16) and the pthreads context prover
    _run_under_sanitizers({"ZK_ORACLE_SO": so},
                          ["tests/test_oracle.py", "tests/test_fast_cpu.py", "-m", "not gpu", "-k", "not pinocchio_literal"], timeout=1500)


@pytest.mark.slow
def test_host_half_of_the_product_library_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "zukelang_amd", "csrc"), "asan-host"])
    so = os.path.join(ROOT, "zukelang_amd", "libzkhost_asan.so")
    # the verify surface (zk_pairing_*, zk_groth16_verify, zk_pinocchio_verify: csrc/pairing_host.hip) against the oracle, under the sanitizers
    _run_under_sanitizers({"ZK_LIBZKMI355X_PATH": so}, ["tests/test_pairing_host.py", "-m", "not gpu"], timeout=1500)
