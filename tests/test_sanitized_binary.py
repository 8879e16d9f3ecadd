"""The host side of the SVDSS binary (svdss_main.cpp, call_host.cpp, smooth_host.cpp: option parsing, FASTA / BAM
reading and writing, the smoothing rules and the record rebuild, the host index builder behind `SVDSS index`) under
AddressSanitizer + UndefinedBehaviorSanitizer: the process-level CPU tests of this directory once more, against a
sanitized build of the binary.  Opt-in (SVDSS_RUN_SANITIZED_BINARY=1): the build and the slower runs take three minutes,
the default CPU suite stays at a few; the result of the last run is in profiles/."""
import os
import subprocess
import sys

import pytest

from tests.common import ROOT

SAN = os.path.join(ROOT, "tests", "_SVDSS_san")
CSRC = os.path.join(ROOT, "svdss_amd", "csrc")


@pytest.mark.skipif(os.environ.get("SVDSS_RUN_SANITIZED_BINARY") != "1", reason="opt-in: SVDSS_RUN_SANITIZED_BINARY=1")
def test_process_level_tests_against_the_sanitized_binary():
    srcs = [os.path.join(CSRC, f) for f in ("svdss_main.cpp", "call_host.cpp", "smooth_host.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if not os.path.exists(SAN) or any(os.path.getmtime(d) > os.path.getmtime(SAN) for d in deps):
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=undefined", "-o", SAN] + srcs +
                       ["-L" + os.path.join(ROOT, "svdss_amd"), "-lsvdss_hip", "-lz", "-ldl",
                        "-Wl,-rpath," + os.path.join(ROOT, "svdss_amd")], check=True)
    env = dict(os.environ, SVDSS_TEST_BIN=SAN, ASAN_OPTIONS="detect_leaks=0:exitcode=99",
               UBSAN_OPTIONS="halt_on_error=1:exitcode=98:print_stacktrace=1")
    env.pop("SVDSS_RUN_SANITIZED_BINARY")
    p = subprocess.run([sys.executable, "-m", "pytest", "tests/test_smooth.py", "tests/test_cli.py", "tests/test_clipped.py",
                        "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=3000)
    assert p.returncode == 0, p.stdout[-4000:] + p.stderr[-2000:]


def test_the_oracle_itself_under_sanitizers():
    """The checker must not owe an answer to undefined behaviour either: the oracle's own tests (brute-force pins of the
    search, the independent score re-derivation, POA invariants) against `make -C oracle san` -- the same C files built
    with -fsanitize=address,undefined -- in a process with libasan preloaded.  Seconds, so it runs in the default suite."""
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan beside this gcc")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "san"], check=True, capture_output=True)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:exitcode=99", OMP_NUM_THREADS="4",
               UBSAN_OPTIONS="halt_on_error=1:exitcode=98:print_stacktrace=1",
               SVDSS_ORACLE_LIB=os.path.join(ROOT, "oracle", "libsvdss_oracle_san.so"))
    p = subprocess.run([sys.executable, "-m", "pytest", "tests/test_oracle.py", "tests/test_oracle_call.py", "tests/test_oracle_poa.py",
                        "tests/test_bench_helpers.py",      # (the batch entry points bench.py's cpu_baseline leg times)
                        "-x", "-q", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-4000:] + p.stderr[-2000:]
