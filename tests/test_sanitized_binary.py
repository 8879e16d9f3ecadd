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
