"""AddressSanitizer + UndefinedBehaviorSanitizer build of the CPU oracle (the
checker), as the reference's own build offers for its code (CMakeLists.txt:
107-109, cpp/open3d/CMakeLists.txt:60-67): `make -C oracle asan` compiles the
oracle sources with -fsanitize=address,undefined and oracle/asan_driver.cpp
walks the hot path once; any report is a failure."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_walk_is_clean_under_asan_and_ubsan():
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"),
                        "asan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    exe = os.path.join(ROOT, "oracle", "_san", "oracle_asan")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               OMP_NUM_THREADS="2")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                       env=env)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:],
                               r.stderr[-4000:])
    assert "oracle sanitizer walk ok" in r.stdout
    assert "runtime error" not in r.stderr
    assert "AddressSanitizer" not in r.stderr
