"""
tools/host_sanitizer_check.py — CPU-only: the host side of libpi_mi355 under AddressSanitizer + UBSan.

    python tools/host_sanitizer_check.py          (build container; no GPU needed or used)

Sanitizer builds are a CPU-build matter: GPU sanitizers are not available on the GPU pool, which refuses to run
trees whose test files carry sanitizer flags, so this check is a tool (listed in .gpurunignore, never part of a
`pytest` run, never on a GPU box) and its outcome is recorded in profiles/r03/host_sanitizers.txt.
"""
from __future__ import annotations

import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def host_runtime_is_clean_under_address_and_ub_sanitizers(tmp_path: Path) -> str:
    """The host side of the library (handles, source generation, hipRTC plumbing, exchange planner, argument
    validation, inference handle) compiled FROM ITS SOURCES with -fsanitize=address,undefined into
    tools/native/host_asan_driver.cpp and run without a GPU: no report, exit code 0.  (The device side has
    the checked build instead: GPU sanitizers do not exist on this platform.)"""
    import os
    import subprocess
    import __graft_entry__ as G
    csrc = ROOT / "dynamicprogramming_amd" / "csrc"
    exe = tmp_path / "host_asan"
    cmd = [G.HIPCC, "-x", "c++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           "-ffp-contract=off", "-std=c++17", "-Wall", "-Wextra", f"-I{ROOT / 'include'}", "-I/opt/rocm/include",
           "-D__HIP_PLATFORM_AMD__", f'-DPI_CSRC_DIR="{csrc}"', f'-DPI_INCLUDE_DIR="{ROOT / "include"}"',
           str(csrc / "pi_api.cpp"), str(csrc / "pi_comm.cpp"), str(csrc / "pi_infer.cpp"), str(csrc / "pi_p2p.cpp"),
           str(ROOT / "tools" / "native" / "host_asan_driver.cpp"), "-o", str(exe),
           "-L/opt/rocm/lib", "-lhiprtc", "-lamdhip64", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    run = subprocess.run([str(exe), str(tmp_path / "cache")], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0 and "host_asan_driver: ok" in run.stdout, run.stdout[-2000:] + run.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error:" not in run.stderr
    return run.stdout.strip()


if __name__ == "__main__":
    with tempfile.TemporaryDirectory(prefix="pi_asan_") as tmp:
        print(host_runtime_is_clean_under_address_and_ub_sanitizers(Path(tmp)))
    print("host_sanitizer_check: clean (AddressSanitizer incl. leak detection, UBSan)")
