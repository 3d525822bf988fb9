"""
oracle/build_ref.py — build the reference's OWN kernel text as a CPU checker.

TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists (this container, never
the GPU box).  It reads the reference's Python files AS TEXT (``ast`` — nothing from the
reference is imported or executed), pulls out

  * the generic-kernel string of each solver class
    (src/cuda_policy_iteration.py: 2D :180-286, 4D :577-694, 6D :1004-1126), and
  * the ``_dynamics_cuda_src`` string of each runner's env class,

writes both to a temporary directory and compiles oracle/ref_driver.cpp around them into
``$TMPDIR/pi_mi355_ref-<uid>/libref_<env>_<hash of the extracted text>.so`` (a per-user directory,
created with mode 0700 and refused unless it belongs to this user with that mode; ``PI_MI355_REF_DIR``
overrides) — OUTSIDE the repository tree, so that neither reference
text nor anything compiled from it can travel to the GPU box with a snapshot of the repo
(tests/test_hygiene.py checks the tree).
``load(env)`` returns an ``oracle.OracleLib`` over that shared object, which
tests/golden/make_golden.py uses to (a) pin oracle/pi_oracle.cpp + this repo's env strings
bit-for-bit against the reference's text and (b) emit the golden vectors.
"""
from __future__ import annotations

import ast
import hashlib
import os
import subprocess
import tempfile
from pathlib import Path

from . import OracleLib, CXX

REFERENCE = Path("/root/reference")
_HERE = Path(__file__).resolve().parent
REF_DIR = Path(os.environ.get("PI_MI355_REF_DIR",
                              str(Path(tempfile.gettempdir()) / f"pi_mi355_ref-{os.getuid()}")))

# env name -> (runner file, class name, D)
RUNNERS = {
    "pendulum": ("runners/pendulum_cuda.py", "PendulumCuda", 2),
    "mountain_car": ("runners/mountain_car_cuda.py", "MountainCarCuda", 2),
    "continuous_mountain_car": ("runners/continuous_mountain_car_cuda.py",
                                "ContinuousMountainCarCuda", 2),
    "cartpole": ("runners/cartpole_cuda.py", "CartPoleCuda", 4),
    "cartpole_swingup": ("runners/cartpole_swingup_cuda.py", "CartPoleSwingUpCuda", 4),
    "double_pendulum_swingup": ("runners/double_pendulum_swingup_cuda.py",
                                "DoublePendulumSwingUpCuda", 4),
    "overhead_crane": ("runners/overhead_crane_cuda.py", "OverheadCraneCuda", 4),
    "double_cartpole": ("runners/double_cartpole_cuda.py", "DoubleCartPoleCuda", 6),
    "double_cartpole_swingup": ("runners/double_cartpole_swingup_cuda.py",
                                "DoubleCartPoleSwingUpCuda", 6),
}
_SOLVER_CLASS = {2: "CudaPolicyIteration2D", 4: "CudaPolicyIteration4D", 6: "CudaPolicyIteration6D"}
REF_CXXFLAGS = ["-O2", "-mfma", "-msse4.1", "-ffp-contract=off", "-fno-fast-math", "-shared",
                "-fPIC", "-std=c++17"]


def available() -> bool:
    return (REFERENCE / "src" / "cuda_policy_iteration.py").exists()


def _method_strings(path: Path, cls_name: str, method: str) -> list[str]:
    tree = ast.parse(path.read_text())
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == cls_name:
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name == method:
                    return [c.value for c in ast.walk(item)
                            if isinstance(c, ast.Constant) and isinstance(c.value, str)
                            and "__device__" in c.value]
    raise LookupError(f"{cls_name}.{method} not found in {path}")


def generic_kernel_text(D: int) -> str:
    (text,) = _method_strings(REFERENCE / "src" / "cuda_policy_iteration.py", _SOLVER_CLASS[D],
                              "_compile_cuda_module")
    return text


def dynamics_text(env: str) -> str:
    rel, cls, _ = RUNNERS[env]
    (text,) = _method_strings(REFERENCE / rel, cls, "_dynamics_cuda_src")
    if env == "overhead_crane":   # the runner bakes target_x = 0.0 in by string replace (:157)
        text = text.replace("__OC_X_TARGET__", f"{0.0:.6f}")
    return text


def _private_dir() -> Path:
    """REF_DIR, created 0700 and owned by this user (a shared, predictable directory would let
    another local user pre-plant a library that load() then dlopens)."""
    REF_DIR.mkdir(parents=True, exist_ok=True, mode=0o700)
    st = REF_DIR.stat()
    if st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError(f"{REF_DIR} must belong to uid {os.getuid()} with mode 0700 "
                           f"(found uid {st.st_uid}, mode {st.st_mode & 0o777:o})")
    return REF_DIR


def load(env: str, rebuild: bool = False) -> OracleLib:
    _, _, D = RUNNERS[env]
    if not available():
        raise RuntimeError("/root/reference is not present: cannot build the reference checker")
    dyn_text, gen_text = dynamics_text(env), generic_kernel_text(D)
    # the file name carries a hash of everything the library is built from, so a change of the
    # reference text (or of the driver / flags) can never be served from a stale build
    key = hashlib.sha256("\0".join([dyn_text, gen_text, (_HERE / "ref_driver.cpp").read_text(),
                                    " ".join(REF_CXXFLAGS)]).encode()).hexdigest()[:16]
    so = _private_dir() / f"libref_{env}_{key}.so"
    if rebuild or not so.exists():
        with tempfile.TemporaryDirectory(prefix="pi_ref_") as tmp:
            dyn = Path(tmp) / "dyn.inc"
            gen = Path(tmp) / "generic.inc"
            dyn.write_text(dyn_text)
            gen.write_text(gen_text)
            cmd = [CXX, *REF_CXXFLAGS, f"-DPI_D={D}", f'-DREF_DYN_FILE="{dyn}"',
                   f'-DREF_GENERIC_FILE="{gen}"', str(_HERE / "ref_driver.cpp"), "-o", str(so)]
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError(f"reference build failed:\n{' '.join(cmd)}\n{res.stderr}")
    return OracleLib(so, D)


def build_all(rebuild: bool = False) -> dict:
    return {env: load(env, rebuild) for env in RUNNERS}


if __name__ == "__main__":
    for name, lib in build_all(rebuild=True).items():
        print(f"{name:28s} D={lib.D}  {lib.path}")
