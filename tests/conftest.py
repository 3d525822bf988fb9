import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # libpi_mi355.so is a git-ignored build product: build it when a fresh checkout is tested
    # before __graft_entry__.build() has run (hipcc cross-compiles without a GPU).
    lib = ROOT / "dynamicprogramming_amd" / "libpi_mi355.so"
    if not lib.exists():
        import subprocess
        res = subprocess.run(["make", "-C", str(ROOT / "dynamicprogramming_amd" / "csrc")],
                             capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("could not build libpi_mi355.so:\n" + res.stdout + res.stderr)


@pytest.fixture(scope="session")
def cuda_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but torch.cuda.is_available() is False")
    return torch.device("cuda:0")
