"""tools/ack_stale_profiles.py — state, in the tree, that the committed profiles are of an OLDER kernel version.

tests/test_hygiene.py fails when profiles/rNN/bench_c4.json carries another kernel hash than the device code in the tree,
unless profiles/STALE_EVIDENCE_OK names the current hash.  Run this after changing csrc/pi_sweep_kernels.hip or
include/pi_math.h while the profiles have not been regenerated yet (tools/refresh_profiles.sh, GPU box); delete the file
once they have.  `--clear` removes it.
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from dynamicprogramming_amd import _native

path = Path(__file__).resolve().parents[1] / "profiles" / "STALE_EVIDENCE_OK"
if "--clear" in sys.argv:
    path.unlink(missing_ok=True)
    print("removed", path)
else:
    path.write_text(f"{_native.kernel_source_hash()}  kernels changed; profiles/rNN not yet regenerated for this hash\n")
    print(path.read_text().strip())
