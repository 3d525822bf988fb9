"""
The 4-D / 6-D env plugins against reference-EXECUTED code (CPU half; the GPU half is
tests/test_gpu_parity.py::test_dynamics_match_reference_step_python).

tests/golden/step_python.npz holds what the reference's own `_step_python` functions
(/root/reference/runners/cartpole_swingup_cuda.py:138-169, double_pendulum_swingup_cuda.py:211-268,
double_cartpole_cuda.py:186-235, double_cartpole_swingup_cuda.py:250-325, overhead_crane_cuda.py:211-245)
returned for 2 000 seeded (state, action) pairs per env — the only reference-executed statement about the
dynamics of the headline config (double pendulum) and of every other 4-D / 6-D env.  Here: this
repository's plugin strings, compiled into the CPU checker in both arithmetic modes, against those vectors.
Tolerances (float32 kernel arithmetic vs a float64 mirror) are in tests/helpers.py.
"""
import numpy as np
import pytest

from tests import helpers as H


def test_fixture_is_complete():
    g = np.load(H.GOLDEN / "step_python.npz")
    for name, wraps in H.STEP_PYTHON_ENVS.items():
        st = g[f"{name}_states"]
        assert st.dtype == np.float32 and st.shape[0] >= 2000
        assert g[f"{name}_next"].shape == st.shape and g[f"{name}_next"].dtype == np.float32
        assert g[f"{name}_reward"].dtype == np.float64 and g[f"{name}_term"].dtype == bool
        # the sample reaches what it was built to reach: wraps, both outcomes of the termination test
        for d in wraps:
            jumped = np.abs(g[f"{name}_next"][:, d] - st[:, d]) > np.pi
            assert jumped.sum() >= 50, (name, d, int(jumped.sum()))
        if name != "double_pendulum_swingup":
            assert 100 <= g[f"{name}_term"].sum() <= len(st) - 100


@pytest.mark.parametrize("libm", [True, False], ids=["glibc", "pi_math"])
@pytest.mark.parametrize("name", list(H.STEP_PYTHON_ENVS))
def test_plugin_strings_match_reference_step_python(name, libm):
    g = np.load(H.GOLDEN / "step_python.npz")
    nxt, rew, done = H.oracle_for(name, libm=libm).step(g[f"{name}_states"], g[f"{name}_actions"])
    got = H.check_against_step_python(name, nxt, rew, done, "checker (glibc)" if libm else "checker (pi_math)")
    assert got["flags_compared"] >= got["pairs"] - 10
