"""
tests/golden/make_barycentric_golden.py — golden vectors for the inference helper, produced by the
REFERENCE'S OWN function (/root/reference/utils/barycentric.py:15-77, :80-112) run in this build
container.  The reference cannot travel, so only the outputs are committed
(tests/golden/barycentric_utils.npz); tests/test_inference_utils.py pins utils/barycentric.py to them.

The reference's function is a numba @njit kernel; numba is not installed here, so the module text is
executed with `njit` replaced by the identity decorator (nothing is copied into the repo; the text
is read, compiled and run in memory).  One typing difference matters for the last bit of the
weights: numba types the literal in `w = 1.0` / `1.0 - t[d]` as float64, so the corner weights are
float64 products rounded to float32 once, while numpy >= 2 treats a Python float as a weak scalar
and would keep those products in float32.  The generator therefore runs the function twice:
  * `numba` typing  — float literals wrapped in np.float64 (an AST transform of the in-memory copy),
                      i.e. what the deployed reference computes;        -> weights_numba
  * plain execution — the function body as numpy 2 evaluates it;       -> weights_plain
Indices are identical in both.
"""
from __future__ import annotations

import ast
import sys
from itertools import product
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests.helpers import env_bins, sample_states  # noqa: E402
import oracle  # noqa: E402

REF = Path("/root/reference/utils/barycentric.py")
OUT = Path(__file__).resolve().parent / "barycentric_utils.npz"
CASES = {2: ("pendulum", (11, 7)), 4: ("cartpole", (5, 4, 6, 3)), 6: ("double_cartpole", (3, 4, 2, 5, 3, 2))}


class _Float64Literals(ast.NodeTransformer):
    def visit_Constant(self, node):
        if isinstance(node.value, float):
            return ast.copy_location(
                ast.Call(func=ast.Attribute(value=ast.Name(id="np", ctx=ast.Load()), attr="float64", ctx=ast.Load()),
                         args=[node], keywords=[]), node)
        return node


def load_reference(numba_typing: bool):
    tree = ast.parse(REF.read_text())
    body = []
    for node in tree.body:          # drop `from numba import njit`; an identity decorator stands in
        if isinstance(node, ast.ImportFrom) and node.module == "numba":
            continue
        body.append(node)
    tree.body = body
    if numba_typing:
        tree = _Float64Literals().visit(tree)
    ast.fix_missing_locations(tree)
    ns = {"njit": lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))}
    exec(compile(tree, str(REF), "exec"), ns)
    return ns["get_barycentric_weights_and_indices"], ns["get_optimal_action"]


def main() -> None:
    if not REF.exists():
        raise SystemExit("/root/reference is not present: fixtures can only be generated in the build container")
    ref_numba, act_numba = load_reference(True)
    ref_plain, _ = load_reference(False)
    out = {}
    for D, (name, shape) in CASES.items():
        bins = env_bins(name, shape)
        lo, hi, gshape, strides = oracle.grid_metadata(bins)
        bits = np.array(list(product([0, 1], repeat=D)), dtype=np.int32)
        rng = np.random.default_rng(100 + D)
        pts = sample_states(rng, bins, 2000)
        w_n, idx_n = ref_numba(pts, lo, hi, gshape, strides, bits)
        w_p, idx_p = ref_plain(pts, lo, hi, gshape, strides, bits)
        assert np.array_equal(idx_n, idx_p)
        policy = rng.integers(0, 5, size=int(np.prod(shape))).astype(np.int32)
        actions = np.linspace(-2.0, 2.0, 5).astype(np.float32)
        acts = np.array([act_numba(pts[k], policy, actions, lo, hi, gshape, strides, bits) for k in range(300)],
                        dtype=np.float64)
        out.update({f"d{D}_env": np.array(name), f"d{D}_shape": np.asarray(shape, np.int32),
                    f"d{D}_points": pts, f"d{D}_indices": idx_n, f"d{D}_weights_numba": w_n,
                    f"d{D}_weights_plain": w_p, f"d{D}_policy": policy, f"d{D}_actions": actions,
                    f"d{D}_optimal_action": acts})
        print(f"D={D}: {len(pts)} points, weights differ between typings at "
              f"{int((w_n.view(np.uint32) != w_p.view(np.uint32)).sum())} of {w_n.size} entries")
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
