"""tools/gather_lines.py — CPU analysis (no GPU): how many distinct 128-byte lines of V does one
wave-wide corner load of the evaluation sweep touch, for a given lane->state mapping?

For sampled waves (64 lanes) of the BASELINE C4 grid it runs the env dynamics through the test
oracle's `step` + `interp` (analysis only, never product code), takes each lane's cell base index
and counts, per corner-pair load (8 per state in 4-D: the two corners along the last dimension are
one 8-byte load), the distinct 128-B lines over the wave.  The L1 (TCP) looks up one line per
cycle, so lines/load x loads is the gather's floor in TCP cycles per wave.
"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import oracle
from dynamicprogramming_amd import envs

env, bins = "double_pendulum_swingup", 80
if len(sys.argv) > 2:
    env, bins = sys.argv[1], int(sys.argv[2])
cls = envs.ENVS[env]
D = cls._D
tables = [np.asarray(b, np.float32) for b in cls.bins_space(bins).values()]
lo, hi, shape, strides = oracle.grid_metadata(tables)
chk = oracle.build(D, envs.dynamics_source(env))
n = int(np.prod(shape))
rng = np.random.default_rng(0)
LINE = 32   # floats per 128-B line


def coords_of(flat):
    idx = np.stack(np.unravel_index(flat, tuple(shape)), axis=1)
    return np.stack([tables[d][idx[:, d]] for d in range(D)], axis=1).astype(np.float32), idx


def fit_shear():
    """Least-squares slope of (successor cell - own index) in every dimension against the index of
    the last (lane) dimension, over random states and actions: theta' = theta + dt * omega makes the
    successor's theta_2 row drift linearly along a wave's omega_2 lanes."""
    m = 100000
    flat = rng.choice(n, size=m, replace=False)
    st, idx = coords_of(flat)
    ns, _, _ = chk.step(st, rng.choice(cls.ACTIONS, size=m).astype(np.float32))
    beta = np.zeros(D)
    for d in range(D - 1):
        dc = (ns[:, d] - lo[d]) / (hi[d] - lo[d]) * (shape[d] - 1) - idx[:, d]
        ok = np.abs(dc) < shape[d] / 3                       # drop wrapped angles
        A = np.stack([idx[ok, D - 1], np.ones(ok.sum())], axis=1)
        beta[d] = np.linalg.lstsq(A, dc[ok], rcond=None)[0][0]
    return beta


_shear = None
quad_cost = []


def wave_states(mapping, w):
    """flat indices of the 64 lanes of sampled wave number w under a mapping."""
    if mapping == "flat":                       # lanes = consecutive flat indices (current kernel)
        return w * 64 + np.arange(64)
    if mapping == "shear":                      # lanes follow the drift: i_d = (r_d - round(beta_d * i_last)) mod g_d
        global _shear
        if _shear is None:
            beta = fit_shear()
            print("shear slopes (cells per lane index):", np.round(beta, 4), flush=True)
            _shear = [np.rint(beta[d] * (np.arange(shape[-1]) - shape[-1] // 2)).astype(int) for d in range(D)]
        pi = np.stack(np.unravel_index((w * 64 + np.arange(64)) % n, tuple(shape)), axis=1)
        for d in range(D - 1):
            pi[:, d] = (pi[:, d] - _shear[d][pi[:, D - 1]]) % shape[d]
        return np.ravel_multi_index(tuple(pi.T), tuple(shape))
    if mapping.startswith("tile"):              # lanes = t2 x t3 tile of the last two dimensions
        t2, t3 = map(int, mapping[4:].split("x"))
        g2, g3 = shape[-2], shape[-1]
        tiles3, tiles2 = -(-g3 // t3), -(-g2 // t2)
        outer, r = divmod(w, tiles2 * tiles3)
        a, b = divmod(r, tiles3)
        i2 = np.minimum(a * t2 + np.arange(64) // t3, g2 - 1)
        i3 = np.minimum(b * t3 + np.arange(64) % t3, g3 - 1)
        return (outer * g2 * g3 + i2 * g3 + i3) % n
    raise ValueError(mapping)


def analyse(mapping, action_mode, n_waves=4000):
    total_waves = n // 64
    ws = rng.choice(total_waves, size=n_waves, replace=False)
    lines_per_load, cells = [], []
    global quad_cost
    quad_cost = []
    for w in ws:
        flat = wave_states(mapping, int(w))
        st, _ = coords_of(flat)
        if action_mode == "bang":               # bang-bang like the converged policy: sign by lane-coherent rule
            act = np.full(64, cls.ACTIONS[-1] if (w & 1) else cls.ACTIONS[0], np.float32)
        elif action_mode == "random":
            act = rng.choice(cls.ACTIONS, size=64).astype(np.float32)
        else:
            act = np.full(64, float(action_mode), np.float32)
        ns, rew, done = chk.step(st, act)
        idxs, wg = chk.interp(ns, lo, hi, shape, strides)
        base = idxs.min(axis=1)
        per_load = []
        C = 1 << D
        offs = sorted({int(o) for o in (idxs[0] - base[0])})
        pair_offs = [o for o in offs if (o % 2 == 0 or True)]
        # corner-pair loads: offsets whose last-dimension bit is 0
        pair = [o for o in offs if o + 1 in offs]
        for o in pair:
            a0 = (base + o) // LINE
            a1 = (base + o + 1) // LINE
            per_load.append(len(set(a0.tolist()) | set(a1.tolist())))
            # what the TCP charges (profiles/r03/negative_results.txt (16)): per 4-lane quad, its distinct lines
            quad_cost.append(sum(len(set(a0[q:q + 4].tolist()) | set(a1[q:q + 4].tolist())) for q in range(0, 64, 4)))
        lines_per_load.append(np.mean(per_load))
        cells.append(len(set(base.tolist())))
    lp = np.array(lines_per_load)
    return lp.mean(), np.percentile(lp, [10, 50, 90]), np.mean(cells)


for mapping in ["flat", "shear", "tile8x8", "tile4x16", "tile2x32", "tile16x4"]:
    for mode in ["bang", "random", "0.0"]:
        m, pct, cells = analyse(mapping, mode, 1500)
        print(f"{env}@{bins} mapping={mapping:9s} action={mode:6s}: lines per corner-pair load mean {m:5.1f} "
              f"(p10/p50/p90 {pct[0]:.0f}/{pct[1]:.0f}/{pct[2]:.0f}), distinct cells per wave {cells:.1f}, "
              f"TCP look-ups per load (sum over quads of their lines) {np.mean(quad_cost):5.1f}", flush=True)
