"""
The workgroup -> chunk schedules of the sweeps (slab and strip) on the host: whatever the planner
(pi_plan_schedule) answers, the kernels' map — restated in tests/helpers.schedule_groups — must hand every group of the
launch to exactly one workgroup, on the XCD the schedule promises.  The same map runs on the GPU in
tests/test_gpu_parity.py (pi_probe_coords walks it).  No reference counterpart: the reference launches one thread per state
in index order (src/cuda_policy_iteration.py:305-318).
"""
from __future__ import annotations

import numpy as np
import pytest

from dynamicprogramming_amd import _native, envs
from tests import helpers as H


def _engine(name, shape, order=None):
    cls = envs.ENVS[name]
    bins = H.env_bins(name, shape)
    return _native.Engine(cls._D, [len(b) for b in bins], [b.min() for b in bins], [b.max() for b in bins], bins,
                          cls.ACTIONS, device=-1, order=order)


def _check_cover(sched, block, count):
    n_chunks = -(-count // block)
    groups = -(-n_chunks // sched["cpw"])
    g = H.schedule_groups(sched, n_chunks)
    taken = g[g[:, 0] >= 0, 0]
    assert len(taken) == groups and np.array_equal(np.sort(taken), np.arange(groups)), "a group is missed or taken twice"
    return g


def test_strips_on_the_headline_grid_are_a_plane_of_the_slowest_dimension_when_asked_for():
    eng = _engine("double_pendulum_swingup", (80, 80, 80, 80), order=(0, 2, 1, 3))
    eng.set_option(7, -1)
    assert eng.info(35) == 0                           # the library's choice for 4-D grids: slab (measured slower with strips)
    eng.set_option(7, 80 ** 3)
    assert eng.info(35) == 80 ** 3
    s = eng.plan_schedule(1024, 0, 80 ** 4, chunks_per_workgroup=2)
    assert (s["period"], s["phase"]) == (250, 0) and (s["grid_x"], s["grid_y"]) == (32 * 8, 80)
    g = _check_cover(s, 1024, 80 ** 4)
    # every XCD takes its eighth of every plane (the boundaries move by at most one group) and the same work overall
    plane, pos = g[:, 0] // 250, g[:, 0] % 250
    for x in range(8):
        mine = (g[:, 1] == x) & (g[:, 0] >= 0)
        assert set(plane[mine]) == set(range(80))
        assert pos[mine].min() >= x * 250 // 8 and pos[mine].max() <= ((x + 1) * 250 + 7) // 8
        assert abs(int(mine.sum()) - 80 * 250 // 8) <= 1
    eng.set_option(7, 0)
    s0 = eng.plan_schedule(1024, 0, 80 ** 4, chunks_per_workgroup=2)
    assert s0["period"] == 0 and (s0["grid_x"], s0["grid_y"]) == (20000, 1)
    _check_cover(s0, 1024, 80 ** 4)
    eng.close()


def test_the_librarys_choice_on_big_6d_grids_is_a_few_sub_planes():
    eng = _engine("double_cartpole_swingup", (25,) * 6, order=(4, 5, 2, 3, 0, 1))
    eng.set_option(7, -1)
    assert eng.info(35) == 5 * 25 ** 4                 # ~2^21 states, whole (i0, i1) sub-planes
    s = eng.plan_schedule(256, 0, 25 ** 6, chunks_per_workgroup=4)
    assert s["period"] == round(5 * 25 ** 4 / 1024) and s["grid_y"] == -(-(-(-25 ** 6 // 1024)) // s["period"])
    _check_cover(s, 256, 25 ** 6)
    # a live-state list of 23/25 of the grid (the cart leaves the track in two of 25 bins): the same share of the list
    live = 25 ** 6 // 25 * 23
    sl = eng.plan_schedule(256, 0, live, total=live, chunks_per_workgroup=4)
    assert sl["period"] == round(5 * 25 ** 4 * 23 / 25 / 1024)
    _check_cover(sl, 256, live)
    eng.close()


def test_small_and_2d_grids_keep_the_slab_schedule():
    for name, shape in (("pendulum", (200, 200)), ("cartpole_swingup", (50, 50, 50, 50)), ("double_cartpole", (7,) * 6),
                        ("double_pendulum_swingup", (80, 80, 80, 80))):
        eng = _engine(name, shape)
        eng.set_option(7, -1)
        assert eng.info(35) == 0, (name, shape)
        eng.close()


@pytest.mark.parametrize("seed", range(6))
def test_every_group_is_taken_exactly_once_whatever_the_period_and_the_range(seed):
    rng = np.random.default_rng(seed)
    eng = _engine("cartpole", (31, 23, 29, 19))
    n = eng.info(0)
    for _ in range(40):
        block = int(rng.choice([64, 256, 512, 1024]))
        cpw = int(rng.integers(1, 5))
        period_states = int(rng.integers(block * cpw * 16, n // 2))
        eng.set_option(7, period_states)
        first = int(rng.integers(0, n // 2))
        if rng.random() < 0.5:
            first -= first % (block * cpw)             # ranges that start on a group boundary and ranges that do not
        count = int(rng.integers(1, n - first + 1))
        total = n if rng.random() < 0.6 else int(rng.integers(n // 3, n))       # a list that stands for the grid
        s = eng.plan_schedule(block, first, count, total=total, chunks_per_workgroup=cpw)
        assert s["grid_x"] % 8 == 0 and 1 <= s["grid_y"] <= 65535
        _check_cover(s, block, count)
    eng.close()


def test_a_period_of_too_few_groups_and_a_range_inside_one_period_fall_back_to_the_slab():
    eng = _engine("cartpole", (31, 23, 29, 19))
    eng.set_option(7, 1024 * 8)                       # 8 groups of 1 024: under the 16-group minimum
    assert eng.plan_schedule(1024, 0, eng.info(0))["period"] == 0
    eng.set_option(7, 1024 * 64)
    assert eng.plan_schedule(1024, 0, eng.info(0))["period"] == 64
    assert eng.plan_schedule(1024, 0, 1024 * 40)["period"] == 0          # the whole range is less than one period
    with pytest.raises(RuntimeError):
        eng.set_option(7, eng.info(0) + 1)
    eng.close()
