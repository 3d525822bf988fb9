"""include/pi_math.h on the CPU: accuracy of the deterministic sinf/cosf against float64 libm
and exactness of fmodf against glibc.  (That the GPU produces the same bits is checked by the
-m gpu dynamics tests, which run every env's trig through hipRTC-compiled code.)"""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]

SRC = r'''
#include "pi_math.h"
void t_sin(long n, const float* x, float* y) { for (long i = 0; i < n; ++i) y[i] = pi_sinf(x[i]); }
void t_cos(long n, const float* x, float* y) { for (long i = 0; i < n; ++i) y[i] = pi_cosf(x[i]); }
void t_fmod(long n, const float* x, const float* y, float* z) { for (long i = 0; i < n; ++i) z[i] = pi_fmodf(x[i], y[i]); }
void t_fmod_libm(long n, const float* x, const float* y, float* z) { for (long i = 0; i < n; ++i) z[i] = fmodf(x[i], y[i]); }
'''


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("pimath")
    (d / "t.c").write_text(SRC)
    subprocess.run(["gcc", "-O2", "-mfma", "-msse4.1", "-ffp-contract=off", "-shared", "-fPIC",
                    f"-I{ROOT / 'include'}", str(d / "t.c"), "-o", str(d / "t.so"), "-lm"], check=True)
    return ctypes.CDLL(str(d / "t.so"))


def _call(fn, *arrs):
    out = np.empty_like(arrs[0])
    fp = ctypes.POINTER(ctypes.c_float)
    fn(ctypes.c_long(len(out)), *[a.ctypes.data_as(fp) for a in arrs], out.ctypes.data_as(fp))
    return out


def _ulp_err(got, want64):
    w = want64.astype(np.float32)
    ulp = np.spacing(np.abs(w)).astype(np.float64)
    return np.abs(got.astype(np.float64) - want64) / np.maximum(ulp, 1e-45)


@pytest.mark.parametrize("scale", [3.2, 100.0, 1.0e5, 1.0e9])
def test_sin_cos_accuracy(lib, scale):
    rng = np.random.default_rng(int(scale))
    x = (rng.uniform(-1, 1, 2_000_000) * scale).astype(np.float32)
    k = np.arange(-50_000, 50_000) * (np.pi / 2)
    x = np.concatenate([x, k.astype(np.float32), np.nextafter(k.astype(np.float32), np.float32(np.inf))])
    x = x[np.abs(x) <= scale * 1.0001]
    assert np.max(_ulp_err(_call(lib.t_sin, x), np.sin(x.astype(np.float64)))) <= 2.0
    assert np.max(_ulp_err(_call(lib.t_cos, x), np.cos(x.astype(np.float64)))) <= 2.0


def test_sin_cos_special_values(lib):
    x = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-30, 3.0e38], dtype=np.float32)
    s, c = _call(lib.t_sin, x), _call(lib.t_cos, x)
    assert s[0] == 0.0 and c[0] == 1.0 and s[1] == 0.0
    assert np.isnan(s[2:5]).all() and np.isnan(c[2:5]).all()
    assert s[5] == np.float32(1e-30) and c[5] == 1.0
    assert abs(s[6]) <= 1.0 and abs(c[6]) <= 1.0


def test_fmod_is_exact(lib):
    rng = np.random.default_rng(5)
    n = 3_000_000
    a = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    b = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    a2 = (rng.uniform(-1, 1, n) * 10).astype(np.float32)             # the angle-wrap pattern
    b2 = np.full(n, np.float32(2 * np.float32(np.pi)))
    b3 = (a2 * rng.uniform(0, 3, n)).astype(np.float32)
    edge = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 3.4e38, 1.17549435e-38],
                    dtype=np.float32)
    ea, eb = np.meshgrid(edge, edge)
    for x, y in ((a, b), (a2, b2), (a2, b3), (ea.ravel().copy(), eb.ravel().copy())):
        got, want = _call(lib.t_fmod, x, y), _call(lib.t_fmod_libm, x, y)
        both_nan = np.isnan(got) & np.isnan(want)
        assert np.array_equal(got.view(np.uint32)[~both_nan], want.view(np.uint32)[~both_nan])
