"""
CPU-side checks of libpi_mi355.so: it loads, exports every symbol include/pi_mi355.h
declares, specialises the kernel template for every env through hipRTC (no GPU needed to
COMPILE for gfx950), and fails loudly instead of falling back.
"""
from __future__ import annotations

import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

from dynamicprogramming_amd import _native, envs
from tests import helpers as H

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    text = (ROOT / "include" / "pi_mi355.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pi_[a-z_0-9]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _native.lib()
    names = declared_symbols()
    assert {"pi_create", "pi_compile", "pi_eval_sweep", "pi_eval_sweeps", "pi_improve_sweep",
            "pi_destroy", "pi_last_error"} <= set(names)
    for name in names:
        assert hasattr(lib, name), f"{name} declared in pi_mi355.h but not exported"
        assert name in _native.SIGNATURES, f"{name} has no ctypes signature in _native.py"
    assert lib.pi_abi_version() == _native.ABI_VERSION


def test_no_torch_types_in_the_abi():
    text = (ROOT / "include" / "pi_mi355.h").read_text()
    assert "torch" not in text.lower().replace("torch-rocm tensors' data_ptr", "")
    assert "at::" not in text and "c10::" not in text


def _host_engine(name, shape):
    cls = envs.ENVS[name]
    bins = H.env_bins(name, shape)
    return _native.Engine(cls._D, [len(b) for b in bins], [b.min() for b in bins],
                          [b.max() for b in bins], bins, cls.ACTIONS, device=-1)


@pytest.mark.parametrize("name", H.ENV_NAMES)
def test_every_env_compiles_for_gfx950_without_a_gpu(name, tmp_path):
    shape = {2: (24, 17), 4: (9, 7, 11, 5), 6: (5, 4, 6, 4, 5, 4)}[envs.ENVS[name]._D]
    eng = _host_engine(name, shape)
    eng.compile(envs.dynamics_source(name), cache_dir=tmp_path)
    assert eng.info(7) == 0                       # compiled, not cached
    objs = list(tmp_path.glob("pi_*.hsaco"))
    assert len(objs) == 1 and objs[0].stat().st_size > 4096
    assert objs[0].read_bytes()[:4] == b"\x7fELF"
    eng2 = _host_engine(name, shape)
    eng2.compile(envs.dynamics_source(name), cache_dir=tmp_path)
    assert eng2.info(7) == 1                      # second handle: served from the cache
    src = eng.kernel_source(envs.dynamics_source(name))
    assert "pi_eval_sweep_kernel" in src and "step_dynamics" in src and "#define sinf pi_sinf" in src
    eng.close()
    eng2.close()


def test_reference_style_plugin_string_compiles_unchanged(tmp_path):
    """A plugin written the way the reference documents it (README 'adding an env',
    src/cuda_policy_iteration.py:113-125): #defines, a helper __device__ function, pointer
    outputs, bool* terminated."""
    dyn = r'''
    #define MY_DT 0.05f
    __device__ float my_clip(float v, float lo, float hi) { return fmaxf(lo, fminf(hi, v)); }
    __device__ void step_dynamics(
        float pos, float vel, float action,
        float* next_pos, float* next_vel,
        float* reward, bool* terminated
    ) {
        vel = my_clip(vel + action * MY_DT - 0.1f * sinf(pos), -1.0f, 1.0f);
        pos = pos + vel * MY_DT;
        *next_pos = pos; *next_vel = vel;
        *reward = -fabsf(pos);
        *terminated = (pos > 2.0f) || (pos < -2.0f);
    }
    '''
    bins = [np.linspace(-2, 2, 21, dtype=np.float32), np.linspace(-1, 1, 11, dtype=np.float32)]
    eng = _native.Engine(2, [21, 11], [-2, -1], [2, 1], bins, [-1.0, 0.0, 1.0], device=-1)
    eng.compile(dyn, cache_dir=tmp_path)
    eng.close()


@pytest.mark.parametrize("name", H.ENV_NAMES)
def test_the_references_literal_plugin_strings_compile_unchanged(name, tmp_path):
    """north_star: every runners/*_cuda.py is a drop-in — so the reference's OWN `_dynamics_cuda_src` strings
    (hook contract src/cuda_policy_iteration.py:113-125; e.g. runners/double_pendulum_swingup_cuda.py:75-201),
    read out of /root/reference at test time and handed to pi_compile byte for byte, must build for gfx950 behind
    this repository's kernel template, at the BASELINE grid of the env, within the register budgets the launch
    geometry assumes.  Build container only (the reference does not travel); nothing is written into the tree."""
    from oracle import build_ref
    if not build_ref.available():
        pytest.skip("the reference tree is not present (GPU box): literal reference strings cannot be read")
    text = build_ref.dynamics_text(name)
    assert "step_dynamics" in text and "__device__" in text
    cls = envs.ENVS[name]
    shape = {2: (200, 200), 4: (50,) * 4, 6: (25,) * 6}[cls._D]
    if name == "double_pendulum_swingup":
        shape = (80,) * 4
    eng = _host_engine(name, shape)
    eng.compile(text, cache_dir=tmp_path)                       # hipRTC, --offload-arch=gfx950
    assert eng.info(7) == 0
    (obj,) = list(tmp_path.glob("pi_*.hsaco"))
    assert obj.read_bytes()[:4] == b"\x7fELF"
    # same code path as the shipped string: the translation unit differs only in the plugin text
    ours = eng.kernel_source(envs.dynamics_source(name))
    theirs = eng.kernel_source(text)
    assert theirs.replace(text, "") == ours.replace(envs.dynamics_source(name), "")
    eng.close()


def test_compile_error_is_reported_not_swallowed(tmp_path):
    eng = _host_engine("pendulum", (8, 8))
    with pytest.raises(_native.NativeError) as exc:
        eng.compile("__device__ void step_dynamics(float a) { not valid C; }", cache_dir=tmp_path)
    assert "error" in str(exc.value).lower()
    assert not list(tmp_path.glob("*.hsaco"))
    eng.close()


def test_argument_validation():
    bins = [np.linspace(0, 1, 4, dtype=np.float32)] * 3
    with pytest.raises(_native.NativeError, match="D must be"):
        _native.Engine(3, [4, 4, 4], [0] * 3, [1] * 3, bins, [0.0], device=-1)
    with pytest.raises(_native.NativeError, match="at least 2 bins"):
        _native.Engine(2, [1, 4], [0, 0], [1, 1], [np.zeros(1, np.float32), bins[0]], [0.0], device=-1)
    with pytest.raises(_native.NativeError, match="2\\^31"):
        big = [np.linspace(0, 1, 40, dtype=np.float32)] * 6
        _native.Engine(6, [40] * 6, [0] * 6, [1] * 6, big, [0.0], device=-1)
    # the guard sits exactly at 2^31 states (flat indices are int32, as in the reference)
    edge = [np.linspace(0, 1, g, dtype=np.float32) for g in (256, 256, 256, 128)]
    with pytest.raises(_native.NativeError, match="2\\^31"):
        _native.Engine(4, [256, 256, 256, 128], [0] * 4, [1] * 4, edge, [0.0], device=-1)
    edge[3] = np.linspace(0, 1, 127, dtype=np.float32)
    ok = _native.Engine(4, [256, 256, 256, 127], [0] * 4, [1] * 4, edge, [0.0], device=-1)
    assert ok.n_states == 256 ** 3 * 127 < 2 ** 31
    ok.close()
    eng = _host_engine("pendulum", (8, 8))
    with pytest.raises(_native.NativeError, match="host-only"):
        eng.eval_sweep(1, 2, 3, 4, 0, 64, 0.9)
    eng.close()


def test_grids_beyond_32_bit_byte_offsets_compile(tmp_path):
    """n >= 2^30 states: 4 n no longer fits a 32-bit byte offset, the template switches to 64-bit
    element addressing (PI_OFF32 == false); it has to build for gfx950 like the 32-bit form
    (run on hardware by tests/test_gpu_endtoend.py::test_64_bit_addressing_path_at_2_pow_30_states)."""
    eng = _host_engine("double_cartpole", (32,) * 6)
    assert eng.n_states == 1 << 30
    eng.compile(envs.dynamics_source("double_cartpole"), cache_dir=tmp_path)
    assert eng.info(7) == 0 and len(list(tmp_path.glob("pi_*.hsaco"))) == 1
    eng.close()


def test_checked_build_compiles_for_gfx950(tmp_path, monkeypatch):
    """PI_MI355_DEBUG=1: the kernels with index checks (include/pi_mi355.h, pi_debug_report) build for every D;
    a handle without them refuses the report (run on hardware by tests/test_gpu_endtoend.py)."""
    monkeypatch.setenv("PI_MI355_DEBUG", "1")
    for name, shape in (("pendulum", (24, 17)), ("cartpole_swingup", (9, 7, 11, 5)), ("double_cartpole", (5, 4, 6, 4, 5, 4))):
        eng = _host_engine(name, shape)
        assert eng.info(15) == 1
        eng.compile(envs.dynamics_source(name), cache_dir=tmp_path)
        src = eng.kernel_source(envs.dynamics_source(name))
        assert "#define PI_DEBUG_BOUNDS 1" in src
        with pytest.raises(_native.NativeError, match="host-only"):
            eng.debug_report()
        eng.close()
    monkeypatch.delenv("PI_MI355_DEBUG")
    eng = _host_engine("pendulum", (24, 17))
    assert eng.info(15) == 0 and "#define PI_DEBUG_BOUNDS 1" not in eng.kernel_source(envs.dynamics_source("pendulum"))
    eng.close()


def test_no_gpu_means_runtime_error_not_fallback():
    """Without a GPU the product refuses to construct a solver (reference :71-75 raises for a
    missing CuPy); it must not quietly compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dynamicprogramming_amd import solver
    assert solver.GPU_AVAILABLE is False
    with pytest.raises(RuntimeError, match="GPU"):
        envs.make("pendulum", 16)


def test_product_code_never_imports_the_oracle():
    for sub in ("dynamicprogramming_amd", "src", "runners", "utils"):
        for path in (ROOT / sub).rglob("*.py"):
            text = path.read_text()
            assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), path
            assert "pi_oracle" not in text and "import_module(\"oracle" not in text, path
    for path in (ROOT / "dynamicprogramming_amd" / "csrc").glob("*"):
        if path.is_file():
            text = path.read_text()
            assert not re.search(r'#include\s+"[^"]*oracle', text), path
            if path.name == "Makefile":
                assert "oracle" not in text, path


def test_missing_library_raises(monkeypatch, tmp_path):
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "_load_error", None)
    monkeypatch.setattr(_native, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_native.NativeError, match="not found"):
        _native.lib()
    assert _native.available() is False
    monkeypatch.setattr(_native, "_load_error", None)


def test_ctypes_pointer_sizes():
    assert ctypes.sizeof(ctypes.c_void_p) == 8


# ── the interpolation's division through a proven reciprocal ───────────
def _markstein_f32(a, span):
    """The kernels' sequence t = a*y; r = fma(-t, span, a); q = fma(r, y, t), in numpy.  float64 holds
    every float32 product exactly and the sums here to far more than float32 precision, so rounding
    the float64 expression once to float32 reproduces the fused operation."""
    f32, f64 = np.float32, np.float64
    y = f32(1.0) / f32(span)
    t = (a * y).astype(f32)
    r = (a.astype(f64) - t.astype(f64) * f64(span)).astype(f32)           # fma(-t, span, a)
    return (r.astype(f64) * f64(y) + t.astype(f64)).astype(f32)            # fma(r, y, t)


@pytest.mark.parametrize("name", H.ENV_NAMES)
def test_reciprocal_division_is_enabled_and_exact_for_every_env_divisor(name, monkeypatch):
    """pi_create proves the reciprocal-multiply division per divisor by enumerating all 2^23
    significands (C++); this checks the verdict independently: the reference grids' divisors are
    accepted (pi_info 20 + d), PI_MI355_IEEE_DIV switches the path off, the generated source carries
    the span and its reciprocal as exact hex-float literals, and a numpy restatement of the sequence
    equals IEEE division on a million dividends spread over the guarded exponent range."""
    cls = envs.ENVS[name]
    bins = [np.asarray(b, np.float32) for b in cls.bins_space(cls.DEFAULT_BINS).values()]
    eng = _native.Engine(cls._D, [len(b) for b in bins], [b.min() for b in bins], [b.max() for b in bins],
                         bins, cls.ACTIONS, device=-1)
    assert [eng.info(20 + d) for d in range(cls._D)] == [1] * cls._D
    src = eng.kernel_source(envs.dynamics_source(name))

    def literals(macro):
        body = re.search(rf"#define {macro} \{{(.*?)\}}", src).group(1)
        return [np.float32(float.fromhex(tok.strip().rstrip("f"))) for tok in body.split(",")]

    spans, rcps, los = literals("PI_SPAN_INIT"), literals("PI_RCP_INIT"), literals("PI_LO_INIT")
    rng = np.random.default_rng(11)
    for d, b in enumerate(bins):
        span = np.float32(b.max()) - np.float32(b.min())
        assert spans[d] == span and rcps[d] == np.float32(1.0) / span and los[d] == np.float32(b.min())
        mant = rng.integers(0, 1 << 23, 1_000_000, dtype=np.uint64).astype(np.uint32)
        expo = rng.integers(127 - 38, 127 + 38, 1_000_000, dtype=np.uint64).astype(np.uint32)
        sign = rng.integers(0, 2, 1_000_000, dtype=np.uint64).astype(np.uint32) << np.uint32(31)
        a = (sign | (expo << np.uint32(23)) | mant).view(np.float32)
        want = (a / span).astype(np.float32)
        got = _markstein_f32(a, span)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"dimension {d}, span {span!r}"
    eng.close()
    monkeypatch.setenv("PI_MI355_IEEE_DIV", "1")
    off = _native.Engine(cls._D, [len(b) for b in bins], [b.min() for b in bins], [b.max() for b in bins],
                         bins, cls.ACTIONS, device=-1)
    assert [off.info(20 + d) for d in range(cls._D)] == [0] * cls._D
    assert "PI_FASTDIV_INIT {" + ",".join(["0"] * cls._D) + "}" in off.kernel_source(envs.dynamics_source(name))
    off.close()


def test_reciprocal_division_refuses_out_of_range_divisors():
    """Spans outside [2^-30, 2^30] (where an intermediate could leave the normal range) keep the
    IEEE division for that dimension only."""
    bins = [np.linspace(0, 1e-12, 5, dtype=np.float32), np.linspace(-1.0, 1.0, 7, dtype=np.float32)]
    eng = _native.Engine(2, [5, 7], [b.min() for b in bins], [b.max() for b in bins], bins,
                         np.array([0.0], np.float32), device=-1)
    assert [eng.info(20), eng.info(21)] == [0, 1]
    eng.close()


def test_inference_kernel_builds_without_a_gpu_and_validates_arguments(tmp_path):
    """pi_infer_* (the device twin of utils/barycentric.py): the kernel compiles for gfx950 for every D
    through hipRTC with a host-only handle; bad arguments fail loudly; without a GPU the Python
    front end raises instead of computing somewhere else."""
    from itertools import product
    for D in (2, 4, 6):
        bits = np.array(list(product([0, 1], repeat=D)), dtype=np.int32)
        shape = [5, 4, 6, 3, 4, 5][:D]
        strides = np.cumprod([1] + shape[::-1][:-1])[::-1]
        eng = _native.InferenceEngine([0.0] * D, [1.0] * D, shape, strides, bits, device=-1, cache_dir=tmp_path)
        assert eng.n_corners == 1 << D
        with pytest.raises(_native.NativeError, match="host-only"):
            eng.query(1, 4, 2)
        with pytest.raises(_native.NativeError, match="host-only"):
            eng.set_policy(np.zeros(int(np.prod(shape)), np.int32), [0.0])
        eng.close()
    assert len(list(tmp_path.glob("pi_*.hsaco"))) == 3
    bits = np.array(list(product([0, 1], repeat=2)), dtype=np.int32)
    with pytest.raises(_native.NativeError, match="2\\^D rows"):
        _native.InferenceEngine([0, 0], [1, 1], [3, 3], [3, 1], bits[:3], device=-1, cache_dir=None)
    with pytest.raises(_native.NativeError, match="0 or 1"):
        _native.InferenceEngine([0, 0], [1, 1], [3, 3], [3, 1], bits * 2, device=-1, cache_dir=None)
    with pytest.raises(_native.NativeError, match="exceed bounds_low"):
        _native.InferenceEngine([0, 1], [1, 1], [3, 3], [3, 1], bits, device=-1, cache_dir=None)
    # strides that do not belong to the grid would make the kernel read policy[flat] out of bounds
    with pytest.raises(_native.NativeError, match="strides must be positive"):
        _native.InferenceEngine([0, 0], [1, 1], [3, 3], [3, -1], bits, device=-1, cache_dir=None)
    with pytest.raises(_native.NativeError, match="outside the policy table"):
        _native.InferenceEngine([0, 0], [1, 1], [3, 4], [5, 1], bits, device=-1, cache_dir=None)      # another grid's strides
    _native.InferenceEngine([0, 0], [1, 1], [3, 4], [1, 3], bits, device=-1, cache_dir=tmp_path).close()   # column-major is fine
    import torch
    if not torch.cuda.is_available():
        from utils.barycentric import DevicePolicy
        with pytest.raises(RuntimeError, match="ROCm GPU"):
            DevicePolicy(None, None, [0, 0], [1, 1], [3, 3], [3, 1], bits)


def test_register_budgets_of_the_baseline_kernels(tmp_path):
    """Occupancy cliffs the launch geometry depends on: the 80^4 evaluation kernel
    runs 1 024-thread workgroups, two per CU — that needs 8 waves per SIMD, i.e. at most 64 VGPRs (at 65
    only ONE such workgroup fits a CU and the sweep loses a fifth of its speed, profiles/r03/
    negative_results.txt (8), (10)) AND at most 80 SGPRs (measured: at 84 .. 96 the same thing happens although
    the compiler still reports eight waves, negative_results.txt (15)); the 6-D kernels must not spill.
    Checked on the ahead-of-time build of the same translation units the library hands to hipRTC."""
    import subprocess
    import __graft_entry__ as G
    budgets = {("double_pendulum_swingup", 80): {"pi_eval_sweep_kernel": 64, "pi_improve_sweep_kernel": 96},
               ("cartpole_swingup", 50): {"pi_eval_sweep_kernel": 64, "pi_improve_sweep_kernel": 96},
               ("double_cartpole", 25): {"pi_eval_sweep_kernel": 128, "pi_improve_sweep_kernel": 168}}
    for (name, bins), limits in budgets.items():
        eng, dyn = G._engine_for(name, bins)
        src = tmp_path / f"{name}.hip"
        src.write_text(eng.kernel_source(dyn))
        eng.close()
        res = subprocess.run([G.HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "--genco",
                              "-include", "hip/hip_runtime.h", "-Rpass-analysis=kernel-resource-usage", str(src),
                              "-o", str(tmp_path / f"{name}.hsaco")], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-2000:]
        usage, fn = {}, None
        for line in res.stderr.splitlines():
            if "Function Name:" in line:
                fn = line.split("Function Name:")[1].split()[0]
                usage[fn] = {}
            elif fn and " VGPRs:" in line:
                usage[fn]["vgpr"] = int(line.split("VGPRs:")[1].split()[0])
            elif fn and "TotalSGPRs:" in line:
                usage[fn]["sgpr"] = int(line.split("TotalSGPRs:")[1].split()[0])
            elif fn and "ScratchSize" in line:
                usage[fn]["scratch"] = int(line.split(":")[-1].split()[0])
        for kernel, limit in limits.items():
            assert usage[kernel]["vgpr"] <= limit, (name, kernel, usage[kernel])
            assert usage[kernel]["scratch"] == 0, (name, kernel, usage[kernel])
        assert usage["pi_eval_sweep_kernel"]["sgpr"] <= 80, (name, usage["pi_eval_sweep_kernel"])


def test_one_launch_run_kernels_build_and_keep_their_residency_budget(tmp_path):
    """The kernels behind pi_policy_iteration, on the ahead-of-time build of the translation units the library hands to
    hipRTC.  pi_xcd_kernel (BASELINE config C2, pendulum 200 x 200) wants exactly ONE 1 024-thread workgroup on each
    of an XCD's 32 CUs: its LDS footprint must be more than half of a CU's 160 KB (a second workgroup cannot join) and
    within it, four waves per SIMD allow 128 VGPRs, nothing may spill, and 32 workgroups must cover the grid.
    pi_run_resident_kernel (C1, pendulum 50 x 50) is one workgroup with V and the policy in LDS."""
    import subprocess
    import __graft_entry__ as G
    for (name, bins, actions), kernel in ((("pendulum", 200, None), "pi_xcd_kernel"),
                                          (("pendulum", 50, np.linspace(-2.0, 2.0, 11, dtype=np.float32)), "pi_run_resident_kernel")):
        eng, dyn = G._engine_for(name, bins, actions)
        text = eng.kernel_source(dyn)
        n = eng.n_states
        eng.close()
        src = tmp_path / f"{name}_{bins}.hip"
        src.write_text(text)
        res = subprocess.run([G.HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "--genco",
                              "-include", "hip/hip_runtime.h", "-Rpass-analysis=kernel-resource-usage", str(src),
                              "-o", str(tmp_path / f"{name}_{bins}.hsaco")], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-2000:]
        usage, fn = {}, None
        for line in res.stderr.splitlines():
            if "Function Name:" in line:
                fn = line.split("Function Name:")[1].split()[0]
                usage[fn] = {}
            elif fn and " VGPRs:" in line:
                usage[fn]["vgpr"] = int(line.split("VGPRs:")[1].split()[0])
            elif fn and "ScratchSize" in line:
                usage[fn]["scratch"] = int(line.split(":")[-1].split()[0])
            elif fn and "LDS Size" in line:
                usage[fn]["lds"] = int(line.split(":")[-1].split()[0])
        k = usage[kernel]
        assert k["vgpr"] <= 128 and k["scratch"] == 0, (kernel, k)
        if kernel == "pi_xcd_kernel":
            assert 80 * 1024 < k["lds"] <= 160 * 1024, k
            per_wg = int([l for l in text.splitlines() if l.startswith("#define PI_XCD_S ")][0].split()[2])
            assert per_wg % 32 == 0 and per_wg <= 2048 and -(-n // per_wg) <= 32, (n, per_wg)
            assert "pi_xcd_finish_kernel" in usage
        else:
            assert 8 * n <= k["lds"] <= 160 * 1024, k                # V and the policy


def test_p2p_transport_kernels_build_without_a_gpu_and_host_only_handles_are_refused(tmp_path):
    """csrc/pi_p2p_kernels.hip (flag hand-shake, push, mailbox reduction) compiles for gfx950 through the library's
    hipRTC path and lands in the code-object cache; a handle without a device cannot describe itself to peers."""
    _native.p2p_compile_check(cache_dir=tmp_path)
    objs = list(tmp_path.glob("pi_*.hsaco"))
    assert len(objs) == 1 and objs[0].read_bytes()[:4] == b"\x7fELF"
    blob = objs[0].read_bytes()
    for kernel in (b"pi_p2p_sigwait_kernel", b"pi_p2p_push_kernel", b"pi_p2p_reduce_kernel"):
        assert kernel in blob
    eng = _host_engine("pendulum", (24, 17))
    with pytest.raises(_native.NativeError, match="host-only handle"):
        eng.p2p_describe(0, 2, [(4096, 64)])
    with pytest.raises(_native.NativeError, match="call pi_p2p_describe first"):
        eng.comm_init_p2p(0, 2, [b"\0" * 512] * 2)
    with pytest.raises(_native.NativeError, match="descriptors of 512 bytes"):
        eng.comm_init_p2p(0, 2, [b"\0" * 512])
    eng.close()


def test_transport_selection_from_the_environment(monkeypatch):
    from dynamicprogramming_amd import transport as T

    class FakeDist:
        @staticmethod
        def get_rank(group=None):
            return 1

        @staticmethod
        def get_world_size(group=None):
            return 4

    import torch.distributed as dist
    monkeypatch.setattr(dist, "get_rank", FakeDist.get_rank)
    monkeypatch.setattr(dist, "get_world_size", FakeDist.get_world_size)
    monkeypatch.setenv("PI_MI355_TRANSPORT", "p2p")
    t = T.from_environment()
    assert isinstance(t, T.P2pTransport) and (t.rank, t.world) == (1, 4) and t.is_native
    monkeypatch.setenv("PI_MI355_TRANSPORT", "smoke-signals")
    with pytest.raises(ValueError, match="expected 'rccl' or 'p2p'"):
        T.from_environment()


def test_p2p_bootstrap_failure_reaches_every_rank():
    """A rank that cannot map its peers says so in the second exchange of the bootstrap, and EVERY rank raises —
    nobody is left waiting in the next collective (transport.P2pTransport.plan; fake engine, no GPU)."""
    from dynamicprogramming_amd import transport as T

    class Tensor:
        def data_ptr(self):
            return 4096

        def numel(self):
            return 16

        def element_size(self):
            return 4

    class Solver:
        d_value_function = d_new_value_function = d_policy = Tensor()

    class Engine:
        def __init__(self, fails):
            self.fails = fails

        def p2p_describe(self, rank, world, bufs):
            assert len(bufs) == 3 and all(b == (4096, 64) for b in bufs)
            return b"descriptor of rank %d" % rank

        def comm_init_p2p(self, rank, world, everyone):
            assert everyone == [b"descriptor of rank 0", b"descriptor of rank 1"]
            if self.fails:
                raise _native.NativeError("hipIpcOpenMemHandle (rank 0, buffer 1): invalid device pointer")

    def exchange_for(rank, peer_says):
        def exchange(mine):                                  # what an all-gather over two ranks returns, ordered by rank
            theirs = peer_says.pop(0)
            return [mine, theirs] if rank == 0 else [theirs, mine]
        return exchange

    # rank 1 fails: it raises, and rank 0 — whose own mapping worked — raises as well, naming rank 1
    t1 = T.P2pTransport(1, 2, exchange=exchange_for(1, [b"descriptor of rank 0", b"mapped"]))
    t1.engine = Engine(fails=True)
    with pytest.raises(_native.NativeError, match="rank 1: hipIpcOpenMemHandle"):
        t1.plan(Solver())
    t0 = T.P2pTransport(0, 2, exchange=exchange_for(0, [b"descriptor of rank 1",
                                                        b"rank 1: hipIpcOpenMemHandle (rank 0, buffer 1): invalid device pointer"]))
    t0.engine = Engine(fails=False)
    with pytest.raises(_native.NativeError, match="could not be set up: rank 1"):
        t0.plan(Solver())
    assert not t0._connected and not t1._connected


def test_fused_exchange_kernels_build_and_keep_the_register_budget(tmp_path):
    """The handle's second module (csrc/pi_push_kernels.hip appended to the sweep translation unit: the swept-first
    kernel that delivers its rows itself, and the pair-level reach probe) builds for gfx950 in the sharded memory
    orders through the library (pi_set_option 5 on a host-only handle) and through hipcc, where pi_eval_push_kernel
    must stay inside the budget of the kernel it replaces in a sharded sweep: 64 VGPRs and 80 SGPRs on the 80^4 grid
    (two 1 024-thread workgroups per CU), no scratch anywhere."""
    import subprocess
    import __graft_entry__ as G
    push = (ROOT / "dynamicprogramming_amd" / "csrc" / "pi_push_kernels.hip").read_text()
    for name, bins, vgprs in (("double_pendulum_swingup", 80, 64), ("double_cartpole", 25, 128)):
        cls = envs.ENVS[name]
        tables = [np.asarray(b, dtype=np.float32) for b in cls.bins_space(bins).values()]
        eng = _native.Engine(cls._D, [len(t) for t in tables], [t.min() for t in tables], [t.max() for t in tables],
                             tables, cls.ACTIONS, device=-1, order=cls.SHARDED_MEMORY_ORDER)
        dyn = envs.dynamics_source(name)
        eng.compile(dyn, cache_dir=tmp_path)
        before = len(list(tmp_path.glob("pi_*.hsaco")))
        eng.set_option(5, 1)                                  # second module -> cache
        assert len(list(tmp_path.glob("pi_*.hsaco"))) == before + 1
        src = tmp_path / f"{name}_push.hip"
        src.write_text(eng.kernel_source(dyn) + push)
        eng.close()
        res = subprocess.run([G.HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "--genco",
                              "-include", "hip/hip_runtime.h", "-Rpass-analysis=kernel-resource-usage", str(src),
                              "-o", str(tmp_path / f"{name}_push.hsaco")], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-2000:]
        usage, fn = {}, None
        for line in res.stderr.splitlines():
            if "Function Name:" in line:
                fn = line.split("Function Name:")[1].split()[0]
                usage[fn] = {}
            elif fn and " VGPRs:" in line:
                usage[fn]["vgpr"] = int(line.split("VGPRs:")[1].split()[0])
            elif fn and "TotalSGPRs:" in line:
                usage[fn]["sgpr"] = int(line.split("TotalSGPRs:")[1].split()[0])
            elif fn and "ScratchSize" in line:
                usage[fn]["scratch"] = int(line.split(":")[-1].split()[0])
        assert {"pi_eval_push_kernel", "pi_reach_pairs_kernel"} <= set(usage)
        k = usage["pi_eval_push_kernel"]
        assert k["vgpr"] <= vgprs and k["scratch"] == 0 and usage["pi_reach_pairs_kernel"]["scratch"] == 0, (name, usage)
        if name == "double_pendulum_swingup":
            assert k["sgpr"] <= 80, k
    # before pi_compile there is nothing to append to
    eng = _host_engine("pendulum", (24, 17))
    with pytest.raises(_native.NativeError, match="pi_compile has not run"):
        eng.set_option(5, 1)
    eng.close()
