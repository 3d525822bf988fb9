"""Repository hygiene the grading contract depends on (task statement ③):
the product never touches the oracle, and nothing derived from /root/reference sits in the tree
that is snapshotted to the GPU box."""
from __future__ import annotations

import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
PRODUCT_DIRS = ["dynamicprogramming_amd", "src", "runners", "utils"]


def _product_sources():
    for d in PRODUCT_DIRS:
        for p in (ROOT / d).rglob("*"):
            if p.suffix in (".py", ".cpp", ".h", ".hip") and "__pycache__" not in p.parts:
                yield p


def test_product_code_never_imports_the_oracle():
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b|from\s+tests\b|import\s+tests\b)", re.M)
    offenders = [str(p.relative_to(ROOT)) for p in _product_sources() if p.suffix == ".py"
                 and pat.search(p.read_text())]
    assert not offenders, offenders
    inc = re.compile(r"#\s*include\s*[\"<][^\">]*(oracle|ref_driver)", re.M)
    native = [str(p.relative_to(ROOT)) for p in _product_sources() if p.suffix in (".cpp", ".h", ".hip")
              and inc.search(p.read_text())]
    assert not native, native


def test_product_has_one_backend_and_no_test_transport():
    """The gloo transport and the checker backend are test infrastructure: nothing in the product
    sources names gloo or offers a public way to swap the sweep backend."""
    offenders = [str(p.relative_to(ROOT)) for p in _product_sources()
                 if re.search(r"gloo|backend_factory|TorchDistTransport", p.read_text())]
    assert not offenders, offenders
    import inspect
    from dynamicprogramming_amd.solver import _CudaPolicyIterationBase
    params = inspect.signature(_CudaPolicyIterationBase.__init__).parameters
    assert list(params) == ["self", "bins_space", "action_space", "config", "device", "process_group", "transport"]
    # no seam at all: the class has no attribute through which another backend could be installed (the CPU tests patch
    # the module's own names while they construct a solver: tests/helpers.py)
    assert not any("backend_cls" in name or "backend_factory" in name for name in dir(_CudaPolicyIterationBase))
    import dynamicprogramming_amd.solver as S
    src = Path(S.__file__).read_text()
    assert src.count("HipSweepBackend(") == 1 and "_sweep_backend_cls" not in src


def test_no_reference_derived_artifacts_in_the_tree():
    """The reference-text checker (oracle/build_ref.py) is built under $TMPDIR; no shared object,
    include file or cached text of it may exist under the repository root."""
    assert not list(ROOT.rglob("libref_*.so")), "reference-derived shared objects in the tree"
    ref_dir = ROOT / "oracle" / "_ref"
    assert not ref_dir.exists() or not any(ref_dir.iterdir())
    from oracle import build_ref
    assert ROOT not in build_ref.REF_DIR.resolve().parents and build_ref.REF_DIR.resolve() != ROOT
    # nothing in the tree reads /root/reference at run time except the fixture generator and its helper
    allowed = {"oracle/build_ref.py", "tests/golden/make_golden.py", "tests/golden/make_barycentric_golden.py",
               "tests/golden/make_step_python_golden.py", "tests/test_hygiene.py"}
    offenders = []
    for p in ROOT.rglob("*.py"):
        rel = str(p.relative_to(ROOT))
        if "__pycache__" in p.parts or rel.startswith(("gpurun_out/", ".")) or rel in allowed:
            continue
        if re.search(r"[\"']/root/reference", p.read_text()):
            offenders.append(rel)
    assert not offenders, offenders


def test_committed_bench_evidence_matches_the_kernels_and_the_contract():
    """profiles/r*/bench_<config>.json are the lines `python bench.py [--env ...]` printed: each carries
    every field of the bench contract; its roofline block was computed from counters of the same
    config and (when the kernels have not changed since) the same kernel source; the fraction
    reported is that of the unit named as the bound, every unit is priced against its hardware peak
    and no fraction of the dominant kernel exceeds 1."""
    import json
    import pytest
    from dynamicprogramming_amd import _native
    latest = sorted(ROOT.glob("profiles/r*/bench_c4.json"))[-1].parent
    lines = sorted(latest.glob("bench_c*.json"))
    assert (latest / "bench_c4.json") in lines
    stale = []
    for path in lines:
        d = json.loads(path.read_text())
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                    "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert key in d, (path.name, key)
        assert d["n_gpus"] == 1 and d["higher_is_better"] is True and "workload" in d["config"]
        assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
        c = d["config"]
        backups = d["steps"] * c["states"] * (c["eval_sweeps_per_step"] + c["improve_sweeps_per_step"] * c["actions"])
        assert abs(d["ms_per_step"] * d["steps"] * 1e-3 * d["value"] - backups) < 1e-6 * backups      # value = backups / elapsed
        cb = d["cpu_baseline"]
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in cb, (path.name, key)
        if int(latest.name[1:]) >= 4:
            # round 4: the line says what it measured — terminal states counted or not, which thread count the CPU
            # figure is for, and what "compulsory" bytes mean on this grid
            assert d["value_nonterminal"] <= d["value"] * (1 + 1e-12) and c["nonterminal_states"] <= c["states"]
            assert abs(d["value_nonterminal"] * c["states"] - d["value"] * c["nonterminal_states"]) <= 1e-9 * d["value"] * c["states"]
            assert "terminal" in c["workload"]
            for key in ("value_all_affinity_threads", "affinity_threads", "value_16_threads", "value_1_thread"):
                assert key in cb, (path.name, key)
            assert cb["value"] == max(cb["value_all_affinity_threads"], cb["value_16_threads"])
            assert d["roofline"]["compulsory_bytes_per_state"] in (12.0, 13.0)
            assert "memory_order" in d["check"]
        r = d["roofline"]
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert key in r, (path.name, key)
        if r["kernel_source_hash"] != _native.kernel_source_hash():
            stale.append(path.name)          # evidence of an older kernel version (bench.py withholds it then)
            continue
        assert r["profile"] and r["frac"] is not None, path.name
        prof = json.loads((ROOT / r["profile"]).read_text())
        assert prof["kernel_source_hash"] == r["kernel_source_hash"] and prof["states"] == c["states"]
        units = r["units"]
        if int(latest.name[1:]) >= 5:
            # round 5: the headline fraction is the one the north star names — measured HBM-side bytes (FETCH_SIZE x the
            # calibrated 2.0 + WRITE_SIZE) over the live launch time against 8 TB/s — with the uncorrected figure and the
            # most utilised unit beside it
            assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
            assert r["frac"] == r["frac_hbm"] == r["hbm_frac"] == units["hbm"]["frac"]
            assert r["fetch_correction"]["factor"] == 2.0 and (ROOT / r["fetch_correction"]["source"]).exists()
            top = max(units, key=lambda u: units[u]["frac"])
            assert r["most_utilised_unit"]["name"] == {"valu": "valu-issue", "l1": "l1-load-issue", "hbm": "hbm"}[top]
            assert r["most_utilised_unit"]["frac"] == units[top]["frac"]
            assert r["north_star_target"]["hbm_frac_at_least_0.40"] == (r["frac"] >= 0.40)
        else:
            name = {"valu-issue": "valu", "l1-load-issue": "l1", "hbm": "hbm"}[r["bound"]]
            assert r["frac"] == max(u["frac"] for u in units.values()) == units[name]["frac"]
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert units["valu"]["peak"] == 1228.8 and units["hbm"]["peak"] == 8000.0 and abs(units["l1"]["peak"] - 38.4) < 1e-9
        for u in units.values():
            assert 0.0 < u["frac"] <= 1.0, (path.name, u)
        # ... and every fraction can be recomputed from the committed profile and the line's own launch time
        k = prof["kernels"][r["kernel"]]
        sec = r["avg_launch_ms"] * 1e-3
        valu = k["valu_insts_per_wave"] * k["counters"]["SQ_WAVES"] / sec / 1e9
        l1 = k["vmem_rd_insts_per_wave"] * k["counters"]["SQ_WAVES"] / sec / 1e9
        hbm = (2.0 * k["FETCH_SIZE_bytes"] + k["WRITE_SIZE_bytes"]) / sec / 1e9
        for name, got in (("valu", valu), ("l1", l1), ("hbm", hbm)):
            assert abs(units[name]["achieved"] - got) <= 1e-9 * got, (path.name, name)
            assert abs(units[name]["frac"] - got / units[name]["peak"]) <= 1e-12, (path.name, name)
        assert abs(r["traffic"] - (2.0 * k["FETCH_SIZE_bytes"] + k["WRITE_SIZE_bytes"])) < 1.0
        assert 0.0 < r["hbm_frac"] <= 1.0 and r["traffic"] > 0
        if int(latest.name[1:]) >= 5:
            raw = (k["FETCH_SIZE_bytes"] + k["WRITE_SIZE_bytes"]) / sec / 1e9
            assert abs(r["hbm_frac_raw"] - raw / 8000.0) <= 1e-12 and abs(r["traffic_raw"] - raw * sec * 1e9) < 1.0
            assert r["hbm_frac_raw"] < r["frac_hbm"]
            if path.name == "bench_c4.json" and d.get("extra_configs") is not None:
                # the default command also times the other single-GPU BASELINE configs
                labels = [x["label"] for x in d["extra_configs"]] + [x["label"] for x in d.get("extra_configs_skipped", [])]
                assert set(labels) == {"c2", "c3", "c5", "c5_swingup"}
                for x in d["extra_configs"]:
                    per_step = x["states"] * (c["eval_sweeps_per_step"] + c["improve_sweeps_per_step"] * x["actions"])
                    assert abs(x["backups_per_s"] * x["ms_per_step"] * 1e-3 - per_step) < 1e-6 * per_step
                    assert x["eval_ms"] > 0 and x["improve_ms"] > 0 and "workload" in x
    if stale:
        # Evidence of another kernel version must not pass silently for the metric config: it fails unless
        # profiles/STALE_EVIDENCE_OK names the CURRENT kernel hash (tools/ack_stale_profiles.py) — a committed, visible
        # statement that the kernels have moved on and the profiles are being regenerated (bench.py withholds every
        # profile-derived figure meanwhile).  Secondary configs only skip.
        ack = ROOT / "profiles" / "STALE_EVIDENCE_OK"
        acknowledged = ack.exists() and ack.read_text().split()[:1] == [_native.kernel_source_hash()]
        if "bench_c4.json" in stale and not acknowledged:
            pytest.fail(f"kernels changed since {latest.name}/bench_c4.json was made: re-run tools/refresh_profiles.sh on a "
                        f"GPU box, or acknowledge with `python tools/ack_stale_profiles.py`")
        pytest.skip(f"kernels changed since {latest.name} was made ({', '.join(stale)}): "
                    "re-run tools/refresh_profiles.sh on a GPU box")


def test_tools_parse_and_name_only_existing_entry_points():
    """tools/*.py are run on GPU boxes only; here they must at least parse, and every attribute they take
    from the ctypes layer must exist there (a renamed Engine method would otherwise surface mid-profile)."""
    import ast
    from dynamicprogramming_amd import _native
    for path in sorted((ROOT / "tools").glob("*.py")):
        tree = ast.parse(path.read_text(), filename=str(path))
        for node in ast.walk(tree):
            if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id == "_native":
                assert hasattr(_native, node.attr), f"{path.name}: _native.{node.attr} does not exist"
