"""tools/collect_counters.py OUTDIR LABEL [LABEL...] — fold the rocprofv3 passes that
tools/profile_counters.sh wrote under OUTDIR/<label>_<pass>/ into one JSON per label:
per kernel, the mean over the last `PI_LAST` (default 20) dispatches of every counter, the mean
kernel duration from the kernel trace of the same passes, and the derived figures the roofline
needs (VALU instructions per wave, busy fraction, L2 hit rate, HBM-side bytes).  FETCH_SIZE_bytes / WRITE_SIZE_bytes are
stored AS COUNTED; the consumer (bench.py) applies the calibrated correction — x 2.0 on FETCH_SIZE, measured on the sweeps'
own load shapes and on the product's sweep with identity dynamics (tools/fetch_calibration.sh, profiles/r05/
fetch_calibration.txt), x 1.0 on WRITE_SIZE; `hbm_bytes_corrected` is recorded beside them with the factor used.
"""
import collections, csv, glob, json, os, subprocess, sys

out = sys.argv[1]
labels = sys.argv[2:]
LAST = int(os.environ.get("PI_LAST", "20"))
ENV = os.environ.get("PI_ENV", "double_pendulum_swingup")
BINS = int(os.environ.get("PI_BINS", "80"))


def fetch_factor():
    """The measured FETCH_SIZE correction: mean over the load shapes the sweeps use of the latest committed
    profiles/r*/fetch_calibration.json (tools/fetch_calibration.sh); 2.0 — the value every such measurement gave — without one."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*", "fetch_calibration.json")))
    if not files:
        return 2.0
    micro = json.load(open(files[-1])).get("micro", {})
    shapes = [micro[k]["fetch_factor"] for k in ("cal_overlap8", "cal_stream4_nt", "cal_stream1_nt", "cal_gather8_line")
              if "fetch_factor" in micro.get(k, {})]
    if not shapes or all(1.95 <= f <= 2.05 for f in shapes):
        return 2.0      # one 128-B line per request, tallied at 64 B (the 1.5 % below 2 of the overlapping pairs are extra REQUESTS)
    return round(sum(shapes) / len(shapes), 2)


FETCH_FACTOR = fetch_factor()


def head_hash():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dynamicprogramming_amd import _native
    return _native.kernel_source_hash()


def n_states():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dynamicprogramming_amd import envs
    n = 1
    for b in envs.ENVS[ENV].bins_space(BINS).values():
        n *= len(b)
    return n


def memory_order():
    """The memory order of the dimensions a single-rank solver of (ENV, BINS) uses (class default / PI_MI355_ORDER)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dynamicprogramming_amd import envs
    cls = envs.ENVS[ENV]
    s = object.__new__(cls)
    s.n_states, s._transport_arg, s._process_group = n_states(), False, None
    order = os.environ.get("PI_MI355_ORDER", "") != "auto" and s._choose_memory_order()
    return list(order) if order else list(range(cls._D))


def passes(label):
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    durations = collections.defaultdict(list)
    for d in sorted(glob.glob(os.path.join(out, f"{label}_*"))):
        for cf in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
            rows = list(csv.DictReader(open(cf)))
            per = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in rows:
                if r["Kernel_Name"].startswith("pi_"):
                    per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, cs in per.items():
                for c, v in cs.items():
                    counters[k][c] = v[-LAST:]
        for kf in glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")):
            per = collections.defaultdict(list)
            for r in csv.DictReader(open(kf)):
                if r["Kernel_Name"].startswith("pi_"):
                    per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
            for k, v in per.items():
                durations[k].extend(v[-LAST:])
    return counters, durations


def measured_valu_peak():
    """Best chip-wide wave-instruction rate of tools/valu_issue_bench.hip (same output directory tree)."""
    best = None
    for cand in glob.glob(os.path.join(os.path.dirname(out.rstrip("/")), "valu_issue.txt")) + \
            glob.glob(os.path.join(out, "valu_issue.txt")):
        for line in open(cand):
            f = line.split()
            if line.startswith("#") or len(f) < 6:
                continue
            try:
                g = float(f[-2])
            except ValueError:
                continue
            if f[0].startswith("v_") and (best is None or g > best):
                best = g
    return best


# cycles per wave64 instruction per SIMD at saturation (profiles/r02/valu_issue.txt): fp32
# fma/mul/add class, everything else, transcendentals
COST_A, COST_B, COST_C = 2.3, 4.15, 8.15
cal = None
for label in labels:
    counters, durations = passes(label)
    res = {"label": label, "kernel_source_hash": head_hash(), "dispatches_averaged": LAST,
           "env": ENV, "bins": BINS, "states": n_states(), "memory_order": memory_order(),
           "valu_peak_measured_Ginst_per_s": measured_valu_peak(), "kernels": {}}
    for k in sorted(counters):
        c = {name: sum(v) / len(v) for name, v in counters[k].items() if v}
        e = {"counters": c}
        if durations.get(k):
            e["ms_under_profiler"] = sum(durations[k]) / len(durations[k])
        w = c.get("SQ_WAVES")
        if w:
            if "SQ_INSTS_VALU" in c:
                e["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / w
            if "SQ_INSTS_SALU" in c:
                e["salu_insts_per_wave"] = c["SQ_INSTS_SALU"] / w
            if "SQ_INSTS_VMEM_RD" in c:
                e["vmem_rd_insts_per_wave"] = c["SQ_INSTS_VMEM_RD"] / w
            if "SQ_WAVE_CYCLES" in c:
                e["wave_cycles_per_wave_x4"] = 4 * c["SQ_WAVE_CYCLES"] / w
            if all(k in c for k in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32",
                                    "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_TRANS_F32")):
                a = (c["SQ_INSTS_VALU_FMA_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_ADD_F32"]) / w
                t = c["SQ_INSTS_VALU_TRANS_F32"] / w
                b = c["SQ_INSTS_VALU"] / w - a - t
                e["issue_cycles_model"] = {
                    "fp32_fma_mul_add_per_wave": a, "transcendental_per_wave": t, "other_valu_per_wave": b,
                    "cycles_per_inst": [COST_A, COST_B, COST_C],
                    "simd_cycles_per_wave": a * COST_A + b * COST_B + t * COST_C,
                    "note": "integer adds / and / xor issue at the fp32 rate but are counted as 'other': an upper estimate"}
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
            for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                         "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS"):
                if name in c:
                    e[name.lower() + "_frac_of_wave_cycles"] = c[name] / c["SQ_WAVE_CYCLES"]
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            e["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in c and c.get("TCP_TCC_READ_REQ_sum") is not None and c["TCP_TOTAL_CACHE_ACCESSES_sum"]:
            e["l1_hit_rate"] = 1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
        gui = c.get("GRBM_GUI_ACTIVE")
        if gui:
            cu_cycles = gui / 8.0 * 256.0                   # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            e["cycles_per_launch"] = gui / 8.0
            if e.get("ms_under_profiler"):
                e["clock_GHz"] = gui / 8.0 / (e["ms_under_profiler"] * 1e-3) / 1e9
            e["tcp_per_cu_cycle"] = {name: c[name] / cu_cycles for name in sorted(c)
                                     if name.startswith("TCP_") and "LATENCY" not in name}
            if w and "SQ_INSTS_VALU" in c:
                e["valu_insts_per_simd_cycle"] = c["SQ_INSTS_VALU"] / (cu_cycles * 4.0)
        if c.get("TCP_TCC_READ_REQ_LATENCY_sum") and c.get("TCP_TCC_READ_REQ_sum"):
            e["tcp_tcc_read_latency_cycles"] = c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"]
        if c.get("TCP_TCP_LATENCY_sum") and c.get("TCP_TOTAL_ACCESSES_sum"):
            e["tcp_latency_cycles_per_access"] = c["TCP_TCP_LATENCY_sum"] / c["TCP_TOTAL_ACCESSES_sum"]
        if "FETCH_SIZE" in c:
            e["FETCH_SIZE_bytes"] = c["FETCH_SIZE"] * 1024.0
        if "WRITE_SIZE" in c:
            e["WRITE_SIZE_bytes"] = c["WRITE_SIZE"] * 1024.0
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            e["fetch_correction_factor"] = FETCH_FACTOR
            e["hbm_bytes_corrected"] = FETCH_FACTOR * e["FETCH_SIZE_bytes"] + e["WRITE_SIZE_bytes"]
        res["kernels"][k] = e
    if label == "cal":
        cal = res
    json.dump(res, open(os.path.join(out, f"counters_{label}.json"), "w"), indent=1)
    print(label, json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters"}
                             for k, v in res["kernels"].items()}, indent=1))
