"""tools/fetch_calibration_fold.py OUTDIR — fold the rocprofv3 passes of tools/fetch_calibration.sh into
OUTDIR/fetch_calibration.json and a readable OUTDIR/fetch_calibration.txt: per kernel (access shape) the bytes it is KNOWN
to touch, what FETCH_SIZE / WRITE_SIZE reported, the factor to multiply the counter by, and the raw request counters.
"""
import collections, csv, glob, json, os, re, sys

out = sys.argv[1]


def passes(prefix):
    """kernel -> counter -> [values in dispatch order]; kernel -> [durations ms]"""
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    durations = collections.defaultdict(list)
    for d in sorted(glob.glob(os.path.join(out, f"{prefix}_p*"))):
        if not os.path.isdir(d):
            continue
        for cf in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
            per = collections.defaultdict(lambda: collections.defaultdict(dict))
            for row, r in enumerate(csv.DictReader(open(cf))):
                k = r["Kernel_Name"]
                if k.startswith("cal_") or k.startswith("pi_eval_sweep"):
                    per[k][r["Counter_Name"]][int(r.get("Dispatch_Id") or row)] = float(r["Counter_Value"])
            for k, cs in per.items():
                for c, byd in cs.items():
                    counters[k][c] = [byd[i] for i in sorted(byd)]
        for kf in glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")):
            per = collections.defaultdict(list)
            for r in csv.DictReader(open(kf)):
                k = r["Kernel_Name"]
                if k.startswith("cal_") or k.startswith("pi_eval_sweep"):
                    per[k].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6))
            for k, v in per.items():
                durations[k] = [ms for _, ms in sorted(v)]
    return counters, durations


def mean(v):
    return sum(v) / len(v) if v else None


res = {"micro": {}, "identity": []}
lines = []
known, what = {}, {}
for line in open(os.path.join(out, "fetch_cal_plain.log")):
    m = re.match(r"(cal_\w+)\s+known_bytes\s+(\d+)\s+([\d.]+) ms\s+([\d.]+) GB/s\s+# (.*)", line)
    if m:
        known[m.group(1)] = int(m.group(2))
        what[m.group(1)] = (float(m.group(3)), m.group(5))
counters, durations = passes("micro")
lines.append("FETCH_SIZE / WRITE_SIZE against known byte counts, per access shape (tools/fetch_calibration.hip; MI355X, gfx950)")
lines.append("buffer far larger than L2 and the Infinity Cache; every byte (or line) touched once per launch; mean of the last two of three launches")
lines.append("")
lines.append(f"{'kernel':18s} {'known MB':>10s} {'FETCH MB':>10s} {'factor':>7s} {'WRITE MB':>10s} {'factor':>7s} {'RDREQ':>11s} {'RDREQ_32B':>10s} {'BUBBLE':>9s} {'B/RDREQ':>8s} {'ms':>8s}  shape")
for k in sorted(known, key=lambda k: list(known).index(k)):
    c = {name: mean(v[-2:]) for name, v in counters.get(k, {}).items()}
    e = {"known_bytes": known[k], "what": what[k][1], "ms": what[k][0], "counters": c}
    f = c.get("FETCH_SIZE")
    w = c.get("WRITE_SIZE")
    is_store = k.startswith("cal_store")
    if f is not None:
        e["FETCH_SIZE_bytes"] = f * 1024.0
        if not is_store and f > 0:
            e["fetch_factor"] = known[k] / (f * 1024.0)
    if w is not None:
        e["WRITE_SIZE_bytes"] = w * 1024.0
        if is_store and w > 0:
            e["write_factor"] = known[k] / (w * 1024.0)
    rd = c.get("TCC_EA0_RDREQ_sum")
    if rd:
        e["known_bytes_per_read_request"] = known[k] / rd
    res["micro"][k] = e
    fmt = lambda v, s=1.0: "-" if v is None else f"{v * s:.1f}"
    lines.append(f"{k:18s} {known[k] / 1e6:10.1f} {fmt(f, 1024 / 1e6):>10s} {('%.3f' % e['fetch_factor']) if 'fetch_factor' in e else '-':>7s} "
                 f"{fmt(w, 1024 / 1e6):>10s} {('%.3f' % e['write_factor']) if 'write_factor' in e else '-':>7s} "
                 f"{fmt(rd):>11s} {fmt(c.get('TCC_EA0_RDREQ_32B_sum')):>10s} {fmt(c.get('TCC_BUBBLE_sum')):>9s} "
                 f"{('%.1f' % e['known_bytes_per_read_request']) if 'known_bytes_per_read_request' in e else '-':>8s} {what[k][0]:8.3f}  {what[k][1]}")

# the product's own sweep with identity dynamics: dispatches come in groups of `sweeps_each` per shape, in order
ident = None
for line in open(os.path.join(out, "fetch_cal_identity_plain.log")):
    if line.startswith("{"):
        ident = json.loads(line)
counters, durations = passes("identity")
if ident:
    per = ident["sweeps_each"]
    lines += ["", "pi_eval_sweep_kernel with identity dynamics (tools/fetch_calibration_identity.py): known = 8 B read + 4 B written per state",
              f"{'shape':26s} {'known rd MB':>11s} {'FETCH MB':>10s} {'factor':>7s} {'known wr MB':>11s} {'WRITE MB':>10s} {'factor':>7s} {'ms':>8s}"]
    kname = next((k for k in counters if k.startswith("pi_eval_sweep")), None)
    for j, s in enumerate(ident["identity_sweeps"]):
        sl = slice(j * per + 2, (j + 1) * per)                 # skip the first two launches of every shape
        c = {name: mean(v[sl]) for name, v in counters.get(kname, {}).items()} if kname else {}
        e = dict(s)
        e["counters"] = c
        if c.get("FETCH_SIZE"):
            e["FETCH_SIZE_bytes"] = c["FETCH_SIZE"] * 1024.0
            e["fetch_factor"] = s["known_read_bytes"] / e["FETCH_SIZE_bytes"]
        if c.get("WRITE_SIZE"):
            e["WRITE_SIZE_bytes"] = c["WRITE_SIZE"] * 1024.0
            e["write_factor"] = s["known_write_bytes"] / e["WRITE_SIZE_bytes"]
        res["identity"].append(e)
        g = lambda key: ("%.1f" % (e[key] / 1e6)) if key in e else "-"
        h = lambda key: ("%.3f" % e[key]) if key in e else "-"
        lines.append(f"{'x'.join(map(str, s['shape'])):26s} {s['known_read_bytes'] / 1e6:11.1f} {g('FETCH_SIZE_bytes'):>10s} {h('fetch_factor'):>7s} "
                     f"{s['known_write_bytes'] / 1e6:11.1f} {g('WRITE_SIZE_bytes'):>10s} {h('write_factor'):>7s} {s['ms_per_sweep']:8.3f}")
json.dump(res, open(os.path.join(out, "fetch_calibration.json"), "w"), indent=1)
open(os.path.join(out, "fetch_calibration.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
