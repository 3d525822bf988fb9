"""
tools/collect_traffic.py OUTDIR — turn the PMC passes of tools/profile_bench.sh into per-launch
HBM traffic per kernel (bytes) and a calibration factor.

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  MI355X_MICROARCH.md §HBM: on gfx950
FETCH_SIZE counts TCC_EA0_RDREQ x 64 B and reads exactly 1/2 of the bytes of a wide coalesced
stream; other access widths must be calibrated on a known byte count in one's own access
pattern.  The calibration run is the same sweep kernel with identity dynamics, whose compulsory
read traffic is known: V + policy (4 B each) + terminal mask (1 B) per state.
"""
import collections, csv, glob, json, os, sys

out = sys.argv[1]


def per_kernel(pattern, counter):
    agg = collections.defaultdict(list)
    for cf in glob.glob(os.path.join(out, pattern, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(cf)):
            if r["Counter_Name"] == counter and r["Kernel_Name"].startswith("pi_"):
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


fetch, nf = per_kernel("pmc_FETCH_SIZE", "FETCH_SIZE")
write, _ = per_kernel("pmc_WRITE_SIZE", "WRITE_SIZE")
cal_f, _ = per_kernel("cal_FETCH_SIZE", "FETCH_SIZE")
cal_w, _ = per_kernel("cal_WRITE_SIZE", "WRITE_SIZE")
n = 80 ** 4
known_read = n * (4 + 4 + 1)          # identity-dynamics eval sweep: V, policy, mask read once
known_write = n * 4
res = {"states": n, "kernels": {}, "calibration": {}}
if "pi_eval_sweep_kernel" in cal_f:
    res["calibration"] = {
        "kernel": "pi_eval_sweep_kernel with identity dynamics (tools/gather_probe.py)",
        "known_read_bytes": known_read, "FETCH_SIZE_bytes": cal_f["pi_eval_sweep_kernel"],
        "read_factor": known_read / cal_f["pi_eval_sweep_kernel"],
        "known_write_bytes": known_write, "WRITE_SIZE_bytes": cal_w.get("pi_eval_sweep_kernel"),
    }
factor = res["calibration"].get("read_factor", 1.0)
for k in sorted(fetch):
    res["kernels"][k] = {
        "launches_sampled": nf[k],
        "FETCH_SIZE_bytes": fetch[k], "WRITE_SIZE_bytes": write.get(k),
        "hbm_bytes_raw": fetch[k] + (write.get(k) or 0.0),
        "hbm_bytes_corrected": fetch[k] * factor + (write.get(k) or 0.0),
    }
json.dump(res, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
