"""Summarise tools/sweep_knobs.sh output: per variant, kernel times from bench JSON and mean
PMC counters per pi_* kernel."""
import collections, csv, glob, json, os, sys
out = sys.argv[1]
for jf in sorted(glob.glob(os.path.join(out, "bench_*.json"))):
    tag = os.path.basename(jf)[6:-5]
    line = [l for l in open(jf) if l.startswith("{")]
    if not line:
        print(tag, "FAILED"); continue
    d = json.loads(line[0])
    ks = {k: round(v["avg_launch_ms"], 3) for k, v in d["kernels"].items()}
    print(f"{tag:40s} value {d['value']:.3e}  {ks}")
    for cf in glob.glob(os.path.join(out, f"pmc_{tag}", "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(cf)):
            if r["Kernel_Name"].startswith("pi_"):
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            m = {c: sum(x) / len(x) for c, x in v.items()}
            hit = m.get("TCC_HIT_sum", 0) / max(1.0, m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0))
            print(f"    {k:26s} L2 hit {hit:.3f}  miss {m.get('TCC_MISS_sum', 0):.3e}  "
                  f"tcp->tcc {m.get('TCP_TCC_READ_REQ_sum', 0):.3e}  tcp acc {m.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0):.3e}")
