import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    line = [l for l in open(f) if l.startswith("{")]
    if not line:
        print(os.path.basename(f), "FAILED", open(f[:-5] + ".err").read()[-300:] if os.path.exists(f[:-5] + ".err") else ""); continue
    d = json.loads(line[0])
    ks = {k: round(v["avg_launch_ms"], 3) for k, v in d["kernels"].items()}
    c = d["check"]
    print(f"{os.path.basename(f)[:-5]:60s} {d['value']:.3e}  {ks}  box={c.get('box')} reach={c.get('reach')} tb={c.get('tiled_blocks')}")
