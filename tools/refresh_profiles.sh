#!/bin/bash
# tools/refresh_profiles.sh [profiles/rNN]   (GPU box; run from the repository root)
# Regenerates the evidence that is tied to the kernel source hash (bench.py withholds its roofline
# fraction, and tests/test_hygiene.py fails, when the committed profile is of another kernel version):
#   counters_bench_c4.json, counters_real_c4.json   six --pmc passes each (tools/profile_counters.sh)
#   kernel_stats_bench_c4.csv                       rocprofv3 --kernel-trace --stats of the bench command
#   bench_c4.json                                   the line `python bench.py` prints
# Results land under gpurun_out/refresh/ (merged back by gpurun); copy them into profiles/rNN afterwards:
#   cp gpurun_out/refresh/{counters_bench_c4.json,counters_real_c4.json,kernel_stats_bench_c4.csv,bench_c4.json} profiles/rNN/
set -e
R=$PWD
P=${1:-profiles/r02}
O=gpurun_out/refresh
rm -rf $O; mkdir -p $O
cp $P/valu_issue.txt $O/valu_issue.txt                      # collect_counters reads the measured VALU peak from it
bash tools/profile_counters.sh $O bench_c4 -- python3 $R/tools/eval_states.py --state bench --sweeps 20 --groups 2 --improve 3
python3 tools/eval_states.py --state real --save /tmp/real_state.pt --sweeps 5 --groups 1 > $O/real_prep.log 2>&1
bash tools/profile_counters.sh $O real_c4 -- python3 $R/tools/eval_states.py --state real --load /tmp/real_state.pt --sweeps 20 --groups 2
PI_LAST=20 python3 tools/collect_counters.py $O bench_c4 real_c4 > $O/collect.log
cp $O/counters_bench_c4.json $O/counters_real_c4.json $P/       # bench.py reads the committed location
(cd /tmp && TMPDIR=/tmp timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats -- \
    python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-converged-state --no-full-run > $R/$O/kstats.log 2>&1)
cp $O/kstats/*/*_kernel_stats.csv $O/kernel_stats_bench_c4.csv
python3 bench.py > $O/bench_c4.json 2> $O/bench.err
python3 - <<PY
import json
d = json.load(open("$O/bench_c4.json"))
r = d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", r["frac"], "hash", r["kernel_source_hash"], "traffic", r["traffic"])
PY
