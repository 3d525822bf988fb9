#!/bin/bash
# tools/refresh_profiles.sh [profiles/rNN] [counters|bench|all]   (GPU box; run from the repository root)
# Regenerates the evidence that is tied to the kernel source hash (bench.py withholds every
# profile-derived figure when the committed profile is of another kernel version), for EVERY
# single-GPU BASELINE config:
#   counters_bench_<cfg>.json      PMC passes of tools/profile_counters.sh on the bench state
#   kernel_stats_bench_<cfg>.csv   rocprofv3 --kernel-trace --stats of the same command
#   counters_real_c4.json          the same counters on a policy-iteration state of C4
#   bench_<cfg>.json               the line `python bench.py [--env ... --bins ...]` prints
#   kernel_stats_benchpy_c4.csv    rocprofv3 --kernel-trace --stats of the bench command's timed region on the metric config alone
#                                  (--no-extra-configs: the other configs run the same kernel names and would mix into the averages)
# cfg = c2 (pendulum 200^2), c3 (cartpole swing-up 50^4), c4 (double pendulum 80^4: the metric config),
#       c5 (double cartpole 25^6), c5_swingup (double cartpole swing-up 25^6).
# Results land under gpurun_out/refresh/ (merged back by gpurun) AND are copied into profiles/rNN on the
# box so that the bench lines of the same call already see them; copy them into the tree afterwards:
#   cp gpurun_out/refresh/{counters_*.json,kernel_stats_*.csv,bench_*.json} profiles/rNN/
# The two halves fit one gpurun call each (~10 min): "counters" first, copy its JSONs into profiles/rNN,
# then "bench" (its lines look the counters up by env, bins and kernel hash).
set -e
R=$PWD
P=${1:-profiles/r03}
WHAT=${2:-all}
O=gpurun_out/refresh
mkdir -p $O $P
if [ "$WHAT" != "bench" ]; then
rm -rf $O; mkdir -p $O
cp $P/valu_issue.txt $O/valu_issue.txt 2>/dev/null || cp profiles/r02/valu_issue.txt $O/valu_issue.txt
bash tools/profile_config.sh $O c4 double_pendulum_swingup 80 20 3
bash tools/profile_config.sh $O c5 double_cartpole 25 10 2
bash tools/profile_config.sh $O c5_swingup double_cartpole_swingup 25 10 2
bash tools/profile_config.sh $O c3 cartpole_swingup 50 40 4
PI_MI355_GRAPHS=0 bash tools/profile_config.sh $O c2 pendulum 200 200 4     # eager launches: one dispatch record per sweep
python3 tools/eval_states.py --state real --save /tmp/real_state.pt --sweeps 5 --groups 1 > $O/real_prep.log 2>&1
bash tools/profile_counters.sh $O real_c4 -- python3 $R/tools/eval_states.py --state real --load /tmp/real_state.pt --sweeps 20 --groups 2
PI_LAST=20 python3 tools/collect_counters.py $O real_c4 > $O/collect_real_c4.log
cp $O/counters_*.json $P/                                   # bench.py reads the committed location
rm -rf $O/*_p[0-9]/ $O/kstats_c*/
fi
if [ "$WHAT" != "counters" ]; then
(cd /tmp && TMPDIR=/tmp timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats_benchpy -- \
    python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-converged-state --no-full-run --no-extra-configs > $R/$O/kstats_benchpy.log 2>&1)
cp $O/kstats_benchpy/*/*_kernel_stats.csv $O/kernel_stats_benchpy_c4.csv
python3 bench.py > $O/bench_c4.json 2> $O/bench_c4.err
python3 bench.py --env pendulum --bins 200 --steps 200 --warmup 20 > $O/bench_c2.json 2> $O/bench_c2.err
python3 bench.py --env cartpole_swingup --bins 50 --steps 50 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err
python3 bench.py --env double_cartpole --bins 25 --steps 5 --warmup 1 --no-full-run > $O/bench_c5.json 2> $O/bench_c5.err
python3 bench.py --env double_cartpole_swingup --bins 25 --steps 3 --warmup 1 --no-full-run --no-converged-state > $O/bench_c5_swingup.json 2> $O/bench_c5_swingup.err
rm -rf $O/kstats_benchpy/
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f, "value %.3e" % d["value"], "ms/step %.3f" % d["ms_per_step"], "bound", r["bound"], "frac", r["frac"], "profile", r["profile"])
    except Exception as e:
        print(f, "unreadable:", e)
PY
fi
