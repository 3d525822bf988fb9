#!/bin/bash
# tools/profile_config.sh OUTDIR LABEL ENV BINS SWEEPS IMPROVE   (GPU box; run from the repository root)
# One BASELINE config end to end: the PMC passes of tools/profile_counters.sh on the bench state of
# (ENV, BINS), folded into OUTDIR/counters_bench_LABEL.json, plus the rocprofv3 --kernel-trace --stats
# summary of the same command as OUTDIR/kernel_stats_bench_LABEL.csv.
set -e
O=$1; L=$2; E=$3; B=$4; S=$5; I=$6
R=$PWD
mkdir -p $O
[ -f $O/valu_issue.txt ] || cp profiles/r02/valu_issue.txt $O/valu_issue.txt
bash tools/profile_counters.sh $O bench_$L -- python3 $R/tools/eval_states.py --env $E --bins $B --state bench --sweeps $S --groups 2 --improve $I
PI_ENV=$E PI_BINS=$B PI_LAST=$S python3 tools/collect_counters.py $O bench_$L > $O/collect_$L.log
(cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats_$L -- \
    python3 $R/tools/eval_states.py --env $E --bins $B --state bench --sweeps $S --groups 2 --improve $I > $R/$O/kstats_$L.log 2>&1)
cp $O/kstats_$L/*/*_kernel_stats.csv $O/kernel_stats_bench_$L.csv
echo "profiled $L"
