#!/bin/bash
# tools/profile_bench.sh OUTDIR [bench args...] — on the GPU box: rocprofv3 kernel-trace stats and
# the HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE need separate passes: 3 + 2 of the 4 TCC
# slots) of one bench.py invocation, plus the calibration run (tools/gather_probe.py identity
# dynamics: a sweep whose compulsory traffic is known) under FETCH_SIZE.
OUT=$1; shift
R=$PWD
mkdir -p "$R/$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/kt" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline "$@" > "$R/$OUT/kt.log" 2>&1 || echo "kt failed"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$R/$OUT/pmc_$C" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$R/$OUT/pmc_$C.log" 2>&1 || echo "pmc $C failed"
done
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$R/$OUT/cal_$C" -- python3 "$R/tools/gather_probe.py" 80 identity > "$R/$OUT/cal_$C.log" 2>&1 || echo "cal $C failed"
done
cd "$R"
python3 tools/collect_traffic.py "$OUT"
