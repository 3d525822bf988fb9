#!/bin/bash
# tools/fetch_calibration.sh OUTDIR   (GPU box; run from the repository root)
# FETCH_SIZE / WRITE_SIZE on the access shapes of the sweeps: builds tools/fetch_calibration.hip, runs it and
# tools/fetch_calibration_identity.py under rocprofv3 once per counter group (--pmc with --kernel-trace only; the program
# itself directly after `--`), then folds the CSVs into OUTDIR/fetch_calibration.{txt,json}.
set -e
O=$1
R=$PWD
mkdir -p $R/$O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/fetch_calibration.hip -o /tmp/fetch_cal
/tmp/fetch_cal > $R/$O/fetch_cal_plain.log 2>&1
python3 tools/fetch_calibration_identity.py > $R/$O/fetch_cal_identity_plain.log 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum" ; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$R/$O/micro_p$i" -- /tmp/fetch_cal > "$R/$O/micro_p$i.log" 2>&1 || echo "micro pass $i ($G) failed"
  timeout -k 10 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$R/$O/identity_p$i" -- python3 $R/tools/fetch_calibration_identity.py > "$R/$O/identity_p$i.log" 2>&1 || echo "identity pass $i ($G) failed"
done
cd $R
python3 tools/fetch_calibration_fold.py $O
