#!/bin/bash
# tools/profile_counters.sh OUTDIR LABEL -- <python args...>   (GPU box)
# Runs the given python command under rocprofv3 once per counter group (SQ has 8 slots per pass,
# TCC 4 with FETCH_SIZE costing 3 and WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots"),
# each pass with --kernel-trace only (gpurun refuses --pmc combined with other trace domains).
# Any config: put --env/--bins in the python args (tools/eval_states.py takes them) and tell
# tools/collect_counters.py the same through PI_ENV / PI_BINS.
OUT=$1; LABEL=$2; shift 3
R=$PWD
mkdir -p "$R/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for G in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
  "SQ_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM" \
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE GRBM_GUI_ACTIVE" \
  "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
  "SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_BUSY_CU_CYCLES" ; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$R/$OUT/${LABEL}_p$i" -- "$@" > "$R/$OUT/${LABEL}_p$i.log" 2>&1 || echo "pass $i ($G) failed"
done
cd "$R"
