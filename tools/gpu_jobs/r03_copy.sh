#!/bin/bash
# Round 3, GPU job 1: the streaming-rate sweep + its counter passes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_copy
mkdir -p $OUT
P=$ROOT/tools/probe/copy_sweep
for what in sweep offsets shapes; do
  echo "== $what"; timeout -k 10 300 $P $what > $OUT/$what.jsonl 2> $OUT/$what.err || echo "$what failed"
  wc -l $OUT/$what.jsonl
done
cd /tmp && export TMPDIR=/tmp
for f in 128 1024 4096; do
  i=0
  for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
             "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
             "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
             "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE" \
             "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_REQ_sum" \
             "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_WR_UNCACHED_32B_sum" \
             "GRBM_UTCL2_BUSY GRBM_EA_BUSY GRBM_TC_BUSY GRBM_COUNT" \
             "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
             "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_avr TCP_GATE_EN1_sum"; do
    i=$((i+1))
    echo "== pmc f=$f group $i: $grp"
    timeout -k 10 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_${f}_$i -- $P pmc $f > $OUT/pmc_${f}_$i.log 2>&1 || echo "pmc f=$f group $i failed"
  done
done
find $OUT -name "*counter_collection.csv" | head -40
echo done
