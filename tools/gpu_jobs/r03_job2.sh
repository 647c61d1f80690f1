#!/bin/bash
# Round 3, GPU job 2: the fixed sweep, TLB counters of the grid-stride layout, the GPU suites, the bench line.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job2
mkdir -p $OUT
P=$ROOT/tools/probe/copy_sweep
echo "== sweep"; timeout -k 10 300 $P sweep > $OUT/sweep.jsonl 2> $OUT/sweep.err || echo "sweep failed"
wc -l $OUT/sweep.jsonl
cd /tmp && export TMPDIR=/tmp
for lay in 0 1; do
  i=0
  for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
             "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum" \
             "TCP_UTCL1_LFIFO_FULL_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum GRBM_GUI_ACTIVE" \
             "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    i=$((i+1))
    echo "== pmc layout=$lay group $i"
    timeout -k 10 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_l${lay}_$i -- $P pmc 4096 $lay > $OUT/pmc_l${lay}_$i.log 2>&1 || echo "pmc layout=$lay group $i failed"
  done
done
cd $ROOT
echo "== pytest new"; timeout -k 10 900 python3 -m pytest tests/test_gpu_lengths.py tests/test_gpu_fullsize.py -m gpu -x -q > $OUT/pytest_new.log 2>&1; echo "rc=$?"; tail -5 $OUT/pytest_new.log
echo "== bench"; timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"; tail -c 1500 $OUT/bench.err; head -c 600 $OUT/bench.json
echo done
