#!/bin/bash
# Round 3, GPU job 12: linear counter tables (rotated scan) in both K4 kernels: parity, stage times, dense inputs.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job12
mkdir -p $OUT
cd $ROOT
echo "== stage times"; for t in uint32 uint64 float32; do timeout -k 10 200 python3 tools/stage_times.py $t 1 2>&1 | grep mode; done | tee $OUT/stages.log
echo "== skew stages (default)"; RDST_STAGES=1 timeout -k 10 300 python3 tools/skew_bench.py 1000000000 "reverse sorted,bimodal,ids below,byte 1 in two" > $OUT/skew_default.log 2>&1; tail -10 $OUT/skew_default.log
echo "== pytest (hybrid, lengths, parity)"; timeout -k 10 900 python3 -m pytest tests/test_gpu_hybrid.py tests/test_gpu_lengths.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -4 $OUT/pytest.log
echo done
