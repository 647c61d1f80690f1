#!/bin/bash
# Round 3, GPU job 18: five heavy-digit candidates in the MSD passes: float columns' stage times, parity, uniform A/B.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job18
mkdir -p $OUT
cd $ROOT
echo "== skew stages (default)"; RDST_STAGES=1 timeout -k 10 300 python3 tools/skew_bench.py 1000000000 "f32 normal,f32 uniform,bimodal,uniform random" > $OUT/skew_default.log 2>&1; tail -10 $OUT/skew_default.log
echo "== pytest (hybrid, parity, fullsize skew)"; timeout -k 10 900 python3 -m pytest tests/test_gpu_hybrid.py tests/test_gpu_parity.py "tests/test_gpu_fullsize.py::test_skewed_full_size_inputs" -m gpu -x -q > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -4 $OUT/pytest.log
echo "== stress (floats and giants among the draws)"; timeout -k 10 260 python3 tools/stress.py 43 150 7.3 1.5 2>&1 | tee $OUT/stress.log | tail -3
echo done
