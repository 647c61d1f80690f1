#!/bin/bash
# Round 3, GPU job 16: the giants' expansion as a running maximum: parity and the float columns' stage times.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job16
mkdir -p $OUT
cd $ROOT
echo "== skew stages (default)"; RDST_STAGES=1 timeout -k 10 300 python3 tools/skew_bench.py 1000000000 "f32 normal,f32 uniform,bimodal,16-bit values,256 distinct" > $OUT/skew_default.log 2>&1; tail -12 $OUT/skew_default.log
echo "== pytest (hybrid, fullsize skew, lengths)"; timeout -k 10 900 python3 -m pytest tests/test_gpu_hybrid.py tests/test_gpu_lengths.py "tests/test_gpu_fullsize.py::test_skewed_full_size_inputs" "tests/test_gpu_fullsize.py::test_more_giants_than_tables_take_the_lsd_route" -m gpu -x -q > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -4 $OUT/pytest.log
echo "== stress with giants (kinds 8, 9, 13 come up often enough in 150 s)"; timeout -k 10 260 python3 tools/stress.py 41 150 7.3 1.5 2>&1 | tee $OUT/stress.log | tail -4
echo done
