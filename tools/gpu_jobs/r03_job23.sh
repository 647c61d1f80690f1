#!/bin/bash
# Round 3, GPU job 23: the expanding K4 (marks + max-scan) with the rule on mid-size buckets gone: parity, the skewed inputs, stress.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job23
mkdir -p $OUT
cd $ROOT
echo "== pytest (hybrid, lengths, parity, fullsize skew)"; timeout -k 10 1000 python3 -m pytest tests/test_gpu_hybrid.py tests/test_gpu_lengths.py tests/test_gpu_parity.py "tests/test_gpu_fullsize.py::test_skewed_full_size_inputs" -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "rc=$rc"; tail -5 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
echo "== default"; RDST_STAGES=1 timeout -k 10 400 python3 tools/skew_bench.py 1000000000 > $OUT/skew_default.log 2>&1 && cat $OUT/skew_default.log | grep -v amdgpu.ids
echo "== stress"; timeout -k 10 260 python3 tools/stress.py 47 150 7.3 1.5 2>&1 | tee $OUT/stress.log | tail -3
echo done
