#!/bin/bash
# Round 3, GPU job 34: the giants' counting kernel without per-key bounds in a chunk's inner waves: parity, the inputs with giants.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job34
mkdir -p $OUT
cd $ROOT
echo "== pytest (hybrid)"; timeout -k 10 900 python3 -m pytest tests/test_gpu_hybrid.py "tests/test_gpu_fullsize.py::test_skewed_full_size_inputs" -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "rc=$rc"; tail -4 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
RDST_STAGES=1 timeout -k 10 300 python3 tools/skew_bench.py 1000000000 "bimodal,f32 normal,f32 uniform,16-bit values,256 distinct" 2>&1 | grep -v amdgpu.ids | tee $OUT/skew.log
echo done
