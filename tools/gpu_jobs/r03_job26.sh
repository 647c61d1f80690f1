#!/bin/bash
# Round 3, GPU job 26: the split of 8-byte slices beyond the window, the lowered thresholds.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job26
mkdir -p $OUT
cd $ROOT
echo "== pytest (split, lengths)"; timeout -k 10 900 python3 -m pytest tests/test_gpu_split.py tests/test_gpu_lengths.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "rc=$rc"; tail -15 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 800 python3 tools/size_sweep.py $OUT/size_sweep.json 2>&1 | grep -v amdgpu.ids | tee $OUT/size_sweep.log
echo done
