#!/bin/bash
# Round 3, GPU job 11: two-plane areas for 4-byte keys (18 bytes per key): parity and stage times.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job11
mkdir -p $OUT
cd $ROOT
echo "== stage times u32 / f32 / i32"; for t in uint32 float32 int32; do timeout -k 10 200 python3 tools/stage_times.py $t 1 2>&1 | grep mode; done | tee $OUT/stages_u32.log
echo "== pytest (hybrid, lengths, parity)"; timeout -k 10 900 python3 -m pytest tests/test_gpu_hybrid.py tests/test_gpu_lengths.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -4 $OUT/pytest.log
echo done
