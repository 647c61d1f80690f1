#!/bin/bash
# Round 3, GPU job 22: quick look at the expanding K4 (timeline + the bell-shaped column), no test suite.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job22
mkdir -p $OUT
cd $ROOT
echo "== timeline expand"; RDST_HIP_LIB=$ROOT/tools/_build/librdst_hip_exp.so timeout -k 10 200 python3 tools/timeline2.py uint32g 5 > $OUT/timeline_expand.log 2>&1 && tail -10 $OUT/timeline_expand.log
echo "== mode ${MODE:-16}"; RDST_MODE=${MODE:-16} RDST_STAGES=1 timeout -k 10 300 python3 tools/skew_bench.py 1000000000 "gaussian,f32 normal,ids below" > $OUT/skew_mode16.log 2>&1 && cat $OUT/skew_mode16.log | grep -v amdgpu.ids
echo done
