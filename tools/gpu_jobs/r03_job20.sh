#!/bin/bash
# Round 3, GPU job 20: where a workgroup's time goes in the expanding K4 (a bell-shaped u32 column, mode 16).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job20
mkdir -p $OUT
cd $ROOT
echo "== timeline expand"; RDST_HIP_LIB=$ROOT/tools/_build/librdst_hip_exp.so timeout -k 10 200 python3 tools/timeline2.py uint32g 5 > $OUT/timeline_expand.log 2>&1; tail -12 $OUT/timeline_expand.log
echo done
