#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job17
mkdir -p $OUT
cd $ROOT
for spec in "float32n 2" "float32n 3" "uint32 2"; do
  timeout -k 10 200 python3 tools/timeline2.py $spec > $OUT/timeline_$(echo $spec | tr ' ' '_').log 2>&1 || echo "timeline $spec failed"
  tail -11 $OUT/timeline_$(echo $spec | tr ' ' '_').log
done
echo done
