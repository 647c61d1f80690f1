#!/bin/bash
# Round 3, GPU job 3: A/B of this round's changes against the round-2 behaviour, phase timelines of the kernels furthest from
# the ceiling, rocprof records of the u64 / f32 / u32 commands, the GPU suites.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job3
mkdir -p $OUT
cd $ROOT
echo "== A/B u32"; timeout -k 10 300 python3 tools/ab_stages.py uint32 rdst_amd/librdst_hip.so tools/_build/librdst_r02like.so 2>&1 | tee $OUT/ab_u32.log
echo "== timelines"
for spec in "uint64 1" "uint64 2" "uint64 3" "uint32 2" "uint32 3"; do
  timeout -k 10 200 python3 tools/timeline2.py $spec > $OUT/timeline_$(echo $spec | tr ' ' '_').log 2>&1 || echo "timeline $spec failed"
  tail -12 $OUT/timeline_$(echo $spec | tr ' ' '_').log
done
echo "== profiles"
timeout -k 10 900 bash tools/profile.sh r03_u64 --dtype u64 > $OUT/profile_u64.log 2>&1 || echo "profile u64 failed"
echo "== pytest gpu (all)"; timeout -k 10 1200 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?"; tail -5 $OUT/pytest_gpu.log
echo done
