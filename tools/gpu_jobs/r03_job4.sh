#!/bin/bash
# Round 3, GPU job 4: which of this round's changes cost the MSD passes their 6-9 %; the side stream; the GPU suites.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job4
mkdir -p $OUT
cd $ROOT
echo "== A/B u32"; timeout -k 10 400 python3 tools/ab_stages.py uint32 rdst_amd/librdst_hip.so tools/_build/librdst_r02like.so tools/_build/librdst_slack01.so tools/_build/librdst_exit2.so tools/_build/librdst_exit2b.so 2>&1 | tee $OUT/ab_u32.log
echo "== A/B u64"; timeout -k 10 300 python3 tools/ab_stages.py uint64 rdst_amd/librdst_hip.so tools/_build/librdst_r02like.so tools/_build/librdst_exit2b.so 2>&1 | tee $OUT/ab_u64.log
echo "== pytest gpu (all)"; timeout -k 10 1200 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?"; tail -5 $OUT/pytest_gpu.log
echo "== bench"; timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"; tail -c 600 $OUT/bench.err; head -c 400 $OUT/bench.json
echo done
