#!/bin/bash
# Round 3, GPU job 8: the 8-byte K4 with the list for keys equal in bits [16, 48): parity, stage times.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job8
mkdir -p $OUT
cd $ROOT
echo "== pytest (hybrid, lengths)"; timeout -k 10 900 python3 -m pytest tests/test_gpu_hybrid.py tests/test_gpu_lengths.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -4 $OUT/pytest.log
echo "== stage times u64 / f64 / i64"; for t in uint64 float64 int64; do timeout -k 10 200 python3 tools/stage_times.py $t 1 15 2>&1 | grep mode; done | tee $OUT/stages_64b.log
echo "== rehearsal: bench.py --gpus 2 on one GPU over gloo"
RDST_BENCH_SINGLE_DEVICE=1 RDST_BENCH_BACKEND=gloo timeout -k 10 600 python3 bench.py --gpus 2 --keys 100000000 --steps 3 --warmup 1 > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err; echo "rc=$?"; tail -c 400 $OUT/bench_gloo2.err; head -c 700 $OUT/bench_gloo2.json
echo done
