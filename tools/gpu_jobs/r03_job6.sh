#!/bin/bash
# Round 3, GPU job 6: the 4-byte K4 in its FAST / VEC forms with quad stores: parity, stage times, counters.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job6
mkdir -p $OUT
cd $ROOT
echo "== pytest (hybrid, lengths, parity)"; timeout -k 10 900 python3 -m pytest tests/test_gpu_hybrid.py tests/test_gpu_lengths.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -4 $OUT/pytest.log
echo "== stage times u32 / f32 / i32"; for t in uint32 float32 int32; do timeout -k 10 200 python3 tools/stage_times.py $t 1 7 2>&1 | grep mode; done | tee $OUT/stages_u32.log
cd /tmp && export TMPDIR=/tmp
echo "== pmc u32"
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $ROOT/gpurun_out/prof_r03_u32b/pmc_$name -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_$name.log 2>&1 || echo "pmc $grp failed"
done
echo done
