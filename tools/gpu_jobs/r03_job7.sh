#!/bin/bash
# Round 3, GPU job 5: the third form of the 8-byte K4 and the final exit of the MSD passes: parity, stage times, timeline, counters.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job7
mkdir -p $OUT
cd $ROOT
echo "== pytest (hybrid, lengths, parity)"; timeout -k 10 900 python3 -m pytest tests/test_gpu_hybrid.py tests/test_gpu_lengths.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -4 $OUT/pytest.log
echo "== stage times u64"; for m in 1 15; do timeout -k 10 200 python3 tools/stage_times.py uint64 $m 2>&1 | grep mode; done | tee $OUT/stages_u64.log
echo "== stage times f64 / i64"; for t in float64 int64; do timeout -k 10 200 python3 tools/stage_times.py $t 1 15 2>&1 | grep mode; done | tee $OUT/stages_64b.log
echo "== stage times u32"; timeout -k 10 200 python3 tools/stage_times.py uint32 1 2>&1 | grep mode | tee $OUT/stages_u32.log
echo "== timeline wide3"; RDST_HIP_LIB=$ROOT/tools/_build/librdst_hip_exp.so timeout -k 10 200 python3 tools/timeline2.py uint64 1 > $OUT/timeline_wide3.log 2>&1; tail -12 $OUT/timeline_wide3.log
cd /tmp && export TMPDIR=/tmp
echo "== pmc u64"
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $ROOT/gpurun_out/prof_r03_u64c/pmc_$name -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --dtype u64 > $OUT/pmc_$name.log 2>&1 || echo "pmc $grp failed"
done
echo done
