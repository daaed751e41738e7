#!/bin/bash
# Quick look at the split-precision path: kernel stats + the two SQ counter passes (subset of tools/profile_round.sh).
#   gpurun --timeout 600 -- 'bash tools/prof_x3_quick.sh tag'
set -eo pipefail
TAG=${1:-x3}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
X3="python3 bench.py --dtype ${DT:-bf16x3} --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-extra-paths --no-traffic-pass"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --dtype ${DT:-bf16x3} --steps 10 --warmup 3 --no-cpu-baseline --no-extra-paths --no-traffic-pass --no-kernel-timer > $OUT/stats.log 2>&1
cp "$(ls $OUT/stats/*/*kernel_stats.csv | head -n 1)" $OUT/kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq1 -- $X3 > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $OUT/sq2 -- $X3 > $OUT/sq2.log 2>&1
python3 profiles/summarise_sq_counters.py $OUT/sq1 $OUT/sq2 $OUT/sq.json "$TAG: $X3" > $OUT/sq.txt
rm -rf $OUT/stats $OUT/sq1 $OUT/sq2
echo done
