#!/bin/bash
# Produces the profile evidence of one round from ONE bench.py command (run on the GPU box, from the repo root):
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r02'
# 1. rocprofv3 --kernel-trace --stats          -> profiles/<round>_bench_kernel_stats.csv (+ the bench line of that run)
# 2. --pmc FETCH_SIZE / --pmc WRITE_SIZE        (separate passes, kernel-trace only) -> profiles/pmc_traffic.json
# 3. two SQ counter passes                      -> profiles/sq_counters.json + profiles/<round>_sq_counters.txt
# rocprofv3 gets the program itself after `--` (python3 ...), never a wrapper.
set -eo pipefail
ROUND=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$ROUND
mkdir -p $OUT profiles
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-extra-paths --no-traffic-pass"

rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-paths --no-traffic-pass --no-kernel-timer > $OUT/stats.log 2>&1
cp "$(ls $OUT/stats/*/*kernel_stats.csv | head -n 1)" profiles/${ROUND}_bench_kernel_stats.csv
echo "stats done"

rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > $OUT/fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > $OUT/write.log 2>&1
echo "write done"
python3 profiles/make_pmc_traffic.py $OUT/fetch $OUT/write profiles/pmc_traffic.json > $OUT/traffic.txt
cp profiles/pmc_traffic.json profiles/${ROUND}_pmc_traffic.json

rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq1 -- $BENCH > $OUT/sq1.log 2>&1
echo "sq1 done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $OUT/sq2 -- $BENCH > $OUT/sq2.log 2>&1
echo "sq2 done"
python3 profiles/summarise_sq_counters.py $OUT/sq1 $OUT/sq2 profiles/sq_counters.json "$ROUND: $BENCH" > profiles/${ROUND}_sq_counters.txt
cp profiles/sq_counters.json profiles/${ROUND}_sq_counters.json
# the same passes for the split-precision path (fp32 tensors, bf16x3 products): kernel stats + HBM traffic + SQ counters
X3="python3 bench.py --dtype bf16x3 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-extra-paths --no-traffic-pass"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/x3_stats -- python3 bench.py --dtype bf16x3 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-paths --no-traffic-pass --no-kernel-timer > $OUT/x3_stats.log 2>&1
cp "$(ls $OUT/x3_stats/*/*kernel_stats.csv | head -n 1)" profiles/${ROUND}_bf16x3_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/x3_fetch -- $X3 > $OUT/x3_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/x3_write -- $X3 > $OUT/x3_write.log 2>&1
python3 profiles/make_pmc_traffic.py $OUT/x3_fetch $OUT/x3_write profiles/pmc_traffic_bf16x3.json > $OUT/x3_traffic.txt
cp profiles/pmc_traffic_bf16x3.json profiles/${ROUND}_pmc_traffic_bf16x3.json
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/x3_sq1 -- $X3 > $OUT/x3_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $OUT/x3_sq2 -- $X3 > $OUT/x3_sq2.log 2>&1
python3 profiles/summarise_sq_counters.py $OUT/x3_sq1 $OUT/x3_sq2 profiles/sq_counters_bf16x3.json "$ROUND: $X3" > profiles/${ROUND}_sq_counters_bf16x3.txt
cp profiles/sq_counters_bf16x3.json profiles/${ROUND}_sq_counters_bf16x3.json
echo "bf16x3 passes done"
# the second encoder configuration (alt_resnet.py widths, 256 tiles @256x256 fwd+bwd: the bench's alt_resnet_path step)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/alt_stats -- python3 tools/prof_alt.py 4 > $OUT/alt_stats.log 2>&1
cp "$(ls $OUT/alt_stats/*/*kernel_stats.csv | head -n 1)" profiles/${ROUND}_alt_kernel_stats.csv
ALT="python3 tools/prof_alt.py 2"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/alt_sq1 -- $ALT > $OUT/alt_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $OUT/alt_sq2 -- $ALT > $OUT/alt_sq2.log 2>&1
python3 profiles/summarise_sq_counters.py $OUT/alt_sq1 $OUT/alt_sq2 profiles/sq_counters_alt.json "$ROUND: $ALT" > profiles/${ROUND}_sq_counters_alt.txt
cp profiles/sq_counters_alt.json profiles/${ROUND}_sq_counters_alt.json
echo "alt_resnet stats + SQ counters done"
# the bench lines last: their roofline.traffic / sq_counters fields read the PMC summaries written just above
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err
tail -n 1 $OUT/bench_line.json > profiles/${ROUND}_bench_line.json
echo "bench done"
# the other single-GPU configurations of BASELINE.json (512x512 tiles; one 4096-tile bag, forward only)
python3 bench.py --size 512 --tiles 128 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-paths --no-traffic-pass 2>> $OUT/bench.err | tail -n 1 > profiles/${ROUND}_bench_line_cfg3_512.json
python3 bench.py --infer --bags 1 --tiles 4096 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-paths --no-traffic-pass 2>> $OUT/bench.err | tail -n 1 > profiles/${ROUND}_bench_line_cfg5_infer4096.json
# the split-precision path at the other single-GPU configurations
python3 bench.py --dtype bf16x3 --size 512 --tiles 128 --steps 5 --warmup 2 --no-cpu-baseline --no-extra-paths --no-traffic-pass 2>> $OUT/bench.err | tail -n 1 > profiles/${ROUND}_bench_line_bf16x3_cfg3_512.json
python3 bench.py --dtype bf16x3 --infer --bags 1 --tiles 4096 --steps 5 --warmup 2 --no-cpu-baseline --no-extra-paths --no-traffic-pass 2>> $OUT/bench.err | tail -n 1 > profiles/${ROUND}_bench_line_bf16x3_cfg5_infer4096.json
echo "cfg3 / cfg5 lines done"
# the live driver's size (gbm/classify_combined.py:412: 300x300 tiles; no BASELINE config): 8 bags x 200 tiles, both compute modes
python3 bench.py --size 300 --tiles 200 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-paths --no-traffic-pass 2>> $OUT/bench.err | tail -n 1 > profiles/${ROUND}_bench_line_live300.json
python3 bench.py --dtype bf16x3 --size 300 --tiles 200 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-paths --no-traffic-pass 2>> $OUT/bench.err | tail -n 1 > profiles/${ROUND}_bench_line_bf16x3_live300.json
echo "live-driver-size lines done"

echo "all profiles written"
mkdir -p gpurun_out/profiles_out
cp profiles/${ROUND}_* profiles/pmc_traffic*.json profiles/sq_counters*.json gpurun_out/profiles_out/
