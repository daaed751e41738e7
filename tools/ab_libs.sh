#!/bin/bash
# A/B of two builds of the library on the same bench command, per kernel (rocprofv3 kernel stats):
#   gpurun --timeout 900 -- 'bash tools/ab_libs.sh <variant> [bench args...]'
# writes gpurun_out/ab_<variant>/{default,<variant>}_kernel_stats.csv and the two bench lines.
set -eo pipefail
VAR=$1; shift
export TMPDIR=/tmp
PKG=$PWD/deep-convolutional-neural-network-resnet-26-and-attention-network_amd
OUT=gpurun_out/ab_$VAR
mkdir -p $OUT
for which in default $VAR; do
  if [ $which = default ]; then unset MIL_LIB_PATH; else export MIL_LIB_PATH=$PKG/libmil_hip_$VAR.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$which -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-paths --no-traffic-pass --no-kernel-timer "$@" > $OUT/$which.json 2> $OUT/$which.err
  cp "$(ls $OUT/$which/*/*kernel_stats.csv | head -n 1)" $OUT/${which}_kernel_stats.csv
  rm -rf $OUT/$which
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-paths --no-traffic-pass --no-kernel-timer "$@" > $OUT/${which}_line.json 2>> $OUT/$which.err
done
echo done
