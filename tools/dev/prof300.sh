#!/bin/bash
# kernel stats of the live-driver size (8 bags x 200 tiles @300x300), both compute modes
set -eo pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof300
mkdir -p $OUT
for dt in bf16x3 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$dt -- python3 bench.py --size 300 --tiles 200 --dtype $dt --steps 6 --warmup 2 --no-cpu-baseline --no-extra-paths --no-traffic-pass --no-kernel-timer > $OUT/$dt.log 2>&1
  cp "$(ls $OUT/$dt/*/*kernel_stats.csv | head -n 1)" $OUT/${dt}_kernel_stats.csv
  rm -rf $OUT/$dt
  python3 bench.py --size 300 --tiles 200 --dtype $dt --steps 10 --warmup 3 --no-cpu-baseline --no-extra-paths --no-traffic-pass --no-kernel-timer > $OUT/${dt}_line.json 2>/dev/null
done
echo done
