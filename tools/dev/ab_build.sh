#!/bin/bash
# A/B of compile-time variants on ONE box: tools/dev/ab_build.sh <source.hip stem> "<script>" "<-Dflags A>" "<-Dflags B>" ...
# rebuilds only that object per variant and runs the timing script twice (run on the GPU box through gpurun)
PKG=deep-convolutional-neural-network-resnet-26-and-attention-network_amd
SRC=$1; SCRIPT=$2; shift 2
for v in "$@"; do
  rm -f $PKG/csrc/build/$SRC.o
  make -C $PKG/csrc -j16 EXTRA="$v" > gpurun_out/ab_build.log 2>&1 || { tail -5 gpurun_out/ab_build.log; exit 1; }
  echo "== $v"
  python $SCRIPT 2>&1 | grep -v amdgpu.ids
  python $SCRIPT 2>&1 | grep -v amdgpu.ids
done
