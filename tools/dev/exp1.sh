P=$PWD/deep-convolutional-neural-network-resnet-26-and-attention-network_amd
for m in bf16 bf16x3; do for v in "" nog; do
  if [ -z "$v" ]; then unset MIL_LIB_PATH; else export MIL_LIB_PATH=$P/libmil_hip_$v.so; fi
  echo "== $m '$v'"; python tools/dev/time_stem.py $m 2>&1 | grep "stem_bwd"
done; done
