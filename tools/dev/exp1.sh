P=$PWD/deep-convolutional-neural-network-resnet-26-and-attention-network_amd
python tools/dev/time_stem.py bf16 2>&1 | grep stem_fwd
python tools/dev/time_stem.py bf16x3 2>&1 | grep stem_fwd
for v in s32 s64 s100; do echo "== $v"; MIL_LIB_PATH=$P/libmil_hip_$v.so python tools/dev/time_stem.py bf16 2>&1 | grep stem_fwd;  MIL_LIB_PATH=$P/libmil_hip_$v.so python tools/dev/time_stem.py bf16x3 2>&1 | grep stem_fwd; done
