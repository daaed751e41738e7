P=$PWD/deep-convolutional-neural-network-resnet-26-and-attention-network_amd
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "stem" 2>&1 | tail -12
for m in bf16 bf16x3; do python tools/dev/time_stem.py $m 2>&1 | grep stem_bwd; done
MIL_LIB_PATH=$P/libmil_hip_stamp.so python tools/dev/time_stem.py bf16 2>&1 | grep "stem_bwd" | tail -2
MIL_LIB_PATH=$P/libmil_hip_stamp.so python tools/dev/time_stem.py bf16x3 2>&1 | grep "stem_bwd_fused" | tail -1
