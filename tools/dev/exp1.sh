P=$PWD/deep-convolutional-neural-network-resnet-26-and-attention-network_amd
python tools/dev/time_stem.py bf16 2>&1 | grep stem_fwd
python tools/dev/time_stem.py bf16x3 2>&1 | grep stem_fwd
for v in p1 p3; do echo "== $v"; MIL_LIB_PATH=$P/libmil_hip_$v.so python tools/dev/time_stem.py bf16 2>&1 | grep stem_fwd;  MIL_LIB_PATH=$P/libmil_hip_$v.so python tools/dev/time_stem.py bf16x3 2>&1 | grep stem_fwd; done
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_preprocess.py -x -q -m gpu 2>&1 | tail -15
