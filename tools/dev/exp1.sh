P=$PWD/deep-convolutional-neural-network-resnet-26-and-attention-network_amd
export MIL_LIB_PATH=$P/libmil_hip_ab.so
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-paths --no-kernel-timer"
for dt in bf16x3 bf16; do
for cfg in "" "--overlap" ; do
for lim in 0 128 192; do
  echo "== $dt $cfg limit=$lim"; MIL_CU_LIMIT=$lim $B --dtype $dt $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done; done
