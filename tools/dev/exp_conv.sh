#!/bin/bash
# builds variants of the library and times the layer-1 forward conv launch (bench kernel timer)
CS="$GRAFT_REPO_ROOT/deep-convolutional-neural-network-resnet-26-and-attention-network_amd/csrc"
for v in "" "-DMIL_EXP_SAMETILE" "-DMIL_EXP_NOMFMA" "-DMIL_EXP_NOSTORE" "-DMIL_EXP_NOMFMA -DMIL_EXP_NOSTORE" "-DMIL_EXP_NOMFMA -DMIL_EXP_NOSTORE -DMIL_EXP_SAMETILE"; do
  touch $CS/conv_igemm.hip; make -C $CS -j8 EXTRA="$v" > /dev/null 2>&1
  cd "$GRAFT_REPO_ROOT"
  r=$(timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_ms']*1000,1), round(d['ms_per_step'],2))")
  echo "variant [$v]: conv24 fwd launch us, ms/step = $r"
done
