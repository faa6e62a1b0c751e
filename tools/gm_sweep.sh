#!/bin/bash
for cfg in 1,1,32 1,1,64 1,1,1024 1,0,1024 1,1,8 1,1,16; do
  echo "== STEREO_GM_CFG=$cfg"
  STEREO_GM_CFG=$cfg timeout -k 10 200 python tools/kernel_bench.py --only A4_group_mean 2>&1 | grep "group_mean_soft_argmin cfg-2" | cut -c1-130
done
