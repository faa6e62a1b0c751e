#!/bin/bash
for cfg in 16,12 -8,12 -12,8 -8,8; do
  echo "== STEREO_VOL_CFG=$cfg"
  STEREO_VOL_CFG=$cfg timeout -k 10 200 python tools/kernel_bench.py --only A1_corr 2>&1 | tail -1 | cut -c1-110
  STEREO_VOL_CFG=$cfg timeout -k 10 200 python tools/kernel_bench.py --only A3_gwc 2>&1 | tail -1 | cut -c1-110
  STEREO_VOL_CFG=$cfg timeout -k 10 200 python tools/kernel_bench.py --only A1A4 2>&1 | tail -1 | cut -c1-110
done
