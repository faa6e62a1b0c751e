#!/bin/bash
# A/B the pipelined volume kernel configurations (one device, back to back).
for cfg in 16,12 16,6 16,8 12,8; do
  echo "== STEREO_VOL_CFG=$cfg"
  for k in A1_corr A3_gwc; do
  STEREO_VOL_CFG=$cfg timeout -k 10 200 python tools/kernel_bench.py --only $k 2>&1 | tail -1 | cut -c1-100
  done
done
