#!/bin/bash
# A/B the volume kernel configurations (one device, back to back). negative dc = channel-pair packed kernel
for cfg in 16,12 -8,12 -12,8 -16,8; do
  echo "== STEREO_VOL_CFG=$cfg"
  for k in A1_corr A1A4; do
  STEREO_VOL_CFG=$cfg timeout -k 10 200 python tools/kernel_bench.py --only $k 2>&1 | tail -1 | cut -c1-100
  done
done
