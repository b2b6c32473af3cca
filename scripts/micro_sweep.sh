#!/bin/bash
# A/B of the bf16 3x3 kernels over the U-Net layer shapes.
#   G=0 R=0 -> conv3x3.hip   G=0 R=1 -> conv3x3r.hip   G=1 -> conv3x3g.hip where it applies (C % 32 == 0, N > 64)
for v in "0 0" "0 1" "1 1"; do
  set -- $v
  for cfg in "64 64 256" "128 64 256" "128 128 128" "256 128 128" "256 256 64" "512 256 64" "512 512 32" "1024 512 32"; do
    UNETRIR_CONV3X3G=$1 UNETRIR_CONV3X3R=$2 python scripts/micro_conv.py $cfg bf16 | sed "s/^/G=$1 R=$2 /"
  done
done
