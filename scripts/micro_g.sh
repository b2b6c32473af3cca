#!/bin/bash
# A/B of conv3x3g variants (UNETRIR_G_VAR) over the N >= 128 layer shapes, interleaved
for rep in 1 2; do
for v in $@; do
  for cfg in "128 128 128" "512 512 32"; do
    UNETRIR_G_VAR=$v python scripts/micro_conv.py $cfg bf16 | sed "s/^/VAR=$v /" | cut -c1-80
  done
done
done
