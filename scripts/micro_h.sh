#!/bin/bash
# A/B conv3x3h (UNETRIR_CONV3X3H=1) vs conv3x3r<4,1> (=0) on the N = 64 layer shapes
for rep in 1 2; do
for v in 0 1; do
  for cfg in "64 64 256" "128 64 256" "32 32 256"; do
    UNETRIR_CONV3X3H=$v python scripts/micro_conv.py $cfg bf16 | sed "s/^/H=$v /" | cut -c1-80
  done
done
done
