#!/bin/bash
for rep in 1 2; do
for v in 512 1024 2048 100000; do
  for cfg in "64 64 256" "128 64 256"; do
    UNETRIR_H_GRID=$v python scripts/micro_conv.py $cfg bf16 | sed "s/^/GRID=$v /" | cut -c1-80
  done
done
done
