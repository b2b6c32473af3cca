#!/bin/bash
# Step time of configs[1] against the gradient-bucket size (bucket-wise Adam on the optimizer stream): bash scripts/sweep_bucket.sh
for mb in 8 16 32 64 128 512; do
  timeout -k 10 120 python bench.py --lean --no-prof --steps 30 --warmup 5 --bucket-mb $mb 2>/dev/null > /tmp/bucket_$mb.json
  python - "$mb" <<'PY'
import json, sys
d = json.loads(open(f"/tmp/bucket_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("bucket_mb", sys.argv[1], round(d["ms_per_step"], 3))
PY
done
