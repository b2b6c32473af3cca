#!/bin/bash
# Round profile set: rocprofv3 kernel stats of the single-stream and the side-stream schedule + three PMC passes (FETCH_SIZE,
# WRITE_SIZE, MFMA busy / clock), each its own run.  Usage (on the GPU box, from the repo root): bash scripts/profile_round.sh r02
set -u
TAG=${1:-rXX}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py --lean --no-prof --no-cpu-baseline --dtype bf16"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/stats_ss -o s --output-format csv -- python3 $B --steps 10 --warmup 2 --no-overlap > $OUT/stats_ss.json 2> $OUT/stats_ss.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/stats_ov -o s --output-format csv -- python3 $B --steps 10 --warmup 2 > $OUT/stats_ov.json 2> $OUT/stats_ov.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p --output-format csv -- python3 $B --steps 2 --warmup 1 --no-overlap > /dev/null 2> $OUT/pmc_fetch.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o p --output-format csv -- python3 $B --steps 2 --warmup 1 --no-overlap > /dev/null 2> $OUT/pmc_write.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $OUT/pmc_mfma -o p --output-format csv -- python3 $B --steps 2 --warmup 1 --no-overlap > /dev/null 2> $OUT/pmc_mfma.err
# fp32 storage mode (the fp32-tolerance parity mode; north_star's ">= 40 % of the fp32 MFMA roofline"): kernel stats + MFMA-busy pass
B32="$GRAFT_REPO_ROOT/bench.py --lean --no-prof --no-cpu-baseline --dtype f32"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_f32 -o s --output-format csv -- python3 $B32 --steps 6 --warmup 2 --no-overlap > $OUT/stats_f32.json 2> $OUT/stats_f32.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $OUT/pmc_mfma_f32 -o p --output-format csv -- python3 $B32 --steps 2 --warmup 1 --no-overlap > /dev/null 2> $OUT/pmc_mfma_f32.err
# BASELINE.json configs[4] (ResAE) and the reference's own geometry: kernel stats of the single-stream step
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/stats_resae -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prof_resae.py bf16 12 > $OUT/stats_resae.log 2> $OUT/stats_resae.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/stats_refgeom -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prof_refgeom.py 32 12 > $OUT/stats_refgeom.log 2> $OUT/stats_refgeom.err
cd $GRAFT_REPO_ROOT
python3 scripts/stats_summary.py $OUT/stats_f32/s_kernel_stats.csv > $OUT/${TAG}_f32_kernel_stats.csv
python3 scripts/pmc_mfma.py $OUT/pmc_mfma_f32/p_counter_collection.csv > $OUT/${TAG}_f32_pmc_mfma_busy.txt
python3 scripts/stats_summary.py $OUT/stats_resae/s_kernel_stats.csv > $OUT/${TAG}_resae_bf16_kernel_stats.csv
python3 scripts/stats_summary.py $OUT/stats_refgeom/s_kernel_stats.csv > $OUT/${TAG}_refgeom_bf16_kernel_stats.csv
rm -rf $OUT/stats_f32 $OUT/pmc_mfma_f32 $OUT/stats_resae $OUT/stats_refgeom
for d in stats_ss stats_ov; do python3 scripts/stats_summary.py $OUT/$d/s_kernel_stats.csv > $OUT/${TAG}_bf16_kernel_stats_${d#stats_}.csv; done
python3 scripts/pmc_summary.py $OUT/pmc_fetch/p_counter_collection.csv > $OUT/${TAG}_bf16_pmc_fetch_size.txt
python3 scripts/pmc_summary.py $OUT/pmc_write/p_counter_collection.csv > $OUT/${TAG}_bf16_pmc_write_size.txt
python3 scripts/pmc_mfma.py $OUT/pmc_mfma/p_counter_collection.csv > $OUT/${TAG}_bf16_pmc_mfma_busy.txt
python3 scripts/pmc_traffic.py $OUT/${TAG}_bf16_pmc_fetch_size.txt $OUT/${TAG}_bf16_pmc_write_size.txt conv3x3p_bf16_kernel $TAG > $OUT/pmc_traffic.json
rm -rf $OUT/stats_ss $OUT/stats_ov $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma
ls -la $OUT
