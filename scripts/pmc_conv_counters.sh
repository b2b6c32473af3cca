set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4i
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
M="$GRAFT_REPO_ROOT/scripts/micro_conv.py 128,128,128;256,256,64"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/p1 -o p --output-format csv -- python3 $M > $OUT/p1.log 2> $OUT/p1.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM -d $OUT/p2 -o p --output-format csv -- python3 $M > $OUT/p2.log 2> $OUT/p2.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_LDS_ADDR_CONFLICT -d $OUT/p3 -o p --output-format csv -- python3 $M > $OUT/p3.log 2> $OUT/p3.err
cd $GRAFT_REPO_ROOT
for p in p1 p2 p3; do python3 - $OUT/$p/p_counter_collection.csv <<'PY' > $OUT/$p.txt 2>&1
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    if "conv3x3" in k:
        print(k, {c: round(v) for c, v in d.items()})
PY
done
cat $OUT/p1.txt $OUT/p2.txt $OUT/p3.txt
