#!/bin/bash
# PMC passes over the pair attention kernel of the guided forward (tools/att_only.py; separate passes, kernel-trace only)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/att_pmc; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  name=$(echo $grp | tr ' ' '+')
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$name -o run -- python $ROOT/tools/att_only.py > $OUT/$name.log 2>&1
done
cd $ROOT
python - <<'P'
import csv, glob, os
res = {}
for f in glob.glob("gpurun_out/att_pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "attention_kernel" in row["Kernel_Name"]:
            res.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
for k, v in sorted(res.items()): print(f"{k:28s} {sum(v)/len(v):16.0f}  ({len(v)} launches)")
P
