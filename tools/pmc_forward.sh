#!/bin/bash
# Counter passes over the network forward kernel (tools/prof_forward.py), each in a run of its own.
# usage (on the GPU box, from the repo root): bash tools/pmc_forward.sh
set -e
OUT=gpurun_out/pmc_fwd
rm -rf $OUT && mkdir -p $OUT
export TMPDIR=/tmp
P="python3 tools/prof_forward.py"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/a -- $P > $OUT/a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/b -- $P > $OUT/b.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/c -- $P > $OUT/c.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in "abc":
    for f in glob.glob(f"gpurun_out/pmc_fwd/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "k_forward" in r["Kernel_Name"]:
                acc[r["Counter_Name"]][0] += float(r["Counter_Value"]); acc[r["Counter_Name"]][1] += 1
        for k, (v, n) in sorted(acc.items()):
            print(f"{k:32s} {v / n:16.0f}  per launch ({n} launches)")
PY
