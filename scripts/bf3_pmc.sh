#!/bin/bash
# SQ counters of ONE bf16x3 instance and the fp32-MFMA instance it replaces (tools/bf3bench6_<name> <Cin> 0), one rocprofv3 pass per counter group
# (kernel trace only).  Usage: scripts/bf3_pmc.sh [name=l7] [Cin=256] -> gpurun_out/<ROUND>/bf3_pmc_<name>.md
N=${1:-l7}; C=${2:-256}
O=$PWD/gpurun_out/${ROUND:-r06}; mkdir -p $O
export TMPDIR=/tmp
REPO=$PWD
PASSES=(
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS"
)
PASSES_ALL=(
 "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA"
 "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD"
 "SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES"
 "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
)
cd /tmp
i=0
for g in "${PASSES[@]}"; do
  rm -rf /tmp/bf3pmc_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/bf3pmc_$i -o p -- $REPO/tools/bf3bench6_$N $C 0 > /dev/null 2> $O/bf3_pmc_$i.err
  i=$((i+1))
done
cd $REPO
python - "$N" "$O" <<'PY'
import csv, glob, sys, collections
name, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob("/tmp/bf3pmc_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("edmp::", "").replace("void ", "")
            if "conv_kernel" not in k:
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{out}/bf3_pmc_{name}.md", "w") as fo:
    for k, cs in agg.items():
        fo.write(f"## {k}\n\n| counter | mean per dispatch | dispatches |\n|---|---:|---:|\n")
        for c, v in sorted(cs.items()):
            fo.write(f"| {c} | {sum(v) / len(v):.4g} | {len(v)} |\n")
        fo.write("\n")
print(open(f"{out}/bf3_pmc_{name}.md").read())
PY
