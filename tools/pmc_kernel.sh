#!/bin/bash
# usage: tools/pmc_kernel.sh <kernel-name-substring> <out-prefix> -- <program> [args]   (run on the GPU box from the repo root)
# One rocprofv3 --pmc pass per counter group (never combined with the trace domains), then the per-kernel averages of the named kernel.
pat="$1"; out="$2"; shift 3
R=$PWD; export TMPDIR=/tmp
groups=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA")
i=0
for g in "${groups[@]}"; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $g --output-format csv -d $R/gpurun_out/$out/g$i -o p -- "$@" > $R/gpurun_out/$out.g$i.log 2>&1) || exit 1
  i=$((i+1))
done
python3 - "$pat" "$R/gpurun_out/$out" <<'PY'
import csv, glob, sys, collections
pat, d = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc): print("%-28s avg/launch %.4g  (%d samples)" % (k, acc[k][0] / acc[k][1], acc[k][1]))
PY
