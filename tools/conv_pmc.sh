#!/bin/bash
# PMC counters of the 1x1 implicit-GEMM kernel on one layer shape (one rocprofv3 pass per counter group):
#   tools/conv_pmc.sh <tag> <Cin> <Cout> <HW side> [ks=1] [what=fwd|wgrad] [kernel-name substring=conv1x1_tr]
set -e
TAG=$1; CIN=$2; COUT=$3; SIDE=$4; KS=${5:-1}; WHAT=${6:-fwd}; PAT=${7:-conv1x1_tr}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cat > /tmp/conv_one.py <<PY
import sys, torch
sys.path.insert(0, "$REPO")
from custom_d_fine_amd import hip as H
dev = torch.device("cuda", 0)
x = torch.randn(32, $CIN, $SIDE, $SIDE, device=dev).to(torch.bfloat16)
w = torch.randn($COUT, $CIN, $KS, $KS, device=dev)
w2 = H.conv_pack_weights(w, False)
dy = torch.randn(32, $COUT, $SIDE, $SIDE, device=dev).to(torch.bfloat16)
for _ in range(5):
    if "$WHAT" == "wgrad": H.conv_wgrad_bf16(x, dy, $KS)
    else: y = H.conv_forward_bf16(x, w2, $COUT, $KS)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
for GRP in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $GRP | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/cp_$N
  rocprofv3 --pmc $GRP --kernel-trace --output-format csv -d /tmp/cp_$N -o c -- python /tmp/conv_one.py > "$OUT/log_$N.txt" 2>&1 || true
    python3 - /tmp/cp_$N "$PAT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, s) in agg.items():
    print(f"{k:34s} {s/n:16.1f} per dispatch ({n} dispatches)")
PY
done
