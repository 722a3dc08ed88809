#!/bin/bash
# Diagnostic rocprofv3 --pmc passes (where does a wave's time go): latencies, instruction mix, issue-cycle split.
#   tests/pmc_diag_passes.sh <tag> <workload> <frames>     (node format pinned by the caller)
tag=$1; shift
export TMPDIR=/tmp
i=0
for set in "VmemLatency" "LdsLatency" "InstrFetchLatency" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" \
           "SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_SALU SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CU_CYCLES"; do   # (a seventh set -- TCP_TCP_LATENCY / TCP_TA_TCP_STATE_READ / TCP_PENDING_STALL_CYCLES / TCP_GATE_EN1 / _EN2 -- aborts rocprofv3 7.2 on this chip)
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d gpurun_out/diag_${tag}/pass$i -- python3 tests/profile_run.py "$@" > gpurun_out/diag_${tag}_pass$i.log 2>&1 || { echo "diag pass $i ($set) failed"; tail -3 gpurun_out/diag_${tag}_pass$i.log; }
  echo "diag pass $i done"
done
python3 - "$tag" <<'PY'
import csv, glob, sys, collections, re
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"gpurun_out/diag_{tag}/pass*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])   # ("void rayca::(anonymous namespace)::k_generation<...>(...)")
        k = m.group(1) if m else r["Kernel_Name"][:30]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if not k.startswith("k_"): continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {sum(v)/len(v):18.2f}  (n={len(v)})")
PY
