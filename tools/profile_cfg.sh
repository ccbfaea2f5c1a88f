#!/bin/bash
# rocprofv3 kernel trace + a few PMC passes of ONE configuration of tools/stage_times.py (e.g. the open 4096^2 scene).
# Usage: tools/profile_cfg.sh <tag> <stage_times.py arguments...>   (PT_KERNEL in the environment selects the kernel)
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CMD="python3 $R/tools/stage_times.py $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH" \
           "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $SET --output-format csv -d "$OUT/pmc$i" -- $CMD > "$OUT/pmc$i.log" 2>&1 || { echo "pmc pass $i ($SET) failed"; tail -3 "$OUT/pmc$i.log"; }
    echo "pmc pass $i done: $SET" | tee -a "$OUT/progress.txt"
done
python3 "$R/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
grep "extend\|persist\|shade<false, false, true" "$OUT/summary.txt" | grep -v "true, 8\|ILb1" | cut -c1-170
