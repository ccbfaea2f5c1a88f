#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py.
# Usage: tools/profile_gpu.sh <tag> [extra bench args...]
# Outputs under gpurun_out/prof_<tag>/; copy the summaries you want judged into profiles/.
set -o pipefail
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $R/bench.py --steps 20 --warmup 3 --cpu-frames 0 --no-cpu-reference --no-extra --no-pmc $*"

timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
echo "trace done"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_IFETCH" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --pmc $SET --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/pmc$i.log" 2>&1 || { echo "pmc pass $i ($SET) failed"; tail -3 "$OUT/pmc$i.log"; }
    echo "pmc pass $i done: $SET" | tee -a "$OUT/progress.txt"
done
python3 "$R/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
python3 "$R/tools/pmc_traffic.py" "$OUT" "$OUT/pmc_traffic.json" > /dev/null 2>&1
cat "$OUT/summary.txt"
