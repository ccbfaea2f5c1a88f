#!/bin/bash
# Memory-pipeline PMC passes (TA / TCP / TD) for the path-trace kernel.  Usage as profile_gpu.sh.
set -o pipefail
TAG=${1:-mem}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $R/bench.py --steps 20 --warmup 3 --cpu-frames 0 --no-cpu-reference --no-extra --no-pmc $*"
i=0
for SET in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" \
           "TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TAGRAM0_REQ_sum" \
           "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "TD_LOAD_WAVEFRONT_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --pmc $SET --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/pmc$i.log" 2>&1 || { echo "pmc pass $i ($SET) failed"; tail -3 "$OUT/pmc$i.log"; }
    echo "pmc pass $i done: $SET" | tee -a "$OUT/progress.txt"
done
python3 "$R/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
grep -v "true>" "$OUT/summary.txt"
