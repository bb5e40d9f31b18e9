#!/bin/bash
# PMC passes over the headline workload, one rocprofv3 run per counter group (the hardware
# cannot collect more per pass).  Usage on the GPU box: bash scripts/profile_pmc.sh <tag> [extra args of render_once.py after the frame count]
# Results: gpurun_out/pmc_<tag>/<pass>/*counter_collection.csv
set -u
TAG=${1:-run}
shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PASSES=${PASSES:-"fetch write sq1 sq2 tcp"}
run_pass() {
    name=$1; shift
    case " $PASSES " in *" $name "*) ;; *) return 0;; esac
    timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python3 "$ROOT/scripts/render_once.py" 3 $EXTRA > "$OUT/$name.log" 2>&1
    rc=$?
    echo "pass $name rc=$rc"
    return $rc
}
EXTRA="$*"
run_pass fetch FETCH_SIZE &&
run_pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum &&
run_pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD &&
run_pass sq2 GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD &&
run_pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_READ_sum &&
run_pass ta TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE &&
run_pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TD_LOAD_WAVEFRONT_sum &&
run_pass sq3 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INSTS_VMEM
