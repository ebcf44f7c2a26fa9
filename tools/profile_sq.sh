#!/bin/bash
# Run ON THE GPU BOX (via gpurun): two SQ counter passes over a short bench.py run (instruction mix,
# LDS bank conflicts, issue activity).  Separate runs, at most 8 SQ counters each.
#   usage: tools/profile_sq.sh <tag>
set -o pipefail
TAG=${1:-sq}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d "$OUT/sq1" -- python3 "$R/bench.py" --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2> "$OUT/sq1.err" || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES \
    --output-format csv -d "$OUT/sq2" -- python3 "$R/bench.py" --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2> "$OUT/sq2.err" || exit 1
echo "SQ counters written to $OUT"
