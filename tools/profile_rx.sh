#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel times and two SQ counter passes over the loop-back verifier
# (tools/scratch/rx_split.py: ofdmtx_rx_demod and ofdmtx_rx_decode on the ragged and the headline batch).
#   usage: tools/profile_rx.sh <tag>
set -o pipefail
TAG=${1:-rx}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/tools/scratch/rx_split.py" > "$OUT/trace.out" 2> "$OUT/trace.err" || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d "$OUT/sq1" -- python3 "$R/tools/scratch/rx_split.py" > /dev/null 2> "$OUT/sq1.err" || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES \
    --output-format csv -d "$OUT/sq2" -- python3 "$R/tools/scratch/rx_split.py" > /dev/null 2> "$OUT/sq2.err" || exit 1
echo "written to $OUT"
