#!/bin/bash
# Run ON THE GPU BOX: SQ + TCC counter passes over tools/scratch/raw_tx_one.py.  usage: prof_raw.sh <tag> N bitrate
set -o pipefail
TAG=$1; N=$2; BR=$3
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d "$OUT/sq1" -- python3 "$R/tools/scratch/raw_tx_one.py" $N $BR > "$OUT/sq1.out" 2> "$OUT/sq1.err" || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES \
    --output-format csv -d "$OUT/sq2" -- python3 "$R/tools/scratch/raw_tx_one.py" $N $BR > "$OUT/sq2.out" 2> "$OUT/sq2.err" || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$R/tools/scratch/raw_tx_one.py" $N $BR > /dev/null 2> "$OUT/fetch.err" || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$R/tools/scratch/raw_tx_one.py" $N $BR > /dev/null 2> "$OUT/write.err" || exit 1
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/tools/scratch/raw_tx_one.py" $N $BR 10 > /dev/null 2> "$OUT/trace.err" || exit 1
python3 "$R/tools/scratch/pmc_summary.py" "$OUT/sq1" "$OUT/sq2" "$OUT/fetch" "$OUT/write" > "$OUT/summary.json"
cat "$OUT/summary.json"
