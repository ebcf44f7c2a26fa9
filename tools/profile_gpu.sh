#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace statistics of the default bench.py
# command plus separate --pmc passes for HBM traffic.  Results land in gpurun_out/<tag>/;
# tools/summarize_profiles.py turns them into the files committed under profiles/.
#   usage: tools/profile_gpu.sh <tag> [extra bench.py args]
set -o pipefail
TAG=${1:-prof}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
    python3 "$R/bench.py" --steps 20 --warmup 5 --cpu-sample 0 --placement-compare 0 "$@" > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err" || exit 1
for CTR in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 200 rocprofv3 --pmc $CTR --output-format csv -d "$OUT/pmc_$CTR" -- \
        python3 "$R/bench.py" --steps 2 --warmup 1 --cpu-sample 0 --placement-compare 0 "$@" > /dev/null 2> "$OUT/pmc_$CTR.err" || exit 1
done
echo "profiles written to $OUT"
