# interleaved A/B of library variants on the rawofdm bytes -> IQ workloads: usage ab_chain.sh lib1 lib2 ...
set -e
for rep in 1 2 3; do for L in "$@"; do
  OFDMTX_LIB=$PWD/$L timeout -k 10 300 python tools/bench_extra.py --only chain --steps 20 > gpurun_out/ab_chain.jsonl 2>/dev/null
  python - "$L" <<'PY'
import json, sys
print(sys.argv[1].split("/")[-1].ljust(24), " ".join("%.3f" % json.loads(l)["kernel_ms"] for l in open("gpurun_out/ab_chain.jsonl")))
PY
done; done
