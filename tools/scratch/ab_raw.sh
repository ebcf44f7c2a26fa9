# interleaved A/B of library variants on the rawofdm workloads: usage ab_raw.sh lib1 lib2 ...
set -e
for rep in 1 2; do for L in "$@"; do
  OFDMTX_LIB=$PWD/$L timeout -k 10 300 python tools/bench_extra.py --steps 10 > gpurun_out/ab_raw.jsonl 2>/dev/null
  python - "$L" <<'PY'
import json, sys
out = []
for l in open("gpurun_out/ab_raw.jsonl"):
    d = json.loads(l)
    w = d["workload"]
    if w.startswith(("variable-size", "rawofdm")):
        out.append("%.3f" % d["kernel_ms"])
print(sys.argv[1].split("/")[-1].ljust(24), " ".join(out))
PY
done; done
