# interleaved A/B of library variants on the headline bench: usage ab_bench.sh lib1 lib2 ...
set -e
for rep in 1 2 3 4 5 6 7 8; do for L in "$@"; do
  OFDMTX_LIB=$PWD/$L timeout -k 10 200 python bench.py --cpu-sample 0 --steps 30 --stage-steps 0 > gpurun_out/ab.json
  python -c "
import json,sys
d=json.load(open('gpurun_out/ab.json')); print('$L'.split('/')[-1].ljust(24), round(d['kernel_ms']['k_tx'],3), round(d['roofline']['frac'],4))"
done; done
