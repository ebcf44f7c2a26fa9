# interleaved A/B of library variants on the fused rawofdm kernel (default settings):
#   usage: ab_rawtx.sh lib1 lib2 ...   (paths relative to the repo root)
set -e
for rep in 1 2 3; do for L in "$@"; do
  OFDMTX_LIB=$PWD/$L timeout -k 10 200 python tools/scratch/raw_tx_sweep.py ${SIZES:-256 64 128 512 1024} 2>/dev/null | python -c "
import sys, json
rows = [json.loads(l) for l in sys.stdin if l.startswith('{')]
print('$L'.split('/')[-1].ljust(16), ' '.join('%d/%d:%.3f%s|%.3f' % (r['N'], r['bitrate'], r['ms'], '' if r['same'] else '!!', r['modulator_ms']) for r in rows))
"
done; done
