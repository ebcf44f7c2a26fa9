set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/ab_tests.txt 2>&1 || { tail -20 gpurun_out/ab_tests.txt; exit 1; }
tail -1 gpurun_out/ab_tests.txt
for rep in 1 2 3; do for L in build/libofdmtx_old.so build/libofdmtx_buf0.so ofdm_amd/libofdmtx.so; do
  OFDMTX_LIB=$PWD/$L timeout -k 10 200 python bench.py --cpu-sample 0 --steps 10 > gpurun_out/ab.json
  python -c "
import json,sys
d=json.load(open('gpurun_out/ab.json')); print('$L', round(d['kernel_ms']['k_tx'],3), round(d['two_kernel_path']['k_tx_msg_ms'],3), round(d['roofline']['frac'],4))"
done; done
