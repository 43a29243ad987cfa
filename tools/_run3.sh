set -o pipefail
O=gpurun_out/r3_j; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -x -q -k "not c5_full" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-from-host > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['forward_mfma_frac'], d['kernel_ms_per_step'])"
