set -o pipefail
O=gpurun_out/r3_k; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -8 $O/tests.log
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-from-host > $GRAFT_REPO_ROOT/$O/bench.json 2> $GRAFT_REPO_ROOT/$O/bench.err; echo "bench rc=$?"
cd $GRAFT_REPO_ROOT
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['forward_mfma_frac'], d['kernel_ms_per_step'])"
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -14 {} | cut -c1-150'
