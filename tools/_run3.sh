set -o pipefail
O=gpurun_out/r3_m; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tiles.py tests/test_gpu_pipeline.py -m gpu -x -q -k "preprocess or c3 or tiles or c1_" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
timeout -k 10 300 python tools/fuzz_parity.py k1 > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --no-from-host > $O/bench_c3.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$O/bench_c3.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['preprocess_hbm'])"
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python3 $GRAFT_REPO_ROOT/tools/bench_c3.py > $GRAFT_REPO_ROOT/$O/c3.txt 2>&1
cd $GRAFT_REPO_ROOT; cat $O/c3.txt | tail -4
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'grep -E "resize|resample" {} | cut -c1-200'
