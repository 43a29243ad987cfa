set -o pipefail
O=gpurun_out/r3_b; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "pipelined" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -5 $O/tests.log
timeout -k 10 400 python tools/bench_from_host.py > $O/from_host.txt 2>&1; cat $O/from_host.txt
