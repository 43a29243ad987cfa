set -e
O=$GRAFT_REPO_ROOT/gpurun_out/s3; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/bench_c3.py > $O/prof.log 2>&1
cd $R
# MME_K1_* exist only in the diagnostic build (python -m multimodal_embeddings_amd.build --diag)
export MME_ALLOW_LIB_OVERRIDE=1 MME_LIB_PATH=$R/multimodal_embeddings_amd/libmme_diag.so; test -f $MME_LIB_PATH || { echo "build libmme_diag.so first"; exit 1; }
for v in 8 12 24 32; do MME_K1_VWIN=$v python3 tools/bench_c3.py 2>/dev/null | grep "C3 K1" | sed "s/^/VWIN=$v /" >> $O/sweep.log; done
for v in 12 16 32 48; do MME_K1_HBAND=$v python3 tools/bench_c3.py 2>/dev/null | grep "C3 K1" | sed "s/^/HBAND=$v /" >> $O/sweep.log; done
cat $O/sweep.log
