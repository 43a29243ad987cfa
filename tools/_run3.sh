set -o pipefail
O=gpurun_out/r3_p; mkdir -p $O
export MME_LIB_PATH=$PWD/multimodal_embeddings_amd/libmme_diag.so
timeout -k 10 400 python tools/bench_tilevit.py > $O/tilevit_dma.txt 2>&1; tail -2 $O/tilevit_dma.txt
MME_TILE_CINIT=1 timeout -k 10 400 python tools/bench_tilevit.py > $O/tilevit_cinit.txt 2>&1; tail -2 $O/tilevit_cinit.txt
