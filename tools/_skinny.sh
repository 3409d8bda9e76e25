cd $GRAFT_REPO_ROOT
for s in 1 2 4 8; do python tools/gemm_bench.py 1 4096 250 500 --splits $s; done
for s in 1 2 4; do python tools/gemm_bench.py 0 4096 500 250 --splits $s; done
for s in 2 4 8 16; do python tools/gemm_bench.py 2 500 250 4096 --splits $s; done
for s in 1 2 4 8; do python tools/gemm_bench.py 1 17877 52 2048 --ldb 320 --ldc 320 --splits $s; done
python tools/gemm_bench.py 1 17877 52 2048 --ldb 320 --ldc 320 --tail
python tools/gemm_bench.py 0 17877 2008 320 --tail --mm
python tools/gemm_bench.py 2 2048 320 17877 --splits 8 --mm
python tools/gemm_bench.py 2 2048 320 17877 --splits 4
python tools/gemm_bench.py 2 2048 320 17877 --splits 16
