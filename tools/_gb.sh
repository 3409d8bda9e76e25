cd /root/repo
for args in "0 2459 24736 250 --lda 256 --ldb 256 --tail --mm" "0 2560 24832 256 --tail --mm" "0 2560 24832 512 --tail --mm" "0 2560 24832 1024 --tail --mm" "0 4096 24832 256 --tail --mm" "0 8192 24832 256 --tail --mm" "0 2048 24576 256 --tail --mm" "0 17877 2008 300 --lda 320 --ldb 320 --ldc 2048 --tail --mm"; do
python tools/gemm_bench.py $args 2>&1 | tail -1
done
