cd $GRAFT_REPO_ROOT
for n in 16384 8192 4096; do
echo "--- N=$n syrk_k64"; GPRX_K64_GEMM=0 timeout -k 10 120 python tools/large_probe.py $n 12
echo "--- N=$n K=64 through the DMA GEMM"; GPRX_K64_GEMM=1 timeout -k 10 120 python tools/large_probe.py $n 12
done
echo "--- N=16384 K64 GEMM, outer 1024"; GPRX_K64_GEMM=1 timeout -k 10 120 python tools/large_probe.py 16384 12 1024
echo "--- N=1024 x 512"; GPRX_K64_GEMM=0 python tools/batch_prof.py 1024 512 10; GPRX_K64_GEMM=1 python tools/batch_prof.py 1024 512 10
