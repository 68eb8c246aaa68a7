# Debug build of the library with wall_clock64 cuts inside the cyclic-reduction kernels (build it first, in csrc: the Makefile's
# hipcc line with -DBCR_CLOCKS=1 -o ../lib/libba_amd_clocks.so), one short
# config-3 solve, the last lines of each kernel's cuts.  Run on the GPU box: bash tools/bcr_clocks.sh
set -e
O=gpurun_out/bcr_clocks
mkdir -p $O
BA_AMD_LIB=$PWD/visual-odometry_amd/lib/libba_amd_clocks.so timeout -k 10 120 python bench.py --workload cfg3 --large none --no-cpu-baseline --no-kernel-events --steps 1 --warmup 0 --max-iters 6 > $O/out.txt 2> $O/err.txt
for K in k_bcr_level k_bcr_tail k_bcr_backsolve; do grep "^$K" $O/out.txt | sort | uniq -c | sort -k4,4n -k1,1nr | awk '{print}' | head -40; done
