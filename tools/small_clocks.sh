# Debug build of the library with wall_clock64 cuts inside k_solve_small (build it first, in csrc: the Makefile's hipcc
# line with -DSMALL_CLOCKS=1 -o ../lib/libba_amd_clocks.so), one config-2 solve, the cuts of each linear solve.
# Run on the GPU box: bash tools/small_clocks.sh
set -e
O=gpurun_out/small_clocks
mkdir -p $O
BA_AMD_LIB=$PWD/visual-odometry_amd/lib/libba_amd_clocks.so timeout -k 10 120 python bench.py --workload cfg2 --large none --no-cpu-baseline --no-like-for-like --no-kernel-events --steps 1 --warmup 0 > $O/out.txt 2> $O/err.txt
grep "^k_solve_small" $O/out.txt
