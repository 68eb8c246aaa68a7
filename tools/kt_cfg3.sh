# kernel statistics of a short config-3 run (rocprofv3 --kernel-trace --stats); run on the GPU box
set -e
export TMPDIR=/tmp
O=gpurun_out/kt3
rm -rf $O && mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python bench.py --workload ${1:-cfg3} --large none --no-cpu-baseline --no-like-for-like --no-kernel-events --steps 3 --warmup 1 > $O/log.txt 2>&1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python tools/kstats.py $O/kernel_stats.csv | head -14
rm -rf $O/kt
