# timeline of one config-2 solve: kernel start / end times and the gaps between consecutive kernels (rocprofv3 --kernel-trace)
set -e
export TMPDIR=/tmp
O=gpurun_out/gaps
rm -rf $O && mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/kt -o kt --output-format csv -- python bench.py --workload ${1:-cfg2} --large none --no-cpu-baseline --no-like-for-like --no-kernel-events --steps 3 --warmup 2 > $O/log.txt 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/gaps/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-60:]
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
    print(f"{name:30s} dur {(e - s) / 1e3:7.2f} us   gap {((s - prev) / 1e3 if prev else 0):6.2f} us")
    prev = e
PY
rm -rf $O/kt
