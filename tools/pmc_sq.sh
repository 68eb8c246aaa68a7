set -e
export TMPDIR=/tmp
rm -rf gpurun_out/prof_sq
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d gpurun_out/prof_sq -o sq --output-format csv -- python bench.py --steps 3 --warmup 1 --no-kernel-events > gpurun_out/prof_sq.log 2>&1 || { tail -5 gpurun_out/prof_sq.log; exit 1; }
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/prof_sq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0][-28:]
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    if not any(t in k for t in ('linearize', 'schur', 'backsub', 'gather', 'solve_small', 'k_cost')): continue
    mx = {c: max(x) for c, x in v.items()}
    print(k, {c: round(sum(a for a in x if a >= 0.05 * mx[c]) / max(1, sum(1 for a in x if a >= 0.05 * mx[c]))) for c, x in v.items()})
PY
