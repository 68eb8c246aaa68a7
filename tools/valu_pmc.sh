# Which unit bounds the three large kernels (k_linearize_rows, k_schur_staged, k_backsub_flat)?  Two PMC passes over the
# first LM iterations of a ring scene (default: BASELINE configs[3]); run on the GPU box from the repo root.
#   bash tools/valu_pmc.sh [C L d]      ->  gpurun_out/r03v/valu_pmc.json
# Per kernel and launch: VALU / LDS / VMEM instruction counts and active cycles against the busy cycles of the launch.
set -e
export TMPDIR=/tmp
C=${1:-10000}; L=${2:-2000000}; D=${3:-5}
O=gpurun_out/r03v
rm -rf $O && mkdir -p $O
A="tools/run_scene.py --C $C --L $L --d $D --verbose 0 --opt max_iters=6"
timeout -s KILL 280 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU -d $O/p1 -o p --output-format csv -- python $A > $O/p1.log 2>&1 || { tail -5 $O/p1.log; exit 1; }
timeout -s KILL 280 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR -d $O/p2 -o p --output-format csv -- python $A > $O/p2.log 2>&1 || { tail -5 $O/p2.log; exit 1; }
python - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r03v/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].split("<")[0].split()[-1]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in acc.items():
    if not any(t in k for t in ("k_linearize", "k_schur", "k_backsub", "k_gather")): continue
    out[k] = {"launches": len(next(iter(v.values()))), **{c: sum(x) / len(x) for c, x in v.items()}}
json.dump(out, open("gpurun_out/r03v/valu_pmc.json", "w"), indent=1)
for k, v in out.items(): print(k, {c: round(x, 1) for c, x in v.items()})
PY
rm -rf $O/p1 $O/p2
