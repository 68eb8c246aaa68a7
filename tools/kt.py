import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step', round(d["ms_per_step"],3), 'it', d["lm_iterations"], 'pcg', d['pcg_iterations'], 'cost', d["final_cost"])
if d.get("roofline"): print('   ', {k:(round(v["avg_us"],1), v['launches']) for k,v in d["roofline"]["per_kernel"].items()})
