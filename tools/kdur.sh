#!/bin/bash
# per-kernel duration table of one short bench run
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/p9; ACE_SFNO_LIB=$1 rocprofv3 --kernel-trace --stats -f csv -d /tmp/p9 -o o -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --precision f16x3 > /tmp/p9.json 2>/dev/null
python - <<PY
import csv, collections, json
rows=list(csv.DictReader(open("/tmp/p9/o_kernel_trace.csv")))
d=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"]
    if "ace" in n and "pack_dhconv" not in n and "split_f16" not in n:
        d[(n.replace("void ace::","").replace("_ZN3ace","")[:58], r["Grid_Size_X"], r["VGPR_Count"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(d.items()):
    v=sorted(v); print("%-60s grid %8s vgpr %3s n %4d med %7.1f min %7.1f max %7.1f"%(k[0],k[1],k[2],len(v), v[len(v)//2], v[0], v[-1]))
b=json.load(open("/tmp/p9.json")); print("steps/s", b["value"], "ms", b["ms_per_step"])
for k,v in b["stages"].items(): print("  %-28s %7.3f ms  %7.1f us/launch"%(k, v["ms_per_step"], v["us_per_launch"]))
PY
