#!/bin/bash
# per-kernel duration table + stage times of one short bench run under rocprofv3 --kernel-trace
# usage: kdur2.sh TAG [LIB]   (extra environment is inherited: ACE_NO_STRIP=1 tools/kdur2.sh nostrip; BENCH_EXTRA="--zero-data" adds bench.py flags)
tag=$1; lib=$2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/kdur_$tag; rm -rf /tmp/p_$tag; mkdir -p gpurun_out
[ -n "$lib" ] && export ACE_SFNO_LIB=$lib ACE_LIB=$lib
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/p_$tag -o o -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --precision f16x3 $BENCH_EXTRA > /tmp/p_$tag.json 2>/tmp/p_$tag.err
python - "$tag" > $out.txt 2>&1 <<PY
import csv, collections, json, sys, glob
tag = sys.argv[1]
f = glob.glob(f"/tmp/p_{tag}/**/o_kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "ace" in n and "pack_dhconv" not in n and "split_f16" not in n:
        d[(n.replace("void ace::", "").replace("(anonymous namespace)::", "")[:64], r["Grid_Size_X"], r.get("VGPR_Count", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print("%-66s grid %8s vgpr %3s n %4d med %7.1f min %7.1f max %7.1f share %4.1f%%" % (k[0], k[1], k[2], len(v), v[len(v) // 2], v[0], v[-1], 100 * sum(v) / tot))
b = json.load(open(f"/tmp/p_{tag}.json"))
print("steps/s", b["value"], "ms", b["ms_per_step"], "roofline", b["roofline"]["frac"], "sht", b["roofline_sht"]["frac"])
for k, v in b["stages"].items():
    print("  %-28s %7.3f ms  %7.1f us/launch  %7.1f GB/s" % (k, v["ms_per_step"], v["us_per_launch"], v["gbps"]))
PY
tail -3 /tmp/p_$tag.err >> $out.txt
cp /tmp/p_$tag.json gpurun_out/bench_$tag.json 2>/dev/null
f=$(find /tmp/p_$tag -name o_kernel_stats.csv | head -1); [ -n "$f" ] && cp $f gpurun_out/kernel_stats_$tag.csv
