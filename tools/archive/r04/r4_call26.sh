#!/bin/bash
# round 4, last call: HBM traffic counters (FETCH_SIZE; WRITE_SIZE) of the conditional-layer-norm kernel inside the ERA5-configuration network
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_csfno
for p in "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $p; name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 45 rocprofv3 --pmc "$@" --kernel-trace -f csv -d /tmp/pmc_$name -o o -- python tools/pmc_run_csfno.py > /tmp/pmc_$name.log 2>&1
  echo "pass $name rc=$?"; tail -1 /tmp/pmc_$name.log
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summarize.py "$f" > gpurun_out/pmc_csfno/$name.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/pmc_csfno/*.json")):
    d = json.load(open(f))
    for k, v in d.items():
        if "cln" in k or "pack_pformat" in k:
            print(f.split("/")[-1], k[:70], v)
PY
exit 0
