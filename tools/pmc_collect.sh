#!/bin/bash
# rocprofv3 counter passes over tools/pmc_run.py (one pass per counter group; --pmc is never combined with sys/hip/hsa traces)
# usage: pmc_collect.sh TAG   -> gpurun_out/pmc_TAG.json (+ raw per-pass summaries)
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_$tag
pass() {  # name, counters...
  name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 420 rocprofv3 --pmc "$@" --kernel-trace -f csv -d /tmp/pmc_$name -o o -- python tools/pmc_run.py 2 > /tmp/pmc_$name.log 2>&1
  echo "pass $name rc=$?"; tail -2 /tmp/pmc_$name.log
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summarize.py "$f" > gpurun_out/pmc_$tag/$name.json
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM
python tools/pmc_summarize.py --merge gpurun_out/pmc_$tag > gpurun_out/pmc_$tag.json
grep -h "^lib_sha256" /tmp/pmc_fetch.log | head -1 | awk '{print $2}' > gpurun_out/pmc_$tag.sha256
grep -h "^src_sha256" /tmp/pmc_fetch.log | head -1 | awk '{print $2}' > gpurun_out/pmc_$tag.src_sha256
python tools/pmc_to_profile.py gpurun_out/pmc_$tag.json gpurun_out/pmc_${tag}_traffic.json "$(cat gpurun_out/pmc_$tag.sha256)" "$(cat gpurun_out/pmc_$tag.src_sha256)" > gpurun_out/pmc_${tag}_table.txt; cat gpurun_out/pmc_${tag}_table.txt
head -c 3000 gpurun_out/pmc_$tag.json
