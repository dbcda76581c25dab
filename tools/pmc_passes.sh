#!/bin/bash
# Collects rocprofv3 PMC counters for the bench workload, one pass per counter group
# (gfx950 fits 8 SQ / 4 TCC / 4 TCP counters per pass; FETCH_SIZE and WRITE_SIZE need their own).
# usage: tools/pmc_passes.sh <outdir> [bench args...]      PMC_MIN=1: only the four passes tools/pmc_traffic.py reads
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for ctrs in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" \
  ; do
  i=$((i+1))
  if [ -n "${PMC_MIN:-}" ] && [ $i -ne 1 ] && [ $i -ne 2 ] && [ $i -ne 4 ] && [ $i -ne 5 ]; then continue; fi
  timeout -s KILL ${PMC_TIMEOUT:-120} rocprofv3 --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 bench.py --no-cpu-baseline --no-secondary --no-serial-roofline "$@" > "$out/pass$i.log" 2>&1
  rc=$?
  echo "pass $i ($ctrs): rc=$rc"
  if [ $rc -ne 0 ]; then tail -3 "$out/pass$i.log"; fi
done
