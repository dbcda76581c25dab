#!/bin/bash
# Second set of PMC passes: texture-addresser / L1 / TLB pressure.  usage: tools/pmc_passes2.sh <outdir> [bench args...]
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for ctrs in \
  "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_READ_sum TD_TD_BUSY_sum TD_TC_STALL_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 bench.py --no-cpu-baseline --no-secondary --no-serial-roofline "$@" > "$out/pass$i.log" 2>&1
  echo "pass $i: rc=$?"
done
