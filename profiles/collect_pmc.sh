#!/bin/bash
# Hardware-counter passes over the default bench workload (BASELINE cfg2), one counter group per run as MI355X_MICROARCH.md asks:
#   bash profiles/collect_pmc.sh <tag>          (on the GPU box; results under gpurun_out/pmc_<tag>/, summary -> profiles/<tag>_pmc.json)
# The program itself follows `--` (no env / bash -c hop: the profiler's preloaded library has already initialised the GPU).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-r03}; O=$R/gpurun_out/pmc_$T; mkdir -p $O
ARGS="$R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer"
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $O/p$i -o p --output-format csv -- python3 $ARGS > $O/p$i.log 2>&1
  echo "pass $i ($C) done"
done
python3 $R/profiles/summarize_pmc.py $O $R/profiles/${T}_pmc.json
