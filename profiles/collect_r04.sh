#!/bin/bash
# Round-4 evidence, one box, one call:   bash profiles/collect_r04.sh <tag> [pmc]     (on the GPU box; copies the summaries into profiles/)
#   1. the driver's bench line (--steps 20 --warmup 5) and the default one  -> profiles/<tag>_bench_20_5.json, <tag>_bench_default.json
#   2. rocprofv3 --kernel-trace --stats of the same command                 -> profiles/<tag>_bench_bf16_kernel_stats.csv, <tag>_bench_bf16.log,
#                                                                              <tag>_critical_path.txt (one REPLAYED step: launches per stream)
#   3. hardware-counter passes (collect_pmc.sh)                             -> profiles/r04_pmc.json  (pass `pmc` as 2nd argument)
set -e
R=$GRAFT_REPO_ROOT; T=${1:-r04a}; O=$R/gpurun_out/prof_$T; mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err
cp $O/bench_20_5.json $R/profiles/${T}_bench_20_5.json
python3 bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
cp $O/bench_default.json $R/profiles/${T}_bench_default.json
echo "bench done: $(python3 -c "import json;d=json.load(open('$O/bench_default.json'));print(d['ms_per_step'], d['value'], d['per_rank'][0]['host_busy_ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_ms'], d['roofline']['frac'])")"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o $T -- python3 $R/bench.py --steps 8 --warmup 5 --no-cpu-baseline --no-kernel-timer > $O/bench_profiled.log 2> $O/bench_profiled.err
cd $R
S=$(find $O -name "${T}_kernel_stats.csv" | head -1); K=$(find $O -name "${T}_kernel_trace.csv" | head -1)
cp $S $R/profiles/${T}_bench_bf16_kernel_stats.csv
cp $O/bench_profiled.log $R/profiles/${T}_bench_bf16.log
python3 profiles/critical_path.py $K > $R/profiles/${T}_critical_path.txt
echo "kernel stats + critical path done"
if [ "$2" = "pmc" ]; then bash profiles/collect_pmc.sh r04 > $O/pmc.log 2>&1; tail -3 $O/pmc.log; fi
mkdir -p $R/gpurun_out/profiles_$T && cp $R/profiles/${T}_* $R/gpurun_out/profiles_$T/ && ([ -f $R/profiles/r04_pmc.json ] && cp $R/profiles/r04_pmc.json $R/gpurun_out/profiles_$T/ || true)
