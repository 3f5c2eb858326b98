#!/bin/bash
# Collects the round's profile artefacts on the GPU box into gpurun_out/profiles_r03/ (development aid; copy what is to be judged into profiles/).
#   bash tools/collect_profiles.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/profiles_r03; mkdir -p $O
CMD="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-c2 --in-flight 0"
# 1. kernel trace + stats of the bench command
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMD > $O/bench_under_rocprof.json 2> $O/trace.err
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/r03_bench4096_kernel_stats.csv
ORIP_TRACE_RUN=2 python tools/chain_of_queue.py $O/trace 0.25 > $O/r03_critical_queue_4096.txt 2>&1
ORIP_TRACE_RUN=2 python tools/trace_timeline.py $O/trace > $O/r03_timeline_4096.txt 2>&1
[ "$1" = "trace" ] && exit 0      # `bash tools/collect_profiles.sh trace`: the kernel trace and the two timelines only
# 2. PMC passes (separate runs, --kernel-trace only next to --pmc)
P="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c2 --in-flight 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $P > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $P > $O/pmc_write.json 2> $O/pmc_write.err
python tools/pmc_traffic.py $(ls $O/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $O/pmc_write/*/*counter_collection.csv | head -1) $O/r03_pmc_traffic.json > $O/pmc_summary.txt 2>&1
cp $O/r03_pmc_traffic.json profiles/r03_pmc_traffic.json
# 3. the full default bench line (cpu baseline, c2, pipelined leg), with the PMC file of exactly these sources in place
python3 bench.py > $O/r03_bench4096.json 2> $O/bench.err
find $O -name "*.csv" -size +20M -delete
find $O -name "*.db" -delete
ls -la $O
