#!/bin/bash
# Cycle profile of the stage-04 walker on the GPU box (development aid): rebuilds raster04 with -DORIP_WALK_PROF, runs one bench step with
# ORIP_WALK_DBG=1 and leaves the per-layer lines in gpurun_out/walkprof.txt.  The box's copy of liborip.so is the profiling build afterwards.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}/omnirevolve-image-processor_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DORIP_WALK_PROF -c raster04.hip -o raster04.o 2>/dev/null || exit 1
make ARCH=gfx950 > /dev/null || exit 1
cd ../..
ORIP_WALK_DBG=1 timeout -k 10 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c2 --in-flight 0 > gpurun_out/walkprof.json 2> gpurun_out/walkprof.txt
grep -A1 "walk prof. layer [34]" gpurun_out/walkprof.txt | tail -6
