set -o pipefail
# A/B of compile-time constants of the stage-04 walker (development aid): the default build, then raster04.hip rebuilt with each flag set given on the
# command line (one argument per set, e.g. "-DORIP_CHAIN_MIN=16" "-DORIP_WALK_LEAD=8"), two bench runs each.  usage on the GPU box: bash tools/walk_batch_ab.sh SET...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-c2 --in-flight 0 > gpurun_out/b_wb.log 2>&1 || return 1; python3 -c "
import json;d=json.loads(open('gpurun_out/b_wb.log').read().strip().split(chr(10))[-1]);print('$1',d['value'],d['ms_per_step'])"; }
run default && run default
for v in "$@"; do
  (cd omnirevolve-image-processor_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $v -c raster04.hip -o raster04.o 2>/dev/null && make ARCH=gfx950 > /dev/null) || exit 1
  run "$v" && run "$v"
done
