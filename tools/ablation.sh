set -o pipefail
# What each of the round's switches is worth at HEAD (development aid): the default bench, then the same with one test hook set at a time.
# ORIP_CUM_CHAIN, ORIP_TAIL_OLDSIM and ORIP_KMEANS_1WG select kernels of the variants build only: run `make -C omnirevolve-image-processor_amd/csrc variants`
# first and set ORIP_LIB_VARIANTS=1, or those lines repeat the default.  usage on the GPU box: [ORIP_LIB_VARIANTS=1] bash tools/ablation.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { env $1 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-c2 --in-flight 0 > gpurun_out/b_ab.log 2>&1 || return 1; python3 -c "
import json;d=json.loads(open('gpurun_out/b_ab.log').read().strip().split(chr(10))[-1]);print('%-24s %7.2f ms  %6.1f Mpx/s' % ('$1',d['ms_per_step'],d['value']))"; }
run ORIP_DEFAULT=1 && run ORIP_NO_CHAINS=1 && run ORIP_NO_PREFETCH08=1 && run ORIP_NN_NOASM=1 && run ORIP_ARC_POINTS=1 && run ORIP_CUM_CHAIN=1 && run ORIP_HASH_SORT=1 && run ORIP_CAPS_FULL=1 && run ORIP_TAIL_OLDSIM=1 && run ORIP_KMEANS_1WG=1 && run ORIP_DEFAULT=2
