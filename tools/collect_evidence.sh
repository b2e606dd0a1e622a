#!/bin/bash
# Evidence of a round, recorded on one gpurun box at HEAD: rocprofv3 kernel statistics per leg family, PMC passes for the two
# dominant kernels (k_fixed_msm, k_pip_chunks), the micro-benchmarks, the window sweep, a 200-step sustained run.
# usage (from the repository root, on the GPU box): bash tools/collect_evidence.sh OUTDIR [part]
#   part 1: kernel statistics per leg family      part 2: PMC passes + micro-benchmark      part 3: sweep + sustained
set -o pipefail
OUT=$(readlink -f "$1"); PART=${2:-all}
mkdir -p "$OUT"
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
OFF="--c3-steps 0 --prove-steps 0 --serialized-steps 0 --hard-steps 0 --other-curves-steps 0 --production-steps 0 --single-call-reps 0 --msm-steps 0 --latency-steps 0 --combined-steps 0 --grouped-steps 0 --pipeline-streams 0 --cpu-seconds 0"
prof() {  # name, bench args...
    local name=$1; shift
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -o k -- python3 "$ROOT/bench.py" "$@" > "$OUT/$name.json" 2> "$OUT/$name.err" || return 1
    echo "done $name"
}
if [ "$PART" = all ] || [ "$PART" = 1 ]; then
    prof headline $OFF || exit 1                      # k_fixed_msm<..., 0>: only launches at the bench geometry
    prof latency_combined $OFF --latency-steps 5 --combined-steps 10 || exit 1
    prof grouped $OFF --grouped-steps 5 || exit 1
    prof pipelined $OFF --pipeline-streams 2 || exit 1
    prof c3 --config c3 $OFF --combined-steps 5 || exit 1
    prof prove_serialized_production $OFF --prove-steps 2 --serialized-steps 3 --production-steps 3 || exit 1
    prof msm $OFF --steps 2 --warmup 1 --msm-steps 5 || exit 1
fi
if [ "$PART" = all ] || [ "$PART" = 2 ]; then
    for ctr in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
        timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/pmc_headline_$ctr" -o p -- python3 "$ROOT/bench.py" $OFF --steps 2 --warmup 1 > "$OUT/pmc_headline_$ctr.json" 2> "$OUT/pmc_headline_$ctr.err" || exit 1
        timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/pmc_msm_$ctr" -o p -- python3 "$ROOT/tools/msm_bench.py" --log2n 22 --reps 2 > "$OUT/pmc_msm_$ctr.log" 2>&1 || exit 1
        echo "done pmc $ctr"
    done
    (cd "$ROOT" && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Ibulletproofsplus_amd/csrc -o /tmp/ubench_bin tools/ubench.hip && /tmp/ubench_bin > "$OUT/ubench.json" 2> "$OUT/ubench.err") || exit 1
fi
if [ "$PART" = all ] || [ "$PART" = 3 ]; then
    timeout -k 10 600 python3 "$ROOT/tools/window_sweep.py" --out "$OUT/window_sweep.json" > "$OUT/window_sweep.log" 2>&1 || exit 1
    timeout -k 10 400 python3 "$ROOT/bench.py" $OFF --sustained-steps 200 > "$OUT/sustained.json" 2> "$OUT/sustained.err" || exit 1
    echo "done sweep + sustained"
fi
