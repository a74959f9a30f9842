#!/bin/bash
# Regenerates every measured artifact under profiles/ for the current build (run on the GPU box through gpurun; the raw
# rocprofv3 output directories are summarised on the box and deleted, only the small summaries travel back):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r03'
# then copy gpurun_out/<tag>_refresh/* into profiles/.
set -u
TAG=${1:-rXX}
ROOT=$(pwd)
O=$ROOT/gpurun_out/${TAG}_refresh
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
step() { echo "[refresh] $*" | tee -a "$O/refresh.log"; }

step "kernel trace, three streams (the timed configuration)"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$O/trace.log" 2>&1 || exit 1
python tools/prof_by_grid.py "$O/trace" > "$O/${TAG}_cfg1_by_grid.csv"
cp "$(find "$O/trace" -name '*kernel_stats.csv' | head -1)" "$O/${TAG}_cfg1_kernel_stats.csv"
rm -rf "$O/trace"

step "kernel trace, single stream (kernel durations with the chip to themselves)"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace1" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-overlap > "$O/trace1.log" 2>&1 || exit 1
python tools/prof_by_grid.py "$O/trace1" > "$O/${TAG}_cfg1_single_stream_by_grid.csv"
python tools/hbm_table.py "$O/${TAG}_cfg1_single_stream_by_grid.csv" > "$O/${TAG}_cfg1_hbm_kernels.csv"
rm -rf "$O/trace1"

step "PMC passes (FETCH_SIZE, WRITE_SIZE separately; no trace domains)"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- python3 bench.py --no-overlap --steps 2 --warmup 1 --no-cpu-baseline > "$O/fetch.log" 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -- python3 bench.py --no-overlap --steps 2 --warmup 1 --no-cpu-baseline > "$O/write.log" 2>&1 || exit 1
python tools/pmc_traffic.py "$O/fetch" "$O/write" "wino_kernel<2, 8, 2>" 131072 > "$O/traffic_dominant.json"
rm -rf "$O/fetch" "$O/write"

step "configs[3] / configs[4] (mixed precision): single-stream kernel traces + PMC traffic of their dominant kernel"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace3" -- python3 bench.py --config 3 --steps 4 --warmup 2 --no-cpu-baseline --no-overlap > "$O/trace3.log" 2>&1 || exit 1
python tools/prof_by_grid.py "$O/trace3" > "$O/${TAG}_cfg3_single_stream_by_grid.csv"
rm -rf "$O/trace3"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace4" -- python3 bench.py --config 4 --steps 2 --warmup 2 --no-cpu-baseline --no-overlap > "$O/trace4.log" 2>&1 || exit 1
python tools/prof_by_grid.py "$O/trace4" > "$O/${TAG}_cfg4_single_stream_by_grid.csv"
cp "$(find "$O/trace4" -name '*kernel_stats.csv' | head -1)" "$O/${TAG}_cfg4_kernel_stats.csv"
rm -rf "$O/trace4"
for CFG in 3 4; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch$CFG" -- python3 bench.py --config $CFG --no-overlap --steps 1 --warmup 1 --no-cpu-baseline > "$O/fetch$CFG.log" 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write$CFG" -- python3 bench.py --config $CFG --no-overlap --steps 1 --warmup 1 --no-cpu-baseline > "$O/write$CFG.log" 2>&1 || exit 1
  python tools/pmc_traffic.py "$O/fetch$CFG" "$O/write$CFG" "conv_bf16_kernel<3, 32, 8, true" auto > "$O/traffic_dominant_cfg$CFG.json"
  rm -rf "$O/fetch$CFG" "$O/write$CFG"
done

step "bench lines (configs[3] / [4] are better re-measured in a call of their own: right after the profiler passes they read 8-25 % slow)"
python bench.py --steps 20 --warmup 5 > "$O/${TAG}_bench_cfg1.json" 2> "$O/bench_cfg1.err" || exit 1
python bench.py --config 3 --steps 10 > "$O/${TAG}_bench_cfg3.json" 2> "$O/bench_cfg3.err" || exit 1
python bench.py --config 4 --steps 5 > "$O/${TAG}_bench_cfg4.json" 2> "$O/bench_cfg4.err" || exit 1
python bench.py --arch IRN --steps 10 > "$O/${TAG}_bench_irn.json" 2> "$O/bench_irn.err" || exit 1
step "done"
