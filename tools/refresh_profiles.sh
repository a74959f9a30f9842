#!/bin/bash
# Regenerates every measured artifact under profiles/ for the current build (run on the GPU box through gpurun; the raw
# rocprofv3 output directories are summarised on the box and deleted, only the small summaries travel back):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- "bash tools/refresh_profiles.sh r04 $(git rev-parse --short HEAD)"
# then copy gpurun_out/<tag>_refresh/* into profiles/ (roofline_kernels.json / traffic_dominant.json keep their names: bench.py
# reads them).  The commit is passed in: the GPU box has no .git.
set -u
TAG=${1:-rXX}
COMMIT=${2:-unknown}
PART=${3:-all}          # all | fp32 | bf16 | lines   (the whole refresh does not fit one 20-minute call)
ROOT=$(pwd)
O=$ROOT/gpurun_out/${TAG}_refresh
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
step() { echo "[refresh] $*" | tee -a "$O/refresh.log"; }

if [ "$PART" = all ] || [ "$PART" = fp32 ]; then
step "kernel trace, three streams (the timed configuration)"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$O/trace.log" 2>&1 || exit 1
python tools/prof_by_grid.py "$O/trace" > "$O/${TAG}_cfg1_by_grid.csv"
cp "$(find "$O/trace" -name '*kernel_stats.csv' | head -1)" "$O/${TAG}_cfg1_kernel_stats.csv"
rm -rf "$O/trace"

step "kernel trace, single stream (kernel durations with the chip to themselves)"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace1" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-overlap > "$O/trace1.log" 2>&1 || exit 1
python tools/prof_by_grid.py "$O/trace1" > "$O/${TAG}_cfg1_single_stream_by_grid.csv"
python tools/hbm_table.py "$O/${TAG}_cfg1_single_stream_by_grid.csv" > "$O/${TAG}_cfg1_hbm_kernels.csv"
(cd "$O" && python "$ROOT/tools/roofline_kernels.py" "${TAG}_cfg1_single_stream_by_grid.csv" "$COMMIT" | sed "s#\"source\": \"#\"source\": \"profiles/#" > roofline_kernels.json)
rm -rf "$O/trace1"

step "PMC passes (FETCH_SIZE, WRITE_SIZE separately; no trace domains)"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- python3 bench.py --no-overlap --steps 2 --warmup 1 --no-cpu-baseline > "$O/fetch.log" 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -- python3 bench.py --no-overlap --steps 2 --warmup 1 --no-cpu-baseline > "$O/write.log" 2>&1 || exit 1
python tools/pmc_traffic.py "$O/fetch" "$O/write" "wino_kernel<2, 8, 2>" 131072 | sed "s/\"commit\": \"unknown\"/\"commit\": \"$COMMIT\"/" > "$O/traffic_dominant.json"
python tools/pmc_traffic.py "$O/fetch" "$O/write" "wgrad_wino_group_kernel<4, 2>" auto | sed "s/\"commit\": \"unknown\"/\"commit\": \"$COMMIT\"/" > "$O/${TAG}_traffic_wgrad_group.json"
rm -rf "$O/fetch" "$O/write"

step "PMC pass: matrix-pipe utilisation per kernel (SQ_VALU_MFMA_BUSY_CYCLES against GRBM_GUI_ACTIVE)"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$O/mfma" -- python3 bench.py --no-overlap --steps 2 --warmup 1 --no-cpu-baseline > "$O/mfma.log" 2>&1 || exit 1
python tools/mfma_util.py "$O/mfma" > "$O/${TAG}_cfg1_mfma_util.csv"
rm -rf "$O/mfma"

step "-c 8 reading of configs[1] (SURVEY 8d row 2'): bench line + single-stream trace"
python bench.py --num-coupling 8 --steps 10 --warmup 3 --no-cpu-baseline > "$O/${TAG}_bench_cfg1_c8.json" 2> "$O/bench_c8.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace8" -- python3 bench.py --num-coupling 8 --steps 5 --warmup 3 --no-cpu-baseline --no-overlap > "$O/trace8.log" 2>&1 || exit 1
python tools/prof_by_grid.py "$O/trace8" > "$O/${TAG}_cfg1_c8_single_stream_by_grid.csv"
rm -rf "$O/trace8"
fi

if [ "$PART" = all ] || [ "$PART" = bf16 ]; then
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
  python tools/pmc_traffic.py "$O/fetch$CFG" "$O/write$CFG" "conv_bf16_kernel<3, 32, 8, true" auto | sed "s/\"commit\": \"unknown\"/\"commit\": \"$COMMIT\"/" > "$O/${TAG}_traffic_dominant_cfg$CFG.json"
  rm -rf "$O/fetch$CFG" "$O/write$CFG"
done
fi

if [ "$PART" = all ] || [ "$PART" = lines ]; then
step "bench lines (configs[3] / [4] are better re-measured in a call of their own: right after the profiler passes they read 8-25 % slow)"
python bench.py --steps 20 --warmup 5 > "$O/${TAG}_bench_cfg1.json" 2> "$O/bench_cfg1.err" || exit 1
python bench.py --with-tcr --steps 10 --warmup 3 --no-cpu-baseline > "$O/${TAG}_bench_cfg1_tcr.json" 2> "$O/bench_tcr.err" || exit 1
python bench.py --config 3 --steps 10 > "$O/${TAG}_bench_cfg3.json" 2> "$O/bench_cfg3.err" || exit 1
# configs[4] moves 28 GB of saved tensors per step: the first process on a fresh box reads 15 - 20 % low, the line kept is the second
python bench.py --config 4 --steps 5 --no-cpu-baseline > "$O/${TAG}_bench_cfg4_first_process.json" 2> "$O/bench_cfg4_first.err" || exit 1
python bench.py --config 4 --steps 10 > "$O/${TAG}_bench_cfg4.json" 2> "$O/bench_cfg4.err" || exit 1
python bench.py --arch IRN --steps 10 > "$O/${TAG}_bench_irn.json" 2> "$O/bench_irn.err" || exit 1
fi
step "done"
