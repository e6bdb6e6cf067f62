#!/bin/bash
# Round-end evidence, run ON THE GPU BOX from the repo root (gpurun): the bench line, the rocprofv3 kernel-trace
# summary of the same command, and the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, --kernel-trace only).
# usage: bash tools/refresh_profiles.sh <tag>      -> gpurun_out/<tag>_*
set -e
TAG=${1:-l}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 8 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench done"
rm -rf $OUT/${TAG}_prof $OUT/${TAG}_pmc
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o r -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $OUT/${TAG}_prof.log 2>&1
echo "kernel trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc/f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity > $OUT/${TAG}_pmc_f.log 2>&1
echo "pmc fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc/w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity > $OUT/${TAG}_pmc_w.log 2>&1
echo "pmc write done"
# MFMA utilisation (north_star: "rocprof HBM GB/s and MFMA utilisation"): its own pass, --kernel-trace only beside --pmc
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_mfma -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity > $OUT/${TAG}_pmc_m.log 2>&1
echo "pmc mfma done"
python tools/pmc_mfma_summary.py $OUT/${TAG}_pmc_mfma --min-us 300 --top 10 --json $OUT/${TAG}_pmc_mfma.json > $OUT/${TAG}_pmc_mfma.txt
rm -rf $OUT/${TAG}_pmc_mfma
find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_kernel_stats.csv
# HBM bytes per launch of the three kernels with the most GPU time (the two-per-CU kernel no longer has launches that run alone) (launches that run alone: the half-batch launches of the
# two-stream backbone section carry other template arguments)
python tools/pmc_summary.py $OUT/${TAG}_pmc "conv3x3_panel_kernel<3, 0, 0, 9>" --json $OUT/${TAG}_pmc_panel.json --name conv3x3_panel_kernel --batch 32 --proposals 300 --alg-bytes 3858235392 > $OUT/${TAG}_pmc_summary.txt
python tools/pmc_summary.py $OUT/${TAG}_pmc "conv_gemm4_kernel<false, 0, 0>" --json $OUT/${TAG}_pmc_gemm4.json --name conv_gemm4_kernel --batch 32 --proposals 300 >> $OUT/${TAG}_pmc_summary.txt
python tools/pmc_summary.py $OUT/${TAG}_pmc "conv_ws_kernel" --json $OUT/${TAG}_pmc_ws.json --name conv_ws_kernel --batch 32 --proposals 300 --min-workgroups 200 >> $OUT/${TAG}_pmc_summary.txt
python - <<PY
import json
ks = [json.load(open("$OUT/${TAG}_pmc_%s.json" % k)) for k in ("panel", "gemm4", "ws")]
json.dump({"batch": 32, "proposals": 300, "kernels": ks}, open("$OUT/${TAG}_pmc_traffic.json", "w"), indent=1)
PY
# keep only the summaries (the raw traces are large)
rm -rf $OUT/${TAG}_prof $OUT/${TAG}_pmc
tail -3 $OUT/${TAG}_pmc_summary.txt
