#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the rocprofv3 passes whose summaries are committed under profiles/.
#   tools/collect_profiles.sh <tag>        e.g. round2
# Kernel timing and the PMC counters are separate runs (never --pmc together with other trace domains).
set -e
TAG=${1:-round2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
echo "[collect] bench under rocprofv3 --stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --host-threads 1 > $OUT/bench_under_rocprof.log 2>&1
cp $(find $OUT/bench -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv
echo "[collect] HBM counters, 100k and 500k Systems"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/solve_once.py 100000 2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/solve_once.py 100000 2 > $OUT/pmc_write.log 2>&1
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_pmc_traffic.json 100000 > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch5 -- python3 tools/k1_once.py 500000 3 > $OUT/pmc_fetch5.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write5 -- python3 tools/k1_once.py 500000 3 > $OUT/pmc_write5.log 2>&1
python3 tools/pmc_summary.py $OUT/pmc_fetch5 $OUT/pmc_write5 $OUT/${TAG}_pmc_traffic_500k.json 500000 > /dev/null
echo "[collect] FETCH_SIZE calibration on K1's access widths"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_calib -- tools/probes/fetch_calib.bin > $OUT/fetch_calib.log 2>&1
python3 tools/fetch_calib_summary.py $OUT/pmc_calib $OUT/${TAG}_fetch_calibration.json > /dev/null
echo "[collect] SQ counters of the solve kernels"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- python3 tools/solve_once.py 100000 2 > $OUT/pmc_sq.log 2>&1
python3 tools/pmc_sq_summary.py $OUT/pmc_sq $OUT/${TAG}_pmc_sq.json 100000 > /dev/null
echo "[collect] cfg2"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg2 -- python3 tools/cfg2.py 5000 > $OUT/cfg2.log 2>&1
cp $(find $OUT/cfg2 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_cfg2_kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/cfg2_sq -- python3 tools/cfg2.py 5000 > $OUT/cfg2_sq.log 2>&1
python3 tools/pmc_sq_summary.py $OUT/cfg2_sq $OUT/${TAG}_cfg2_pmc_sq.json 1 "tools/cfg2.py 5000" > /dev/null
echo "[collect] plain bench line"
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
tail -c 400 $OUT/${TAG}_bench.json
