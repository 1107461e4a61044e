#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the rocprofv3 passes whose summaries are committed under profiles/.
#   tools/collect_profiles.sh <tag>        e.g. round5
# One workload per run, so that every row of a summary is one workload (round 2's bench CSV mixed batch sizes).
# Kernel timing and the PMC counters are separate runs (never --pmc together with other trace domains); the program
# itself follows `--` (no env / bash -c hop).
set -e
TAG=${1:-round5}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
stats() {  # stats <name> <program args...>: kernel-trace + stats of one workload -> $OUT/${TAG}_<name>.csv
    local name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1
    cp $(find $OUT/$name -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_$name.csv
}
echo "[collect] K1 alone, 100k and 500k Systems"
stats k1_100k python3 tools/k1_once.py 100000 51
stats k1_500k python3 tools/k1_once.py 500000 51
stats k1_500k_mixed python3 tools/k1_once.py 500000 51 mixed
# (the --stats average holds the cold first launch: the per-dispatch summary reports it apart)
python3 tools/k1_trace_summary.py $OUT/k1_500k "eval_rows_kernel<true" $OUT/${TAG}_k1_500k_launches.json > /dev/null
python3 tools/k1_trace_summary.py $OUT/k1_500k_mixed "eval_rows_kernel<true" $OUT/${TAG}_k1_500k_mixed_launches.json > /dev/null
echo "[collect] the headline: 100k ring16 solves only"
stats headline_kernel_stats python3 tools/solve_only.py 100000 10
echo "[collect] cfg2 resident, the reference's 64-triangle sketch x 256"
stats cfg2_kernel_stats python3 tools/cfg2_resident.py 3
stats hinged64_kernel_stats python3 tools/hinged_batch.py 64 256 5
stats hinged64_single_kernel_stats python3 tools/hinged64_once.py 1 5
stats hinged1_kernel_stats python3 tools/hinged_batch.py 1 100000 5
stats hinged16_kernel_stats python3 tools/hinged_batch.py 16 20000 3
stats qr_kernel_stats python3 tools/qr_once.py 100000 3
stats shard12500_kernel_stats python3 tools/solve_only.py 12500 10
stats hinged11_kernel_stats python3 tools/hinged_batch.py 11 100000 3
echo "[collect] HBM counters, 100k and 500k Systems"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/k1_once.py 100000 3 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/k1_once.py 100000 3 > $OUT/pmc_write.log 2>&1
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_pmc_traffic.json 100000 > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch5 -- python3 tools/k1_once.py 500000 3 > $OUT/pmc_fetch5.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write5 -- python3 tools/k1_once.py 500000 3 > $OUT/pmc_write5.log 2>&1
python3 tools/pmc_summary.py $OUT/pmc_fetch5 $OUT/pmc_write5 $OUT/${TAG}_pmc_traffic_500k.json 500000 > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch5m -- python3 tools/k1_once.py 500000 3 mixed > $OUT/pmc_fetch5m.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write5m -- python3 tools/k1_once.py 500000 3 mixed > $OUT/pmc_write5m.log 2>&1
python3 tools/pmc_summary.py $OUT/pmc_fetch5m $OUT/pmc_write5m $OUT/${TAG}_pmc_traffic_500k_mixed.json 500000 > /dev/null
echo "[collect] SQ counters of the solve kernels"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- python3 tools/solve_only.py 100000 2 > $OUT/pmc_sq.log 2>&1
python3 tools/pmc_sq_summary.py $OUT/pmc_sq $OUT/${TAG}_pmc_sq.json 100000 "tools/solve_only.py 100000 2" > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/qr_sq -- python3 tools/qr_once.py 100000 2 > $OUT/qr_sq.log 2>&1
python3 tools/pmc_sq_summary.py $OUT/qr_sq $OUT/${TAG}_qr_pmc_sq.json 100000 "tools/qr_once.py 100000 2" > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/sparse_sq -- python3 tools/hinged_batch.py 16 20000 2 > $OUT/sparse_sq.log 2>&1
python3 tools/pmc_sq_summary.py $OUT/sparse_sq $OUT/${TAG}_sparse_pmc_sq.json 20000 "tools/hinged_batch.py 16 20000 2" > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/cfg2_sq -- python3 tools/cfg2_resident.py 1 > $OUT/cfg2_sq.log 2>&1
python3 tools/pmc_sq_summary.py $OUT/cfg2_sq $OUT/${TAG}_cfg2_pmc_sq.json 1 "tools/cfg2_resident.py 1" > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/tiny_sq -- python3 tools/hinged_batch.py 1 100000 2 > $OUT/tiny_sq.log 2>&1
python3 tools/pmc_sq_summary.py $OUT/tiny_sq $OUT/${TAG}_tiny_pmc_sq.json 100000 "tools/hinged_batch.py 1 100000 2" > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/h64_sq -- python3 tools/hinged64_once.py 256 2 > $OUT/h64_sq.log 2>&1
python3 tools/pmc_sq_summary.py $OUT/h64_sq $OUT/${TAG}_hinged64_pmc_sq.json 256 "tools/hinged64_once.py 256 2" > /dev/null
echo "[collect] K1 store policy A / B"
python3 tools/k1_stores_ab.py > $OUT/k1_ab_nt.json 2> $OUT/k1_ab.err
FIKSI_AMD_K1_STORES=plain python3 tools/k1_stores_ab.py > $OUT/k1_ab_plain.json 2>> $OUT/k1_ab.err
python3 -c "import json,sys; a=json.loads(open('$OUT/k1_ab_nt.json').read().strip().splitlines()[-1]); b=json.loads(open('$OUT/k1_ab_plain.json').read().strip().splitlines()[-1]); json.dump({'source': 'tools/k1_stores_ab.py, one process per policy (FIKSI_AMD_K1_STORES)', 'non_temporal': a, 'plain': b}, open('$OUT/${TAG}_k1_stores_ab.json','w'), indent=1)"
echo "[collect] host-buffer call, batched RecursiveAssembly, the slow end of a shard, the HBM mix"
for h in 0 1 2 3; do python3 tools/host_path.py 100000 9 $h > $OUT/hp_$h.json 2>> $OUT/misc.err; done
FIKSI_AMD_HOST_CHUNKS=0 python3 tools/host_path.py 100000 9 > $OUT/hp_plain.json 2>> $OUT/misc.err
FIKSI_AMD_TRACE=1 python3 tools/host_path.py 100000 3 1 > /dev/null 2> $OUT/hp_trace.txt || true
python3 -c "import json; L=lambda f: json.loads(open(f).read().strip().splitlines()[-1]); json.dump({'source': 'tools/host_path.py 100000 9 <helped>: 0 nothing, 1 page-locked arrays (fx_host_register) + FX_HINT_ONE_STRUCTURE, 2 page-locked only, 3 the hint only; FIKSI_AMD_HOST_CHUNKS=0 for the uncut plain call; phases: FIKSI_AMD_TRACE=1 of the helped call', 'plain_two_chunks': L('$OUT/hp_0.json'), 'registered_and_hint_in_place': L('$OUT/hp_1.json'), 'registered_only_in_place': L('$OUT/hp_2.json'), 'hint_only_two_chunks': L('$OUT/hp_3.json'), 'plain_uncut': L('$OUT/hp_plain.json'), 'phases_of_the_helped_call': [l.strip() for l in open('$OUT/hp_trace.txt') if 'host call' in l][-6:]}, open('$OUT/${TAG}_host_path.json','w'), indent=1)"
python3 tools/fronts_ab.py 3 2>> $OUT/misc.err | tail -1 > $OUT/${TAG}_fronts_ab.json
python3 tools/tiny_ab.py 100000 2>> $OUT/misc.err | tail -1 > $OUT/${TAG}_tiny_ab.json
python3 tools/single_solve_latency.py > $OUT/${TAG}_single_solve_latency.txt 2>> $OUT/misc.err
python3 tools/ra_batch.py 10000 7 2>> $OUT/misc.err | tail -1 > $OUT/${TAG}_recursive_assembly_batch.json
python3 tools/straggler_probe.py 12500 2>> $OUT/misc.err | tail -1 > $OUT/${TAG}_straggler_probe.json
python3 tools/shard_times.py 100000 off:0 default 2>> $OUT/misc.err | tail -1 > $OUT/${TAG}_cfg4_shard_times.json
python3 tools/ladder_probe.py 100000 full 2>> $OUT/misc.err | tail -1 > $OUT/${TAG}_ladder_probe.json
python3 tools/grouped_c_ab.py 100000 10 2>> $OUT/misc.err | tail -1 > $OUT/${TAG}_grouped_builds_ab.json
python3 tools/grouped_s_ab.py 5 2>> $OUT/misc.err | tail -1 > $OUT/${TAG}_sparse_build_ab.json
tools/probes/rw_mix_probe.bin 2>> $OUT/misc.err | tail -1 > $OUT/${TAG}_hbm_rw_mix.json
tools/probes/valu_cost_probe.bin 2>> $OUT/misc.err | tail -1 > $OUT/${TAG}_valu_cost.json
echo "[collect] plain bench line (it quotes the counter files: this run's go to profiles/ first)"
cp $OUT/${TAG}_pmc_traffic.json $OUT/${TAG}_pmc_traffic_500k.json $OUT/${TAG}_pmc_traffic_500k_mixed.json $OUT/${TAG}_pmc_sq.json $ROOT/profiles/ 2>/dev/null || true
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
tail -c 600 $OUT/${TAG}_bench.json
