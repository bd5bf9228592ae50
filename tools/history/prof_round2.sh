# round-2 profile set (run on the GPU box through gpurun):  bash tools/history/prof_round2.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r02p
rm -rf $O; mkdir -p $O
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
TWX_STREAMS=1 python3 bench.py --steps 5 --warmup 2 --windows 192 --no-cpu-baseline > $O/bench_1slot.json 2>/dev/null
TWX_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_1slot -- python3 bench.py --steps 5 --warmup 2 --windows 192 --no-cpu-baseline > $O/stats_1slot.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_3slot -- python3 bench.py --steps 5 --warmup 2 --windows 192 --no-cpu-baseline > $O/stats_3slot.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --windows 8 --no-cpu-baseline --no-roofline > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 1 --warmup 0 --windows 8 --no-cpu-baseline --no-roofline > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $O/pmc_sq_a -- python3 bench.py --steps 1 --warmup 0 --windows 8 --no-cpu-baseline --no-roofline > $O/pmc_sq_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_sq_b -- python3 bench.py --steps 1 --warmup 0 --windows 8 --no-cpu-baseline --no-roofline > $O/pmc_sq_b.log 2>&1
python3 tools/aux_rates.py > $O/aux_rates.jsonl 2> $O/aux_rates.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_aux -- python3 tools/aux_rates.py > $O/stats_aux.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_caf -- python3 tools/side_rates.py caf > $O/stats_caf.log 2>&1
tools/bin/bw_probe 2 > $O/bw_probe.txt 2>&1
tools/bin/valu_probe > $O/valu_probe.txt 2>&1
find $O -name "*.csv" | head -30
cat $O/aux_rates.jsonl; tail -2 $O/stats_caf.log; tail -c 700 $O/bench_default.json
