# round-4 profile set (run on the GPU box through gpurun):  bash tools/prof_round4.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r04p
rm -rf $O; mkdir -p $O
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
TWX_STREAMS=1 python3 bench.py --steps 5 --warmup 2 --windows 192 --no-cpu-baseline --no-caf --no-pmc > $O/bench_1slot.json 2>/dev/null
TWX_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_1slot -- python3 bench.py --steps 5 --warmup 2 --windows 192 --no-cpu-baseline --no-caf --no-pmc > $O/stats_1slot.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_3slot -- python3 bench.py --steps 5 --warmup 2 --windows 192 --no-cpu-baseline --no-caf --no-pmc > $O/stats_3slot.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --windows 8 --no-cpu-baseline --no-roofline > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 1 --warmup 0 --windows 8 --no-cpu-baseline --no-roofline > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $O/pmc_sq_a -- python3 bench.py --steps 1 --warmup 0 --windows 8 --no-cpu-baseline --no-roofline > $O/pmc_sq_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_sq_b -- python3 bench.py --steps 1 --warmup 0 --windows 8 --no-cpu-baseline --no-roofline > $O/pmc_sq_b.log 2>&1
python3 tools/aux_rates.py > $O/aux_rates.jsonl 2> $O/aux_rates.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_aux -- python3 tools/aux_rates.py > $O/stats_aux.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_caf -- python3 tools/caf_rate.py > $O/stats_caf.log 2>&1
python3 tools/caf_rate.py > $O/caf_rate.jsonl 2>/dev/null
python3 tools/aux_rates.py sliding_scan > $O/sliding_scan.jsonl 2>/dev/null
python3 tools/tracked_rate.py 180 > $O/tracked_rate.jsonl 2> $O/tracked_rate.err
python3 tools/small_n.py > $O/small_n.txt 2>&1
python3 bench.py --gpus 4 --single-process --steps 5 --warmup 2 --windows 150 > $O/bench_single_process_4ctx.json 2> $O/bench_single_process.err
python3 bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-caf > $O/bench_sustained.json 2>/dev/null
python3 tools/prof_n.py 22 3 2500000 f64 > $O/prof_f64.txt 2>&1
python3 tools/n70_rate.py > $O/n70_rate.txt 2>&1
find $O -name "*.csv" | head -30
cat $O/aux_rates.jsonl | cut -c1-250; cat $O/caf_rate.jsonl; cat $O/tracked_rate.jsonl | cut -c1-400; tail -c 700 $O/bench_default.json; tail -3 $O/prof_f64.txt $O/n70_rate.txt
