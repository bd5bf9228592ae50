# round-5 profile set (run on the GPU box through gpurun):  bash tools/prof_round5.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r05p
rm -rf $O; mkdir -p $O
P=/tmp/r05prof; rm -rf $P; mkdir -p $P          # raw profiler output stays on the box: only the summaries come back (64-MiB limit)
keep() { f=$(find $P/$1 -name "*$2" | head -1); [ -n "$f" ] && cp "$f" $O/$1_$2; }
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
TWX_STREAMS=1 python3 bench.py --steps 5 --warmup 2 --windows 192 --no-cpu-baseline --no-caf --no-pmc --no-wideband > $O/bench_1slot.json 2>/dev/null
TWX_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_1slot -- python3 bench.py --steps 5 --warmup 2 --windows 192 --no-cpu-baseline --no-caf --no-pmc --no-wideband > $O/stats_1slot.log 2>&1
keep stats_1slot kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_3slot -- python3 bench.py --steps 5 --warmup 2 --windows 192 --no-cpu-baseline --no-caf --no-pmc --no-wideband > $O/stats_3slot.log 2>&1
keep stats_3slot kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --windows 8 --no-cpu-baseline --no-roofline > $O/pmc_fetch.log 2>&1
keep pmc_fetch counter_collection.csv
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -- python3 bench.py --steps 1 --warmup 0 --windows 8 --no-cpu-baseline --no-roofline > $O/pmc_write.log 2>&1
keep pmc_write counter_collection.csv
# configs[4] legs: FIR + four concurrent correlations (fp32) and the fp64 chain, per-kernel durations
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_wide -- python3 bench.py --wideband-only > $O/stats_wide.log 2>&1
keep stats_wide kernel_stats.csv
python3 bench.py --wideband-only > $O/wideband.json 2>/dev/null
python3 tools/aux_rates.py > $O/aux_rates.jsonl 2> $O/aux_rates.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_aux -- python3 tools/aux_rates.py > $O/stats_aux.log 2>&1
keep stats_aux kernel_stats.csv
python3 tools/aux_rates.py sliding_scan > $O/sliding_scan.jsonl 2>/dev/null
python3 tools/caf_rate.py > $O/caf_rate.jsonl 2>/dev/null
python3 tools/tracked_rate.py 180 > $O/tracked_rate.jsonl 2> $O/tracked_rate.err
# the N > 1 lines on a one-GPU box: RCCL asked for, the exchange falls back (driver form and self-launching form), single process
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 2 --steps 5 --warmup 2 --windows 150 > $O/bench_2ranks_rccl_asked.json 2> $O/bench_2ranks.err
python3 bench.py --gpus 8 --steps 5 --warmup 2 --windows 75 --no-roofline > $O/bench_8ranks_rccl_asked.json 2> $O/bench_8ranks.err
python3 bench.py --gpus 4 --single-process --steps 5 --warmup 2 --windows 150 > $O/bench_single_process_4ctx.json 2> $O/bench_single_process.err
TWX_MULTI_INJECT=init TWX_MULTI_FORCE_RCCL=1 python3 bench.py --gpus 1 --single-process --steps 3 --warmup 1 --windows 64 > $O/bench_single_process_rccl_init_fails.json 2>/dev/null
python3 bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-caf --no-wideband > $O/bench_sustained.json 2>/dev/null
python3 tools/n70_rate.py > $O/n70_rate.txt 2>&1
ls -la $O | head -40
tail -c 1500 $O/bench_default.json; echo; tail -c 600 $O/bench_2ranks_rccl_asked.json; echo; tail -c 400 $O/bench_8ranks_rccl_asked.json
