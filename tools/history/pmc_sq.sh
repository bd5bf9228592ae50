cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc2; mkdir -p gpurun_out/pmc2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc2/a -- python3 bench.py --steps 1 --warmup 0 --windows 8 --no-cpu-baseline --no-roofline > gpurun_out/pmc2/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc2/b -- python3 bench.py --steps 1 --warmup 0 --windows 8 --no-cpu-baseline --no-roofline > gpurun_out/pmc2/b.log 2>&1
ls gpurun_out/pmc2/*/*/ | head
