cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof1
rocprofv3 -L > $R/gpurun_out/prof1/counters.txt 2>&1
cd $R
rocprofv3 --kernel-trace --stats -d gpurun_out/prof1/kt -o kt -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof1/kt.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU -d gpurun_out/prof1/pmc1 -o pmc1 -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof1/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU -d gpurun_out/prof1/pmc2 -o pmc2 -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof1/pmc2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE -d gpurun_out/prof1/pmc3 -o pmc3 -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof1/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d gpurun_out/prof1/pmc4 -o pmc4 -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof1/pmc4.log 2>&1
find gpurun_out/prof1 -name "*.csv" | head -30
