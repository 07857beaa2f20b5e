# rocprofv3 kernel trace + PMC passes of bench.py (run on the GPU box: gpurun -- 'bash tools/prof_counters.sh TAG [bench args]')
TAG=${1:-prof}; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/$TAG
mkdir -p $O
B="python bench.py --no-cpu-baseline $@"
timeout 150 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $B --steps 20 --warmup 3 > $O/kt.log 2>&1
timeout 150 rocprofv3 --pmc SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_IFETCH -d $O/pmc1 -o pmc1 -- $B --steps 5 --warmup 1 > $O/pmc1.log 2>&1
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH -d $O/pmc2 -o pmc2 -- $B --steps 5 --warmup 1 > $O/pmc2.log 2>&1
timeout 150 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE -d $O/pmc3 -o pmc3 -- $B --steps 5 --warmup 1 > $O/pmc3.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc4 -o pmc4 -- $B --steps 5 --warmup 1 > $O/pmc4.log 2>&1
timeout 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM -d $O/pmc5 -o pmc5 -- $B --steps 5 --warmup 1 > $O/pmc5.log 2>&1
echo done
# keep only the summary (the sqlite outputs are tens of MB; gpurun merges at most 64 MiB back)
python tools/prof_summary.py $O $O/summary.json > /dev/null && find $O -name "*.db" -delete
