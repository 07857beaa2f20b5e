#!/bin/bash
# round 6, final tree (on the GPU box): the -m gpu suite, smoke(), the bench line as the driver runs it (+ the process-group path at world
# size 1), the random option sets through the C ABI -> gpurun_out/r06_final/
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_final; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; wc -c $O/bench_line.json; cp gpurun_out/bench_full.json $O/bench_full.json
MPC_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_force_dist.json 2> $O/bench_force_dist.err; echo "dist rc=$?"
TAG=r06_final/fuzz KKT=1 N1=${N1:-6000} N2=${N2:-2500} N3=${N3:-1500} K1=${K1:-3000} K2=${K2:-800} bash tools/gpu_fuzz.sh
FUZZ_GPU=1 timeout 900 python tools/emu_fuzz.py ${N4:-5000} 81 dpp16_pad > $O/fuzz/pad12.log 2>&1; echo "pad12 rc=$?"; tail -1 $O/fuzz/pad12.log
FUZZ_GPU=1 FUZZ_LONG_T=1 timeout 900 python tools/emu_fuzz.py ${N5:-1500} 82 dpp16_pad > $O/fuzz/pad12_long.log 2>&1; echo "pad12 long rc=$?"; tail -1 $O/fuzz/pad12_long.log
python tools/pad12_kkt_bench.py > $O/pad12_kkt_bench.log 2>&1; tail -8 $O/pad12_kkt_bench.log
python tools/pad40_kkt_bench.py > $O/pad40_kkt_bench.log 2>&1; tail -10 $O/pad40_kkt_bench.log
