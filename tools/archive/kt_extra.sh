# kernel trace of tools/bench_extra.py (KKT backward, MPC.forward): bash tools/kt_extra.sh TAG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-kt_extra}
timeout 120 rocprofv3 --kernel-trace --stats -d gpurun_out/$T -o kt -- python tools/bench_extra.py > gpurun_out/$T.log 2>&1
python tools/kt_top.py gpurun_out/$T gpurun_out/${T}_top.json | head -8
find gpurun_out/$T -name "*.db" -delete
grep -A12 "^{" gpurun_out/$T.log | head -14
