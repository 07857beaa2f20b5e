cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf /tmp/tr_c5
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_c5 -o tr -- python tools/trace_mpc_forward.py cfg5 > /dev/null 2>&1
python tools/trace_mpc_forward.py --read /tmp/tr_c5 | cut -c1-130 | tee gpurun_out/r05_trace_cfg5.txt
