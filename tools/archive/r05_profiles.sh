# round 5: one rocprofv3 summary per call kind (tools/prof_one.py), kernel trace (avg_us + sustained_avg_us) + FETCH/WRITE + the SQ passes
# -> gpurun_out/r05_prof_<kind>/summary.json (copied to profiles/r05_prof_<kind>.json)
cd $GRAFT_REPO_ROOT
for k in "$@"; do
  PMC_SQ=1 bash tools/prof_any.sh r05_prof_$k python tools/prof_one.py $k 40 160 > gpurun_out/r05_prof_$k.log 2>&1
  tail -2 gpurun_out/r05_prof_$k.log | cut -c1-400
done
