# round 4: one rocprofv3 summary per call kind (tools/prof_one.py), kernel trace + FETCH/WRITE + the SQ passes -> profiles/r04_prof_<kind>.json
cd $GRAFT_REPO_ROOT
for k in "$@"; do
  PMC_SQ=1 bash tools/prof_any.sh r04_prof_$k python tools/prof_one.py $k 40 120 > gpurun_out/r04_prof_$k.log 2>&1
  tail -2 gpurun_out/r04_prof_$k.log | cut -c1-400
done
