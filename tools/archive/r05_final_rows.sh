#!/bin/bash
# round 5: iteration probe of the bounded solve (time of each of the six launches) then the whole suite + bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_final
python tools/r05_iter_probe.py 6 2>&1 | tail -8 | cut -c1-170 | tee gpurun_out/r05_final/iter_probe.log
bash tools/r05_full.sh r05_final
