# round 5, final tree: the remaining randomised parity runs (fused 32/8 KKT backward, the row-per-problem kernel, the network kernels, the padded 32/8 shapes)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_stress; mkdir -p $O
timeout 900 python tools/stress_kkt40.py 40 > $O/kkt40.log 2>&1; echo "kkt40 rc=$?"; tail -1 $O/kkt40.log | cut -c1-200
timeout 900 python tools/stress_wave1.py > $O/wave1.log 2>&1; echo "wave1 rc=$?"; tail -1 $O/wave1.log | cut -c1-200
timeout 900 python tools/nn_stress.py 60 > $O/nn.log 2>&1; echo "nn rc=$?"; tail -1 $O/nn.log | cut -c1-200
for s in 13,4,50 20,5,40 24,8,30 8,6,25; do timeout 600 python tools/stress_parity.py 256 $s 7 > $O/pad_${s//,/_}.log 2>&1; echo "pad $s rc=$?"; tail -1 $O/pad_${s//,/_}.log | cut -c1-160; done
grep -c VIOLATION $O/*.log
