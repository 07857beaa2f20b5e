cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_stress
python tools/stress_parity.py 2048 12,4,50 3 > gpurun_out/r05_stress/12_4_f32_impl3.log 2>&1; echo "12/4 f32 impl3 rc=$?"; tail -2 gpurun_out/r05_stress/12_4_f32_impl3.log | cut -c1-250
python tools/stress_parity.py 512 12,4,50 2 f64 > gpurun_out/r05_stress/12_4_f64_impl2.log 2>&1; echo "12/4 f64 impl2 rc=$?"; tail -2 gpurun_out/r05_stress/12_4_f64_impl2.log | cut -c1-250
python tools/stress_parity.py 512 7,3,30 2 f64 > gpurun_out/r05_stress/7_3_f64_impl2.log 2>&1; echo "7/3 f64 impl2 rc=$?"; tail -2 gpurun_out/r05_stress/7_3_f64_impl2.log | cut -c1-250
python tools/stress_parity.py 512 32,8,64 5 > gpurun_out/r05_stress/32_8_f32_impl5.log 2>&1; echo "32/8 f32 impl5 rc=$?"; tail -2 gpurun_out/r05_stress/32_8_f32_impl5.log | cut -c1-250
grep -c VIOLATION gpurun_out/r05_stress/*.log
