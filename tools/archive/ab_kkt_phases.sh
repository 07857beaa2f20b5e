# where the fused KKT backward spends its time: variants built with -DMPC_KF_SKIP=1 (no workspace stores in pass 1),
# 2 (no gradient stores in pass 2), 3 (neither), 4 (pass 1 only); bash tools/ab_kkt_phases.sh  (on the GPU box)
for L in default variants/lib_kf1.so variants/lib_kf2.so variants/lib_kf3.so variants/lib_kf4.so; do
  if [ "$L" != default ]; then export MPC_LQR_HIP_LIB=$PWD/$L; else unset MPC_LQR_HIP_LIB; fi
  echo "$L $(python tools/ab_kkt.py 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:v for k,v in d['us_per_backward'].items() if 'fused' in k})")"
done
