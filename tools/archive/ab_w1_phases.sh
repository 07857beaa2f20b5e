# where the wavefront-per-problem kernel spends its time: variants/lib_w1_N.so built with -DMPC_W1_SKIP=N (bit 0: no Riccati
# recursion P2, bit 1: no rollouts P3, bit 2: no trial costs P4; results are then meaningless -- timing only)
#   bash tools/ab_w1_phases.sh     (on the GPU box)
for L in default variants/lib_w1_1.so variants/lib_w1_2.so variants/lib_w1_3.so variants/lib_w1_7.so; do
  if [ "$L" != default ]; then export MPC_LQR_HIP_LIB=$PWD/$L; else unset MPC_LQR_HIP_LIB; fi
  echo "$L $(python tools/tiny_probe2.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:v['real_impl6'] for k,v in d.items() if 'it5' in k})")"
done
