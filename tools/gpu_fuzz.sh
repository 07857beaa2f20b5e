#!/bin/bash
# TAG=r06_gpu_fuzz [KKT=1] [N1= N2= N3= K1= K2= K3= K4=] bash tools/gpu_fuzz.sh   (on the GPU box): logs under gpurun_out/$TAG
# tools/emu_fuzz.py's random option sets through the C ABI on the GPU (FUZZ_GPU=1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r06_gpu_fuzz}; mkdir -p $O
FUZZ_GPU=1 timeout 900 python tools/emu_fuzz.py ${N1:-1500} 61 dpp16,mfma16 > $O/12_4.log 2>&1; echo "12/4 rc=$?"; tail -1 $O/12_4.log
FUZZ_GPU=1 FUZZ_LONG_T=1 timeout 900 python tools/emu_fuzz.py ${N2:-600} 62 dpp16,mfma16,mfma16_f64 > $O/long.log 2>&1; echo "long rc=$?"; tail -1 $O/long.log
FUZZ_GPU=1 FUZZ_LONG_T=1 timeout 900 python tools/emu_fuzz.py ${N3:-400} 63 mfma40,mfma40_pad > $O/32_8.log 2>&1; echo "32/8 rc=$?"; tail -1 $O/32_8.log
grep -h "VIOLATION\|refused (" $O/*.log | cut -c1-330 | head -20
if [ -n "$KKT" ]; then
FUZZ_GPU=1 timeout 600 python tools/emu_fuzz_kkt.py ${K1:-1500} 64 dpp16 > $O/kkt_12_4.log 2>&1; echo "kkt 12/4 rc=$?"; tail -1 $O/kkt_12_4.log
FUZZ_GPU=1 timeout 600 python tools/emu_fuzz_kkt.py ${K3:-1500} 66 dpp16_pad > $O/kkt_pad12.log 2>&1; echo "kkt <=12/4 rc=$?"; tail -1 $O/kkt_pad12.log
FUZZ_GPU=1 timeout 600 python tools/emu_fuzz_kkt.py ${K2:-500} 65 mfma40 > $O/kkt_32_8.log 2>&1; echo "kkt 32/8 rc=$?"; tail -1 $O/kkt_32_8.log
FUZZ_GPU=1 timeout 600 python tools/emu_fuzz_kkt.py ${K4:-500} 67 mfma40_pad > $O/kkt_pad40.log 2>&1; echo "kkt <=32/8 rc=$?"; tail -1 $O/kkt_pad40.log
grep -h "VIOLATION" $O/kkt_*.log | cut -c1-300 | head
fi
