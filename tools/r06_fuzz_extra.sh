#!/bin/bash
# round 6, final tree (on the GPU box): a second fuzz campaign on other seeds than tools/r06_final.sh's -> gpurun_out/r06_fuzz_extra/
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_fuzz_extra; mkdir -p $O
FUZZ_GPU=1 timeout 900 python tools/emu_fuzz.py 8000 161 dpp16,mfma16 > $O/12_4.log 2>&1; echo "12/4 rc=$?"; tail -1 $O/12_4.log
FUZZ_GPU=1 timeout 900 python tools/emu_fuzz.py 8000 162 dpp16_pad > $O/pad12.log 2>&1; echo "pad12 rc=$?"; tail -1 $O/pad12.log
FUZZ_GPU=1 FUZZ_LONG_T=1 timeout 900 python tools/emu_fuzz.py 3000 163 dpp16,dpp16_pad,mfma16,mfma16_f64 > $O/long.log 2>&1; echo "long rc=$?"; tail -1 $O/long.log
FUZZ_GPU=1 FUZZ_LONG_T=1 timeout 900 python tools/emu_fuzz.py 4000 164 mfma40,mfma40_pad > $O/32_8.log 2>&1; echo "32/8 rc=$?"; tail -1 $O/32_8.log
FUZZ_GPU=1 timeout 600 python tools/emu_fuzz_kkt.py 4000 165 dpp16 > $O/kkt_12_4.log 2>&1; echo "kkt 12/4 rc=$?"; tail -1 $O/kkt_12_4.log
FUZZ_GPU=1 timeout 600 python tools/emu_fuzz_kkt.py 6000 166 dpp16_pad > $O/kkt_pad12.log 2>&1; echo "kkt pad12 rc=$?"; tail -1 $O/kkt_pad12.log
FUZZ_GPU=1 timeout 600 python tools/emu_fuzz_kkt.py 1000 167 mfma40 > $O/kkt_32_8.log 2>&1; echo "kkt 32/8 rc=$?"; tail -1 $O/kkt_32_8.log
FUZZ_GPU=1 timeout 600 python tools/emu_fuzz_kkt.py 3000 168 mfma40_pad > $O/kkt_pad40.log 2>&1; echo "kkt pad40 rc=$?"; tail -1 $O/kkt_pad40.log
grep -h "VIOLATION\|refused (" $O/*.log | cut -c1-300 | head -20
