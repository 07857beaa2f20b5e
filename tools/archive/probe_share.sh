#!/bin/bash
# diagnostic: how much of the headline step's time is HBM traffic?  --probe-share expands timestep 0 of C / F over the
# horizon (stride 0): that array's traffic disappears (L2-resident), the arithmetic stays
for s in "" C F CF; do
  for b in "" "--bounded"; do
  timeout 120 python bench.py --no-extra --no-cpu-baseline $b ${s:+--probe-share $s} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('share=%-3s %-9s kernel_ms %.5f' % ('$s', '$b', d['roofline']['kernel_ms']))"
  done
done
