#!/bin/bash
# exact fp32 vs split-bf16 forward vs split-bf16 forward+backward, 8 and 32 clips per step (python bench.py --no-roofline lines)
for b in 8 32 50; do
  for a in "" "--bf16x3-fwd" "--bf16x3-fwd --bf16x3-bwd"; do
    echo "batch $b flags [$a]: $(python bench.py --batch $b --steps 30 --warmup 8 --no-cpu-baseline --no-roofline $a 2>/dev/null | tail -1)"
  done
done
