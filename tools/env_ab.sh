#!/bin/bash
# in-situ A/B of environment knobs: tools/env_ab.sh "VAR=a VAR2=b" "VAR=c" ...   (each argument = one configuration; "" = defaults)
for cfg in "$@"; do
  best=0
  for i in 1 2; do
    v=$(env $cfg python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
    best=$(python -c "print(max($best, $v))")
  done
  echo "[$cfg] $best clips/s"
done
