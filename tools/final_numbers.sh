#!/bin/bash
# All bench configurations quoted in DESIGN.md section 6, one JSON line each -> gpurun_out/final/*.json
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/final; mkdir -p $O
run() { name=$1; shift; python3 bench.py "$@" 2>/dev/null | grep '^{' | tail -1 > $O/$name.json; python3 - <<PY
import json; d=json.load(open("$O/$name.json")); r=d.get("roofline") or {}; c=d.get("cpu_baseline") or {}
print("%-22s %9.2f %s  %7.3f ms/step  roofline %s %s/%s (%s)  cpu %s" % ("$name", d["value"], d["unit"], d["ms_per_step"], r.get("kernel"), r.get("achieved"), r.get("peak"), r.get("frac"), c.get("value")))
PY
}
run train_b8 --steps 40 --warmup 10
run train_b32 --steps 20 --warmup 5 --batch 32 --no-cpu-baseline
run train_b50 --steps 20 --warmup 5 --batch 50 --no-cpu-baseline
run train_audio_b8 --steps 40 --warmup 10 --workload audio --no-cpu-baseline
run train_b8_bf16x3fwd --steps 40 --warmup 10 --bf16x3-fwd --no-cpu-baseline
run train_b8_bf16x3 --steps 40 --warmup 10 --bf16x3-fwd --bf16x3-bwd --no-cpu-baseline
run train_b32_bf16x3 --steps 20 --warmup 5 --batch 32 --bf16x3-fwd --bf16x3-bwd --no-cpu-baseline
run train_b50_bf16x3 --steps 20 --warmup 5 --batch 50 --bf16x3-fwd --bf16x3-bwd --no-cpu-baseline
run infer_b8 --mode infer --steps 40 --warmup 10
run infer_b32 --mode infer --steps 20 --warmup 5 --batch 32 --no-cpu-baseline
run infer_b8_bf16x3 --mode infer --steps 40 --warmup 10 --bf16x3 --no-cpu-baseline
run infer_b32_bf16x3 --mode infer --steps 20 --warmup 5 --batch 32 --bf16x3 --no-cpu-baseline
run e2e_b8 --mode e2e --steps 4 --warmup 1
run e2e_b8_bf16x3 --mode e2e --steps 4 --warmup 1 --bf16x3 --no-cpu-baseline
