#!/bin/bash
# gpurun wrapper: the GPU box gets a snapshot without .git, so the commit the measurements belong to is written to
# .gpurun_commit first (git-ignored, travels with the snapshot); tools/pmc_summary.py, tools/pmc_mfma.py and
# tools/profile_round.sh copy it into every summary they write (bench.py's `measured_at_commit`).
#   tools/gpu.sh [--timeout S] -- '<command>'
cd "$(dirname "$0")/.."
c=$(git rev-parse --short=12 HEAD)
git diff --quiet HEAD -- . ':!PROGRESS.jsonl' || c="$c-dirty"
echo "$c" > .gpurun_commit
exec /usr/local/graft/bin/gpurun "$@"
