#!/bin/bash
# One GPU-box session made of named steps: each step runs under its own timeout and logs to gpurun_out/<tag>/<step>.log;
# a step that times out or is killed stops the chain (no further GPU work after a hang), an ordinary failure does not.
# usage: tools/gpu_session.sh <tag> <<'STEPS'
#   name timeout_s command...
# STEPS
set -u
TAG=${1:-session}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
while read -r name to cmd; do
    [ -z "${name:-}" ] && continue
    case "$name" in \#*) continue;; esac
    echo "=== $name ($(date +%T)): $cmd"
    timeout -k 10 "$to" bash -c "$cmd" > "$OUT/$name.log" 2>&1 < /dev/null
    rc=$?
    echo "rc=$rc" >> "$OUT/$name.log"
    echo "=== $name rc=$rc"
    tail -n 4 "$OUT/$name.log" | cut -c1-600
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out: stopping"; exit $rc; fi
done
echo "=== done"
