#!/bin/bash
# Run GPU steps one after the other: an ordinary failure (assertion) lets the next step run, a step that timed out or was
# killed (rc 124 / 137 / >= 128) stops the sequence (no further GPU work after a hang).
# usage: tools/gpu_steps.sh "name|seconds|command" ...
mkdir -p gpurun_out
overall=0
for spec in "$@"; do
    name="${spec%%|*}"; rest="${spec#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
    echo "=== step $name (limit ${secs}s): $cmd"
    timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
    rc=$?
    echo "=== step $name rc=$rc"; tail -n 6 "gpurun_out/$name.log"
    if [ $rc -ne 0 ]; then overall=$rc; fi
    if [ $rc -ge 124 ]; then echo "=== stopping: step $name timed out or was killed"; break; fi
done
exit $overall
