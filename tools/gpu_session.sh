#!/bin/bash
# One gpurun call: each step under its own timeout; a step that timed out or was killed ends the session (no further GPU
# work after a hang), a step that merely failed (assertions) does not.
#   tools/gpu_session.sh "<seconds> <command ...>" ...
mkdir -p gpurun_out
for spec in "$@"; do
  secs=${spec%% *}; cmd=${spec#* }
  echo "=== [$(date +%H:%M:%S)] $cmd (limit ${secs}s)"
  timeout -k 10 "$secs" bash -c "$cmd"
  rc=$?
  echo "=== rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out / killed: stopping"; exit $rc; fi
done
exit 0
