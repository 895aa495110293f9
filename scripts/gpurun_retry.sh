#!/bin/bash
# gpurun with retries while no box / slot is free (exit code 3: nothing charged).   usage: bash scripts/gpurun_retry.sh [--timeout S] -- 'command'
for attempt in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
