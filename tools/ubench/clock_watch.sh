#!/bin/bash
# Polls the GPU core clock / power while a command runs: tools/ubench/clock_watch.sh <outfile> <cmd...>
out=$1; shift
( for n in $(seq 1 80); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.15; done ) > "$out" &
poll=$!
"$@"
rc=$?
kill $poll 2>/dev/null
wait $poll 2>/dev/null
exit $rc
