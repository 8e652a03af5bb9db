#!/bin/bash
# Like clock_watch.sh with timestamps, so samples can be matched to the phases the command prints.
out=$1; shift
( for n in $(seq 1 200); do echo -n "$(date +%s.%N | cut -c1-14) "; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | grep -o -E "\([0-9]+Mhz\)|: [0-9.]+ *$" | tr '\n' ' '; echo; sleep 0.1; done ) > "$out" &
poll=$!
"$@" | while IFS= read -r line; do echo "$(date +%s.%N | cut -c1-14) $line"; done
kill $poll 2>/dev/null
wait $poll 2>/dev/null
