#!/bin/bash
# sample GPU clocks / power while a command runs: smi_watch.sh <outfile> <cmd...>
out=$1; shift
( while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|edge)" | tr '\n' ' ' ; echo; sleep 0.25; done ) > $out 2>&1 &
wp=$!
"$@"
rc=$?
kill $wp 2>/dev/null
exit $rc
