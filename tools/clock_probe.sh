#!/bin/bash
# Sample shader clock / power while a kernel loop runs: tools/clock_probe.sh <python args...>
( for i in 1 2 3 4 5 6 7 8; do sleep 1; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket" | tr '\n' ' '; echo; done ) &
SAMPLER=$!
timeout 60 python "$@" > /dev/null 2>&1
wait $SAMPLER
