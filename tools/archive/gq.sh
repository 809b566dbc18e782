#!/bin/bash
# quick GPU call: tools/gq.sh '<command>'  (prints the tail of stdout)
/usr/local/graft/bin/gpurun --timeout ${GQ_TIMEOUT:-300} -- "$1" 2>&1 | tail -${GQ_TAIL:-60}
