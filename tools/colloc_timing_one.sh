#!/bin/bash
# usage: colloc_timing_one.sh <lib.so> : runs tools/colloc_timing.py against that build (child process, bounded)
CFZ_LIBRARY=$1 CFZ_COLLOC_PROFILE=1 timeout 300 python tools/colloc_timing.py 2>&1 | grep -v "^  File\|Extension modules" | tail -4
