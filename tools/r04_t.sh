#!/bin/bash
out=gpurun_out/$1; mkdir -p $out; shift
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest "$@" > $out/gputest.log 2>&1; echo "pytest rc $?"; tail -25 $out/gputest.log
