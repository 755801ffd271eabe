#!/bin/bash
# A/B of one environment variable (unset vs 1) on the pipelined small-batch rate and the blocking latency: ab_pipe_env.sh VAR B STEPS
var=$1
for i in 1 2 3; do
  for v in - 1; do
    if [ "$v" = "-" ]; then unset $var; else export $var=1; fi
    echo -n "$var=$v "; timeout -k 10 100 python tools/prof_pipe.py ${2:-1} ${3:-400} || exit 1
  done
done
