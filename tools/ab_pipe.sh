#!/bin/bash
# A/B of two builds (ab/lib_old.so, ab/lib_new.so) on the pipelined small-batch rate: ab_pipe.sh B STEPS
for i in 1 2 3; do
  for v in old new; do
    cp ab/lib_$v.so ppg-slam_amd/libppgfront.so
    echo -n "$v "; timeout -k 10 100 python tools/prof_pipe.py ${1:-1} ${2:-400} || exit 1
  done
done
cp ab/lib_new.so ppg-slam_amd/libppgfront.so
