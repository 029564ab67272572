#!/bin/bash
# A/B timing of two builds of the library (tools/ubench/libA.bin, libB.bin): alternating runs of tools/quick_time.py
for rep in 1 2 3; do
  for v in A B; do
    cp tools/ubench/lib$v.bin jrl-qp_amd/csrc/libjrlqp_hip.so
    echo -n "$v: "; JRLQP_NOPROF=1 timeout -k 10 120 python tools/quick_time.py "$@" 2>&1 | grep "solves/s"
  done
done
