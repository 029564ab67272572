#!/bin/bash
# timing of several builds of the library (tools/ubench/lib<TAG>.bin, from tools/ubench/exp_c3.sh): alternating runs of
# tools/quick_time.py without the phase profile; usage: tools/ubench/abn.sh "A B C" [quick_time args]
TAGS=${1:-"A B"}; shift || true
for rep in 1 2 3; do
  for v in $TAGS; do
    cp tools/ubench/lib$v.bin jrl-qp_amd/csrc/libjrlqp_hip.so
    echo -n "$v: "; JRLQP_NOPROF=1 timeout -k 10 120 python tools/quick_time.py "$@" 2>&1 | grep "solves/s"
  done
done
