#!/bin/bash
cd /root/repo
for v in w2g4 w3g3 w4g2; do
  echo "== wave kernel, $v"
  MP_VI_PERSIST_BLOCK=64 MI355PLAN_LIB=build_variants/$v/libmi355plan.so python tools/micro_vi_persist.py child 2>&1 | grep "C2.*1000"
  echo "== block 256, $v"
  MP_VI_PERSIST_BLOCK=256 MI355PLAN_LIB=build_variants/$v/libmi355plan.so python tools/micro_vi_persist.py child 2>&1 | grep "C2.*1000"
done
