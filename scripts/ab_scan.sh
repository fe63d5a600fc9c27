#!/bin/bash
# Same-box A/B of two library builds on the exact scan / list-major IVF (tuning aid); see scripts/ab_libs.sh
L=semcode_amd/_lib
cp $L/libsemcode_hip.so $L/new.so
for v in new base new base; do
    if [ $v = new ]; then cp $L/new.so $L/libsemcode_hip.so; else cp $L/libsemcode_hip_base.so $L/libsemcode_hip.so; fi
    echo "== $v"
    timeout -k 10 200 python3 scripts/scan_rate.py 3072 2000000 2>&1 | grep "Q=16 SC_SCAN_QSTREAM=1\|Q= 6 SC_SCAN_QSTREAM=1"
done
cp $L/new.so $L/libsemcode_hip.so
