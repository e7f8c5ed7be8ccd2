#!/bin/bash
# thin 1x1 layers: one forced block tile per process (TD_CONV_CFG), product build
for c in ${CFGS:-0 10 15 16 18}; do TD_CONV_CFG=$c timeout -k 10 120 python3 tools/conv_diag.py thin 2>/dev/null || exit 1; done
