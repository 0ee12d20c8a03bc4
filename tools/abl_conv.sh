#!/bin/bash
# Timing ablations of the pipelined conv kernel on two layers (results are wrong by construction). Usage: tools/abl_conv.sh "0 1 2 ..."
for A in ${1:-0 1 2 3 4 5 6 7 8 16}; do echo "ABL=$A"; for SH in 40x40-320-320-3 80x80-320-320-3; do CDET_CONV_ABLATE=$A timeout 120 python tools/conv_shapes.py --what fwd --only $SH 2>&1 | grep "^fwd" | grep " 3 1 "; done; done
