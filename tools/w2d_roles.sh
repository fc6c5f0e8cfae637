#!/bin/bash
# role experiments for conv_w2d (ND_W2D_ROLES bits, see conv_w2d.hip)
for r in 0; do
  echo -n "roles $r: "
  ND_W2D_ROLES=$r timeout -k 10 120 python tools/bench_layers.py --batch 256 --winograd --layers convs1.2,convs2.2 2>&1 | grep "F(5,3)" | awk '{printf "%s %s ms  ", $1, $8}'; echo
done
