#!/bin/bash
# needs the diagnostic build: make -C nind_denoise_amd/csrc clean && make -C nind_denoise_amd/csrc STAMPS=1
# timing ablations of conv_w2d on one layer shape (ND_W2D_DBG bits: 1 no epilogue stores, 2 no epilogue, 4 no weight DMA,
# 8 no pixel loads, 16 no transform / V writes, 32 VALU stand-in for the MFMAs, 64 no barrier)
for d in 128 0; do
  echo "== dbg $d"
  ND_W2D_DBG=$d timeout -k 10 120 python tools/bench_layers.py --batch 256 --winograd --layers convs1.2 2>&1 | grep "F(5,3)\|stamps"
done
