#!/bin/bash
# GEMM rate of the trailing-update shape against k and the tile variant, alone on the masked main stream
for m in 7168 5120 3072; do for k in 384 768 1152; do for t in 0 65; do python scratch/gemm_time.py $m $k 12 $t; done; done; done
