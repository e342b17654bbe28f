#!/bin/bash
# builds scratch/lib_kb_<variant>.so: the product library with kbuild.hip compiled under a measurement macro
cd /root/repo/gptools_amd/csrc
for v in ${VARIANTS:-NOCOMPUTE NOSTORE}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DKB_DEBUG_$v -c kbuild.hip -o /tmp/kbuild_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/lib_kb_$v.so /tmp/kbuild_$v.o build/kbuild_batch.o build/kbuild_prod.o build/gemm.o build/potrf.o build/solve.o build/api.o
done
