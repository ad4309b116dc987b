#!/bin/bash
# round 6, A/B 1: one asm statement per column part (ZK_MUL29_ASM=2) against one per multiply-add
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6_ab1
V=$PWD/webauthn-halo2_amd/build/libzkmi355_blk.so
( ZKMI355_LIB=$V timeout 600 python -m pytest tests/test_gpu_msm_wide.py tests/test_gpu_ops.py -x -q 2>&1 | tail -3
for r in 1 2; do
  python tools/msm_parts.py
  ZKMI355_LIB=$V python tools/msm_parts.py
done
tools/bench_ab.sh base blk base blk ) 2>&1 | tee gpurun_out/r6_ab1/log.txt
