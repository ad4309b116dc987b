#!/bin/bash
# round 6, GPU call 8: do plain instructions pair up in runs? (ubench) and the result-limb masks as one run per product (variant maskrun)
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r6_run8
mkdir -p $O
tools/ubench_isa 2>&1 | grep "mad\|v_and" > $O/ubench.txt; cat $O/ubench.txt
V=$PWD/webauthn-halo2_amd/build/libzkmi355_maskrun.so
ZKMI355_LIB=$V timeout 300 python -m pytest tests/test_gpu_msm_wide.py -q -x 2>&1 | tail -2
for r in 1 2 3; do python tools/msm_parts.py; ZKMI355_LIB=$V python tools/msm_parts.py; done > $O/parts.txt 2>&1; cat $O/parts.txt
