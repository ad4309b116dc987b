#!/bin/bash
# round 6, GPU call 5: audit tests (after the pinned-memory fix), T1 / torchrun / batch tests; mad_first A/B; NTT / quotient block-asm variants; VALU pass
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r6_run5
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_audit.py tests/test_gpu_msm_wide.py tests/test_gpu_torchrun.py tests/test_gpu_prove_batch.py tests/test_gpu_abi_errors.py -q > $O/tests.txt 2>&1
tail -15 $O/tests.txt
for r in 1 2; do python tools/msm_parts.py; done > $O/parts.txt 2>&1
cat $O/parts.txt
for v in base nttblk quotblk base nttblk quotblk; do
  if [ $v = base ]; then unset ZKMI355_LIB; else export ZKMI355_LIB=$PWD/webauthn-halo2_amd/build/libzkmi355_$v.so; fi
  echo "== $v"; python tools/ntt_time.py 19; python tools/prove_timing.py 2>/dev/null | tail -3
done > $O/ntt_quot.txt 2>&1
unset ZKMI355_LIB
cat $O/ntt_quot.txt
bash tools/pmc_valu.sh r6 > $O/pmc_valu.txt 2>&1
tail -22 $O/pmc_valu.txt
