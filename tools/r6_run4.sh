#!/bin/bash
# round 6, GPU call 4: audit + T1 + two-rank tests; ISA rates; T1 per bucket against per part (lone pass and under bench.py); VALU counter pass
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r6_run4
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_audit.py tests/test_gpu_msm_wide.py tests/test_gpu_torchrun.py tests/test_gpu_prove_batch.py -x -q > $O/tests.txt 2>&1
tail -5 $O/tests.txt
tools/ubench_isa > $O/ubench_isa.txt 2>&1
for r in 1 2; do
  OPTS=10=2 python tools/msm_parts.py; OPTS=10=1 python tools/msm_parts.py
done > $O/t1_parts.txt 2>&1
cat $O/t1_parts.txt
for r in 1 2; do
  for o in 10=2 10=1; do
    python bench.py --no-cpu-baseline --k17-steps 0 --steps 40 --opt $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$o', 'value %.2f first %.2f repeats %s single %.2f'%(d['value'], d['value_first'], ['%.1f'%x for x in d['value_repeats']], d['single_proof_ms']))"
  done
done > $O/t1_bench.txt 2>&1
cat $O/t1_bench.txt
bash tools/pmc_valu.sh r6 > $O/pmc_valu.txt 2>&1
tail -20 $O/pmc_valu.txt
