# stream / hardware-queue topology under bench.py: the in-tree build and the shared-tail-stream build, with HIP's default
# number of hardware queues and with 2 / 8 (GPU_MAX_HW_QUEUES is read by the HIP runtime, not by the engine)
for q in default 2 8; do
  for v in base tail1; do
    if [ "$v" = base ]; then unset ZKMI355_LIB; else export ZKMI355_LIB=$PWD/webauthn-halo2_amd/build/libzkmi355_$v.so; fi
    if [ "$q" = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    python bench.py --no-cpu-baseline --steps 60 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues $q lib $v', 'proofs/s %.2f (repeats %s) single %.2f'%(d['value'], ' '.join('%.1f'%x for x in d['value_repeats']), d['single_proof_ms']))"
  done
done
