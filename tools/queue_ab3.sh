for cfg in "tail0 4 default" "tail0 5 default" "tail0 6 default" "tail0 8 default" "tail0 5 8" "tail0 6 8" "tail0 8 8" "tail0 8 16" "tail0 4 default"; do
  set -- $cfg
  if [ "$1" = base ]; then unset ZKMI355_LIB; else export ZKMI355_LIB=$PWD/webauthn-halo2_amd/build/libzkmi355_$1.so; fi
  if [ "$3" = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$3; fi
  python bench.py --no-cpu-baseline --steps 80 --inflight $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib $1 inflight $2 queues $3:', 'proofs/s %.2f (repeats %s) single %.2f'%(d['value'], ' '.join('%.1f'%x for x in d['value_repeats']), d['single_proof_ms']))"
done
