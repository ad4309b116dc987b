# pipelines in flight x tail-stream topology x hardware queues under bench.py (tools/ab_variants.sh builds tail1 / tail0)
for cfg in "base 2 default" "tail1 2 default" "tail1 3 default" "tail1 3 8" "tail1 4 8" "tail0 2 default" "tail0 3 default" "tail0 4 default" "tail0 4 8" "tail1 2 default"; do
  set -- $cfg
  if [ "$1" = base ]; then unset ZKMI355_LIB; else export ZKMI355_LIB=$PWD/webauthn-halo2_amd/build/libzkmi355_$1.so; fi
  if [ "$3" = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$3; fi
  python bench.py --no-cpu-baseline --steps 60 --inflight $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib $1 inflight $2 queues $3:', 'proofs/s %.2f (repeats %s) single %.2f'%(d['value'], ' '.join('%.1f'%x for x in d['value_repeats']), d['single_proof_ms']))"
done
