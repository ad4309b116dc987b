for v in old base old base; do
  if [ "$v" = base ]; then unset ZKMI355_LIB; else export ZKMI355_LIB=$PWD/webauthn-halo2_amd/build/libzkmi355_$v.so; fi
  python bench.py --no-cpu-baseline --steps 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'proofs/s %.2f single %.2f evm %.2f accum_ms %.3f'%(d['value'], d['single_proof_ms'], d['single_proof_evm_ms'], d['roofline']['avg_launch_ms']))"
done
