# A/B of library builds under bench.py: tools/bench_ab.sh <variant tags...> ("base" = the in-tree build)
for v in "$@"; do
  if [ "$v" = base ]; then unset ZKMI355_LIB; else export ZKMI355_LIB=$PWD/webauthn-halo2_amd/build/libzkmi355_$v.so; fi
  python bench.py --no-cpu-baseline --steps 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'proofs/s %.2f single %.2f evm %.2f accum_ms %.3f msm_head %s'%(d['value'], d['single_proof_ms'], d['single_proof_evm_ms'], d['roofline']['avg_launch_ms'], d['roofline']['note'][70:130]))"
done
