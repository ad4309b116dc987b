for v in "" s32g4 s32g8 s16g4; do
  if [ -n "$v" ]; then export ZKMI355_LIB=$PWD/webauthn-halo2_amd/build/libzkmi355_$v.so; else unset ZKMI355_LIB; fi
  for fl in 1 2; do
    python bench.py --inflight $fl --no-cpu-baseline --steps 12 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'inflight', $fl, 'proofs/s %.2f single %.2f accum_ms %.3f'%(d['value'], d['single_proof_ms'], d['roofline']['avg_launch_ms']))"
  done
done
