for v in base t10 t11; do
  if [ "$v" = base ]; then unset ZKMI355_LIB; else export ZKMI355_LIB=$PWD/webauthn-halo2_amd/build/libzkmi355_$v.so; fi
  echo "== $v"; python tools/ntt_sweep.py 2>&1 | tail -2
  for o in "" "--opt 3=10" "--opt 3=9"; do
  python bench.py --no-cpu-baseline --steps 40 $o 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v [$o]', 'proofs/s %.2f single %.2f'%(d['value'], d['single_proof_ms']))"
  done
done
