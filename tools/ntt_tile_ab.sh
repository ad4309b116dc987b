# NTT build variants side by side: solo transform times (tools/ntt_sweep.py) and bench.py: tools/ntt_tile_ab.sh <variant tags...>
for v in "$@"; do
  if [ "$v" = base ]; then unset ZKMI355_LIB; else export ZKMI355_LIB=$PWD/webauthn-halo2_amd/build/libzkmi355_$v.so; fi
  echo "== $v"; python tools/ntt_sweep.py 2>&1 | tail -2 | cut -c1-20,60-100
  python bench.py --no-cpu-baseline --steps 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'proofs/s %.2f single %.2f'%(d['value'], d['single_proof_ms']))"
done
