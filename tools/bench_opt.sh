# bench.py under engine options: tools/bench_opt.sh "<--opt a=b ...>" "<--opt ...>" ...   ("" = defaults)
for o in "$@"; do
  python bench.py --no-cpu-baseline --steps 40 $o 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$o]', 'proofs/s %.2f single %.2f evm %.2f'%(d['value'], d['single_proof_ms'], d['single_proof_evm_ms']))"
done
