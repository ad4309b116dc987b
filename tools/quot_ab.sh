for v in base q0 q16 q18; do
  if [ "$v" = base ]; then unset ZKMI355_LIB; else export ZKMI355_LIB=$PWD/webauthn-halo2_amd/build/libzkmi355_$v.so; fi
  echo "== $v"; ROWS=11,12,13,14,15,16 python tools/bench_rows.py 7 | tail -6
done
