# accumulate time per column for library variants: tools/wl_ab.sh <variant tags...>
for v in "$@"; do
  if [ "$v" = base ]; then unset ZKMI355_LIB; else export ZKMI355_LIB=$PWD/webauthn-halo2_amd/build/libzkmi355_$v.so; fi
  echo "== $v"; COLS=1,2,4 python tools/msm_window_ab.py 16 2>&1 | tail -4
done
