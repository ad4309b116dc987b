# A/B of NTT builds: tools/ntt_ab.sh <script.py> <variant tags...> (libraries built by tools/ab_variants.sh)
script=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset ZKMI355_LIB; else export ZKMI355_LIB=$PWD/webauthn-halo2_amd/build/libzkmi355_$v.so; fi
  echo "== variant $v"; python $script 2>&1 | tail -4
done
