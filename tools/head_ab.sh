# MSM head kernels (lone 2^19 columns, tools/prof_ops.py) under library variants: tools/head_ab.sh <tags...>
for v in "$@"; do
  if [ "$v" = base ]; then unset ZKMI355_LIB; else export ZKMI355_LIB=$GRAFT_REPO_ROOT/webauthn-halo2_amd/build/libzkmi355_$v.so; fi
  echo "== $v"
  bash tools/prof_cmd.sh headab_$v python tools/prof_ops.py 19 6 2>&1 | grep -E "msm_w|msm_wacc" | grep -v "table"
done
