# Durations of the MSM kernels in one lone k=19 proof for library variants: tools/tail_times.sh <variant tags...> ("base" = in-tree)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset ZKMI355_LIB; else export ZKMI355_LIB=$R/webauthn-halo2_amd/build/libzkmi355_$v.so; fi
  O=$R/gpurun_out/tt_$v; mkdir -p $O
  ( cd $R && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python tools/trace_one.py ${SPEC:-K19} ${KIND:-blake2b} 8 > $O/run.log 2>&1 )
  f=$(find $O/raw -name "*kernel_trace.csv" | head -1)
  echo "== $v $(grep bytes $O/run.log)"
  python3 $R/tools/timeline.py $f | grep -E "period|msm_w|msm_gather|msm_bitsum|union busy"
  rm -rf $O/raw
done
