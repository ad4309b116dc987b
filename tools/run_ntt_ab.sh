python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize_parity.py -x -q 2>&1 | tail -5 > gpurun_out/r4_t2.log
for v in base r3ntt foldonly; do
  if [ "$v" = base ]; then unset ZKMI355_LIB; else export ZKMI355_LIB=$PWD/webauthn-halo2_amd/build/libzkmi355_$v.so; fi
  echo "== $v"; python tools/ntt_time.py 19
done > gpurun_out/r4_ntt_ab.log 2>&1
unset ZKMI355_LIB
cat gpurun_out/r4_t2.log; cat gpurun_out/r4_ntt_ab.log
