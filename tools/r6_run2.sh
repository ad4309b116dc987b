#!/bin/bash
# round 6, GPU call 2: the soak test on the pre-fix library (28762db^) and on HEAD; new ADVICE tests; clock probe; bench
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6_run2
O=$PWD/gpurun_out/r6_run2
( cd .r6_old && for r in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_soak.py -x -q -s -k k19 2>&1 | tail -4; done ) > $O/soak_old.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_soak.py "tests/test_gpu_prove_batch.py::test_transcript_repr_set_between_batches_reaches_every_member" "tests/test_gpu_abi_errors.py::test_phase_calls_refuse_repeated_output_handles" -x -q -s > $O/new_tests.txt 2>&1
python - > $O/clock.txt 2>&1 <<'PY'
import webauthn_halo2_amd as zk
e = zk.Engine(0)
for ms in (20, 100, 100):
    c, r, m = e.clock_probe(ms)
    print("probe %d ms: memtime %d realtime %d mads %d -> sclk %.1f MHz (if memtime is the shader clock), %.2f memtime ticks per dependent mad" % (ms, c, r, m, c / r * 100.0, c / m))
print("mem", zk.engine.device_mem_info(0))
PY
bash tools/bench_ab.sh base base > $O/bench.txt 2>&1
tail -3 $O/soak_old.txt $O/new_tests.txt $O/clock.txt $O/bench.txt
