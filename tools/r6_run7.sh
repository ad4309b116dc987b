#!/bin/bash
# round 6, GPU call 7: the full-size audited proof; the suite on a build that poisons every MSM partial-sum buffer; a longer soak
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r6_run7
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_audit.py -q -k "lone_k19" > $O/audit_k19.txt 2>&1; tail -3 $O/audit_k19.txt
ZKMI355_LIB=$PWD/webauthn-halo2_amd/build/libzkmi355_poison.so timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_host_example.py --deselect tests/test_gpu_host_phases.py > $O/poison.txt 2>&1; tail -3 $O/poison.txt
( python tools/soak.py 1000 19 4; python tools/soak.py 500 19 1; OPTS=10=1 python tools/soak.py 500 19 4; python tools/soak.py 1500 17 4; python tools/soak.py 200 19 2 4 ) > $O/soak.txt 2>&1; cat $O/soak.txt
