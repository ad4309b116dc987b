# Does the HIP runtime's per-process launch path bound the throughput?  The same pipelines as threads of one process against
# separate processes on one GPU.  tools/multiproc_test.sh
cd $GRAFT_REPO_ROOT
echo "== k=17: one process, 2 and 4 pipelines"
python tools/inflight_k17.py 2 4 | tail -2
echo "== k=17: two processes x 2 pipelines (sum the two lines)"
python tools/inflight_k17.py 2 > gpurun_out/mp_a.log 2>&1 &
python tools/inflight_k17.py 2 > gpurun_out/mp_b.log 2>&1 &
wait
tail -1 gpurun_out/mp_a.log; tail -1 gpurun_out/mp_b.log
echo "== k=19: one process x 2 pipelines"
python bench.py --no-cpu-baseline --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"
echo "== k=19: two processes x 1 pipeline (sum)"
python bench.py --no-cpu-baseline --steps 60 --inflight 1 2>/dev/null > gpurun_out/mp_c.log &
python bench.py --no-cpu-baseline --steps 60 --inflight 1 2>/dev/null > gpurun_out/mp_d.log &
wait
python -c "
import json
for f in ('gpurun_out/mp_c.log','gpurun_out/mp_d.log'):
    print(json.loads(open(f).read().strip().splitlines()[-1])['value'])"
echo "== k=19: two processes x 2 pipelines (sum)"
python bench.py --no-cpu-baseline --steps 60 2>/dev/null > gpurun_out/mp_e.log &
python bench.py --no-cpu-baseline --steps 60 2>/dev/null > gpurun_out/mp_f.log &
wait
python -c "
import json
for f in ('gpurun_out/mp_e.log','gpurun_out/mp_f.log'):
    print(json.loads(open(f).read().strip().splitlines()[-1])['value'])"
