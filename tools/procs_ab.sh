# One process x four pipelines against two processes x two pipelines on ONE GPU: the in-process activity count
# (ZK_OPT_MSM_TAIL_STREAM auto) does not see the other process, so each process of the pair runs its tails on side streams unless
# pinned.  tools/procs_ab.sh   (sums of the two processes' own proofs/s; their timed regions overlap: started together, same set-up)
cd $GRAFT_REPO_ROOT
val() { python -c "
import json,sys
v=[json.loads(open(f).read().strip().splitlines()[-1])['value_advice_resident'] for f in sys.argv[1:]]
print(' + '.join('%.1f'%x for x in v), '= %.1f proofs/s'%sum(v))" "$@"; }
B="python bench.py --no-cpu-baseline --steps 80"
echo "== k=19, 1 process x 4 pipelines"; $B --inflight 4 2>/dev/null > gpurun_out/pa_1.log; val gpurun_out/pa_1.log
echo "== k=19, 2 processes x 2 pipelines, tails auto (side streams: each process sees two active contexts)"
$B --inflight 2 2>/dev/null > gpurun_out/pa_2.log & $B --inflight 2 2>/dev/null > gpurun_out/pa_3.log & wait; val gpurun_out/pa_2.log gpurun_out/pa_3.log
echo "== k=19, 2 processes x 2 pipelines, tails pinned to the main streams (--opt 5=2)"
$B --inflight 2 --opt 5=2 2>/dev/null > gpurun_out/pa_4.log & $B --inflight 2 --opt 5=2 2>/dev/null > gpurun_out/pa_5.log & wait; val gpurun_out/pa_4.log gpurun_out/pa_5.log
echo "== k=19, 2 processes x 1 pipeline x lock-step 4"
$B --inflight 1 --lockstep 4 2>/dev/null > gpurun_out/pa_6.log & $B --inflight 1 --lockstep 4 2>/dev/null > gpurun_out/pa_7.log & wait; val gpurun_out/pa_6.log gpurun_out/pa_7.log
echo "== k=17 EVM, 1 process: 2 pipelines, 4 pipelines, 2 x lock-step 4"
python tools/inflight_k17.py 2 4 2x4 | cut -c1-80
echo "== k=17 EVM, 2 processes x 2 pipelines (sum the two lines)"
python tools/inflight_k17.py 2 > gpurun_out/pa_8.log 2>&1 & python tools/inflight_k17.py 2 > gpurun_out/pa_9.log 2>&1 & wait
tail -1 gpurun_out/pa_8.log | cut -c1-80; tail -1 gpurun_out/pa_9.log | cut -c1-80
echo "== k=17 EVM, 2 processes x 1 pipeline x lock-step 4 (sum the two lines)"
python tools/inflight_k17.py 1x4 > gpurun_out/pa_10.log 2>&1 & python tools/inflight_k17.py 1x4 > gpurun_out/pa_11.log 2>&1 & wait
tail -1 gpurun_out/pa_10.log | cut -c1-80; tail -1 gpurun_out/pa_11.log | cut -c1-80
echo "== k=17 EVM, 2 processes x 2 pipelines x lock-step 4 (sum the two lines)"
python tools/inflight_k17.py 2x4 > gpurun_out/pa_12.log 2>&1 & python tools/inflight_k17.py 2x4 > gpurun_out/pa_13.log 2>&1 & wait
tail -1 gpurun_out/pa_12.log | cut -c1-80; tail -1 gpurun_out/pa_13.log | cut -c1-80
