# tools/trace_timeline_opt.sh <tag> <spec> <kind> <msm window>: one proof's timeline with an MSM window override
tag=$1; spec=${2:-K19}; kind=${3:-blake2b}; win=${4:-0}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$tag
mkdir -p $O
( cd $R && MSM_WINDOW=$win timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python tools/trace_one.py $spec $kind 8 > $O/run.log 2>&1 )
f=$(find $O/raw -name "*kernel_trace.csv" | head -1)
python3 $R/tools/timeline.py $f --full > $O/timeline_full.txt
rm -rf $O/raw
grep -v "us " $O/timeline_full.txt | head -40
grep "bytes" $O/run.log
