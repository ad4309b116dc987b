# tools/trace_timeline2.sh <tag> <K19|K17> <blake2b|evm> [from_us to_us]: a lone proof's kernel timeline by queue (tools/timeline2.py)
tag=$1; spec=${2:-K19}; kind=${3:-blake2b}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$tag
mkdir -p $O
( cd $R && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python tools/trace_one.py $spec $kind 8 > $O/run.log 2>&1 )
f=$(find $O/raw -name "*kernel_trace.csv" | head -1)
head -1 $f | cut -c1-400
python3 $R/tools/timeline2.py $f $4 $5 | tee $O/timeline2.txt
rm -rf $O/raw
