# round 5, step 19: where the in-launch group sum loses its time: poll interval, ablations (WRONG results), conv_hh underneath
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s19
mkdir -p $O
run() {  # label, env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --steps 240 --warmup 12 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label 240 steps:', round(d['value'],1))" >> $O/rates.txt
  env "$@" timeout 300 python bench.py --steps 120 --warmup 12 --streams 1 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label one in flight:', round(d['value'],1))" >> $O/rates.txt
}
build() { rm -f canonicalvoting_amd/_C/obj/sparse_conv.hip.o; CV_SC_DEFS="$1" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1; }
run "finish launches" CV_GFUSE=0
run "gfuse sleep 8" CV_GFUSE=1
run "gfuse sleep 8 conv_hh" CV_GFUSE=1 CV_HD_SHAPE=3
build "-DGF_SLEEP=100"
run "gfuse sleep 100" CV_GFUSE=1
build "-DGF_ABL=1"
run "gfuse no wait (wrong)" CV_GFUSE=1
build "-DGF_ABL=3"
run "gfuse no wait no partial reads (wrong)" CV_GFUSE=1
cat $O/rates.txt
