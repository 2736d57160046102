# round 5, step 62: vote tile shapes with seven scenes in flight and 12288 records per part
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s62
mkdir -p $O
build() { rm -f canonicalvoting_amd/_C/obj/hv_vote.hip.o; CV_HV_DEFS="$1" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1; }
run() {
  for i in 1 2; do
  timeout 300 python bench.py --steps 240 --warmup 12 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1: 240 steps', round(d['value'],1), '| vote kernel alone', round(r.get('isolated_avg_ms') or 0, 4), 'ms, in-region frac', round(r['frac'],3))" >> $O/vote_shapes.txt
  done
}
run "16 x 32 cells, 8 waves (default)"
build "-DHV_TW=4"; run "16 x 32 cells, 4 waves"
build "-DHV_TX=32 -DHV_TW=16"; run "32 x 32 cells, 16 waves"
build "-DHV_TX=32 -DHV_TW=8"; run "32 x 32 cells, 8 waves"
build ""
cat $O/vote_shapes.txt
