# round 5, step 47: the vote's hot-plane parts re-tuned on the round-4/5 kernel (u64 part merges): records per part, parts per tile
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s47
mkdir -p $O
build() { rm -f canonicalvoting_amd/_C/obj/hv_vote.hip.o; CV_HV_DEFS="$1" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1; }
run() {
  timeout 300 python bench.py --steps 240 --warmup 12 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1: 240 steps', round(d['value'],1), '| vote kernel one in flight', round(r.get('isolated_avg_ms') or 0, 4), 'ms, isolated frac', round(r['isolated_frac'],3), ', in-region frac', round(r['frac'],3))" >> $O/vote_parts.txt
}
run "default (4096 records per part, <= 8 parts)"
build "-DHV_PART_RECORDS=2048"; run "2048 records per part"
build "-DHV_PART_RECORDS=8192"; run "8192 records per part"
build "-DHV_MAX_PARTS=4"; run "<= 4 parts"
build "-DHV_MAX_PARTS=16"; run "<= 16 parts"
build "-DHV_PART_RECORDS=2048 -DHV_MAX_PARTS=16"; run "2048 records, <= 16 parts"
build ""; run "default again"
cat $O/vote_parts.txt
