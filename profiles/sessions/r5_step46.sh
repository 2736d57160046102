# round 5, step 46: what else fits beside a conv_hd workgroup: unused LDS padded onto conv_hd<3, 8, 2> (101.5 KB -> 125 / 157 KB)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s46
mkdir -p $O
build() { rm -f canonicalvoting_amd/_C/obj/sparse_conv.hip.o; CV_SC_DEFS="$1" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1; }
run() {
  timeout 300 python bench.py --steps 240 --warmup 12 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 240 steps:', round(d['value'],1))" >> $O/hd_lds_pad.txt
}
run "pad 0 (101.5 KB: 56 KB left on the CU)"
build "-DHD_LDS_PAD=24000"; run "pad 24000 (125 KB: 35 KB left - one conv_hl workgroup, no vote)"
build "-DHD_LDS_PAD=56000"; run "pad 56000 (157 KB: nothing else fits)"
build ""; run "pad 0 again"
cat $O/hd_lds_pad.txt
