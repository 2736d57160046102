# Counters for the regime bench.py times: `bench.py` with its default scene threads (seven in flight).
#   pass T   kernel trace only (does NOT serialise): start / end of every kernel of the in-flight run -> how long the chip has
#            0, 1, 2 ... kernels running, the in-flight duration of every kernel family
#   pass A-C rocprofv3 --pmc (dispatch counters: the profiler runs the kernels ONE AT A TIME - what they measure is each
#            kernel's work with the chip to itself, not the overlap): GRBM_GUI_ACTIVE, SQ_BUSY_CU_CYCLES, SQ_WAVES,
#            SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_VALU_MFMA_BUSY_CYCLES | FETCH_SIZE | WRITE_SIZE, separate passes
# profiles/in_flight_summary.py turns the four outputs into profiles/r6/in_flight_counters.txt.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
STEPS=${STEPS:-240}
CMD="python $R/bench.py --steps $STEPS --cpu-scenes 0 --train-steps 0 --measure-traffic 0"
rm -rf /tmp/ifT /tmp/ifA /tmp/ifB /tmp/ifC
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/ifT -- $CMD > $O/if_trace_bench.json 2> /tmp/ifT.err
PCMD="python $R/bench.py --steps 56 --warmup 14 --min-warm-seconds 0 --cpu-scenes 0 --train-steps 0 --measure-traffic 0"
timeout 1500 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d /tmp/ifA --output-format csv -- $PCMD > /tmp/ifA.log 2>&1
timeout 1500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/ifB --output-format csv -- $PCMD > /tmp/ifB.log 2>&1
timeout 1500 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/ifC --output-format csv -- $PCMD > /tmp/ifC.log 2>&1
IN_FLIGHT_JSON=$O/in_flight_counters.json python $R/profiles/in_flight_summary.py /tmp/ifT /tmp/ifA /tmp/ifB /tmp/ifC $O/if_trace_bench.json > $O/in_flight_counters.txt 2> $O/in_flight_counters.err
tail -40 $O/in_flight_counters.txt
