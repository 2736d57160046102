# memory-side traffic of one network forward: FETCH_SIZE and WRITE_SIZE summed over the conv / finish / stem launches of
# the last forwards of a one-in-flight run (separate --pmc passes with --kernel-trace only)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
# (--train-steps 0: round 3's number was wrong by 10x because the windows between the last stem launches held the
# training steps of the `train_step_ms` side field, whose kernels also carry "conv_" in their names)
CMD="python $R/bench.py --streams 1 --steps 6 --warmup 2 --cpu-scenes 0 --train-steps 0 --measure-traffic 0"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/nt_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/nt_$c --output-format csv -- $CMD > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/nt_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    stems = [i for i, r in enumerate(rows) if "conv_stem_mfma" in r["Kernel_Name"]]      # one per EVAL forward
    per = collections.defaultdict(float)
    nfw = 0
    for a, b in zip(stems[-4:-1], stems[-3:]):
        nfw += 1
        for r in rows[a:b]:
            n = r["Kernel_Name"]
            if "conv_hl" in n or "conv_finish" in n or "conv_stem_mfma" in n:
                key = "finish" if "finish" in n else "stem" if "stem" in n else "conv_hl"
                per[key] += float(r["Counter_Value"])
    assert nfw == 3, "expected three whole forwards between the last four stem launches"
    out[c] = {k: v / nfw / 1024.0 for k, v in per.items()}
    print(c, "MB per forward (raw KB counter / 1024):", {k: round(v, 1) for k, v in out[c].items()}, "sum %.1f" % sum(out[c].values()))
f2 = 2 * sum(out["FETCH_SIZE"].values()); w = sum(out["WRITE_SIZE"].values())
print("gfx950 correction (2 x FETCH_SIZE for wide coalesced reads; the gathers are 128-byte requests): fetch <= %.0f MB, write %.0f MB per forward" % (f2, w))
PY
