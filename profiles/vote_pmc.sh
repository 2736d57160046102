# HBM traffic of the vote kernel from the PMC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (with --kernel-trace
# only), per launch of hv_fwd_tiles; gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE (KB) counts half of a wide
# coalesced read stream -> doubled; WRITE_SIZE taken as is.  Writes gpurun_out/vote_pmc/vote_hbm_traffic.json (copied to profiles/r6/) (bench.py reads it).
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --streams 1 --stage vote_decode --steps 10 --warmup 2 --cpu-scenes 0 --measure-traffic 0"
rm -rf /tmp/vp_f /tmp/vp_w
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/vp_f --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/vp_w --output-format csv -- $CMD > /dev/null 2>&1
mkdir -p $R/gpurun_out/vote_pmc
cp $(find /tmp/vp_f -name "*counter_collection.csv" | head -1) $R/gpurun_out/vote_pmc/vote_pmc_fetch_size.csv
cp $(find /tmp/vp_w -name "*counter_collection.csv" | head -1) $R/gpurun_out/vote_pmc/vote_pmc_write_size.csv
cd $R && python - <<'PY'
import csv, json
def mean(path, counter):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if "hv_fwd_tiles" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(v) / len(v), len(v)
f, nf = mean("gpurun_out/vote_pmc/vote_pmc_fetch_size.csv", "FETCH_SIZE")
w, nw = mean("gpurun_out/vote_pmc/vote_pmc_write_size.csv", "WRITE_SIZE")
out = {"kernel": "hv_fwd_tiles<0>", "workload": "80k-point scene, bench.py --streams 1 --stage vote_decode (16 x 32-cell tiles, 8 waves per workgroup)",
       "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "launches": [nf, nw],
       "correction": "gfx950: FETCH_SIZE reads 1/2 of a wide coalesced stream (MI355X_MICROARCH.md HBM section) -> doubled; WRITE_SIZE uncalibrated, taken as is",
       "hbm_bytes_per_launch": (2 * f + w) * 1024.0,
       "source": ["profiles/r6/vote_pmc_fetch_size.csv", "profiles/r6/vote_pmc_write_size.csv"],
       "note": "the CSVs under profiles/r6 keep the hv_fwd_tiles rows only", "collected": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (profiles/vote_pmc.sh)"}
json.dump(out, open("gpurun_out/vote_pmc/vote_hbm_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
