# timing ablations of conv_rows_wp (CV_WP_ABL bits; results are wrong by construction): per-layer times of one forward
cd $GRAFT_REPO_ROOT
O=gpurun_out/wp_ablate; mkdir -p $O
for abl in 0 1 2 3 4 8 15; do
  touch canonicalvoting_amd/csrc/sparse_conv.hip
  CV_SC_DEFS="-DCV_WP_ABL=$abl" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  python profiles/layer_times.py > $O/abl_$abl.txt 2>&1
  echo "abl $abl: $(tail -1 $O/abl_$abl.txt)"
done
python - <<'PY'
import re
rows = {}
for abl in (0, 1, 2, 3, 4, 8, 15):
    for ln in open("gpurun_out/wp_ablate/abl_%d.txt" % abl):
        f = ln.split()
        if len(f) >= 7 and f[0].isdigit():
            rows.setdefault(int(f[0]), {})[abl] = float(f[6])
            rows[int(f[0])]["d"] = " ".join(f[1:6])
print("layer (n_out K cin cout grp) | us with CV_WP_ABL = 0 1 2 3 4 8 15")
for k in sorted(rows):
    r = rows[k]
    print("%2d %-26s" % (k, r["d"]), " ".join("%7.1f" % r.get(a, -1) for a in (0, 1, 2, 3, 4, 8, 15)))
PY
