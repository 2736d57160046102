"""profiles/in_flight_counters.sh -> profiles/r6/in_flight_counters.txt

argv: <trace dir> <pmc dir: SQ/GRBM> <pmc dir: FETCH_SIZE> <pmc dir: WRITE_SIZE> <bench json of the trace run>

Part 1 (kernel trace of the seven-in-flight run, nothing serialised): per kernel family the summed in-flight duration per scene,
and how the wall time of the timed region splits by the number of kernels running at once.
Part 2 (dispatch counters; the profiler runs one kernel at a time): per family and per scene the counters of each kernel with the
chip to itself.  Work conservation: sum over a scene's kernels of SQ_BUSY_CU_CYCLES (normalised by the ratio a chip-filling kernel
reaches = all 256 CUs busy) is the CU time the scene needs; divided by the CU time the chip offers per scene at the measured rate
it says whether the CUs are full (-> ~1) or the scene chains leave them idle (-> well below 1)."""
import collections
import csv
import glob
import json
import re
import sys


def family(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    for key, fam in (("conv_hl", "conv_hl"), ("conv_hd", "conv_hd"), ("conv_finish", "conv_finish"), ("conv_stem", "conv_stem"),
                     ("conv_rows", "conv_rows"), ("hv_fwd_tiles", "hv_fwd_tiles"), ("hv_", "hv_prep"),
                     ("minmax", "hv_prep"), ("dec_", "decode"), ("head_", "head"), ("build_kernel_maps", "plan"), ("mp_", "plan"),
                     ("sort_", "plan"), ("insert_all", "plan"), ("flag_levels", "plan"), ("emit_levels", "plan"), ("table_clear", "plan"),
                     ("up_map", "plan"), ("__amd_rocclr", "runtime copy/fill")):
        if n.startswith(key) or (key in n and key.startswith("__")):
            return fam
    return "other (" + re.split(r"[<(]", n)[0][:24] + ")"


def find(d, pat):
    f = glob.glob("%s/**/*%s" % (d, pat), recursive=True)
    if not f:
        raise SystemExit("no %s under %s" % (pat, d))
    return sorted(f)[-1]


def main():
    tdir, adir, bdir, cdir, bench_json = sys.argv[1:6]
    bench = json.loads(open(bench_json).read().strip().splitlines()[-1])
    rate, steps = bench["value"], bench["steps"]
    ms_per_scene = 1e3 / rate
    print("in-flight run: %.1f scenes/s over %d steps = %.3f ms of chip per scene, %d scenes in flight, policy %s"
          % (rate, steps, ms_per_scene, bench["config"]["scenes_in_flight_per_gpu"],
             {k: bench["config"][k] for k in ("conv_split_target", "vote_part_records", "masked_min_rows")}))
    # ---- part 1: the trace
    import gzip
    tfile = tdir if tdir.endswith(".gz") else find(tdir, "kernel_trace.csv")
    rows = list(csv.DictReader(gzip.open(tfile, "rt") if tfile.endswith(".gz") else open(tfile)))
    def wgs(r):
        try:
            g = [int(r[k]) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")]
            w = [max(1, int(r[k])) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z")]
            return max(1, (g[0] // w[0]) * max(1, g[1] // w[1]) * max(1, g[2] // w[2]))
        except (KeyError, ValueError):
            return int(r.get("Grid_Size", 0)) // max(1, int(r.get("Workgroup_Size", 1)))
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], wgs(r)) for r in rows]
    ev.sort()
    heads = [e for e in ev if family(e[2]) == "head"]
    # the steady state of the timed region: bench.py ends with its one-scene-in-flight side pass (2 x resident scenes to warm the
    # main thread's stream + max(min(steps, 48), 24) measured scenes), the timed region's `steps` scenes come before it; a scene
    # has exactly one head_joint launch.  20 scenes are cut off at either end (fill and drain of the seven scene threads).
    iso = (2 * 4 + max(min(steps, 48), 24)) if bench["config"]["scenes_in_flight_per_gpu"] > 1 else 0
    assert len(heads) >= steps + iso, (len(heads), steps, iso)
    t_lo = heads[-(steps + iso) + 20][0]
    t_hi = heads[-(iso + 20) - 1][1]
    ev = [e for e in ev if e[0] < t_hi]
    ev = [(s_, min(e_, t_hi), n_, w_) for s_, e_, n_, w_ in ev]
    reg = [e for e in ev if e[1] > t_lo]
    dur = collections.defaultdict(float)
    cnt = collections.Counter()
    for s, e, n, _ in reg:
        dur[family(n)] += (e - max(s, t_lo)) * 1e-6
        cnt[family(n)] += 1
    pts = []
    for i, (s, e, _, _) in enumerate(reg):
        pts.append((max(s, t_lo), 1, i))
        pts.append((e, -1, i))
    pts.sort()
    by_depth = collections.defaultdict(float)
    alone = collections.defaultdict(float)         # while exactly ONE kernel runs: which family, and does it fill the chip
    by_wgs = collections.defaultdict(float)        # wall time by the workgroups of all running kernels together
    running = set()
    last = t_lo
    for t, d, i in pts:
        dt = (t - last) * 1e-6
        by_depth[len(running)] += dt
        total_wgs = sum(reg[j][3] for j in running)
        by_wgs["0" if not running else "< 256" if total_wgs < 256 else "256-1023" if total_wgs < 1024 else ">= 1024"] += dt
        if len(running) == 1:
            j = next(iter(running))
            alone[(family(reg[j][2]), "< 256 workgroups" if reg[j][3] < 256 else ">= 256 workgroups")] += dt
        if d > 0:
            running.add(i)
        else:
            running.discard(i)
        last = t
    wall = (t_hi - t_lo) * 1e-6
    scenes = len([h for h in heads if t_lo < h[1] <= t_hi])
    print("\n== part 1: kernel trace of the in-flight run (not serialised) - %.1f ms of the timed region's steady state = %d scenes = %.1f scenes/s under the tracer"
          % (wall, scenes, scenes / wall * 1e3))
    print("kernels running at once -> share of the wall time:  " +
          "  ".join("%d: %.1f%%" % (k, 100 * v / wall) for k, v in sorted(by_depth.items()) if v / wall > 0.002))
    print("mean kernels running: %.2f;  time with NO kernel running: %.1f%%" %
          (sum(k * v for k, v in by_depth.items()) / wall, 100 * by_depth.get(0, 0.0) / wall))
    print("workgroups of all running kernels together -> share of the wall time:  " +
          "  ".join("%s: %.1f%%" % (k, 100 * by_wgs.get(k, 0.0) / wall) for k in ("0", "< 256", "256-1023", ">= 1024")) +
          "   (256 CUs: below 256 workgroups part of the chip is certainly idle)")
    print("while exactly one kernel runs (%.1f%% of the wall time) it is:  " % (100 * by_depth.get(1, 0.0) / wall) +
          "  ".join("%s %s: %.1f%%" % (f, w, 100 * v / wall) for (f, w), v in sorted(alone.items(), key=lambda kv: -kv[1])[:8]))
    print("%-22s %10s %16s" % ("family", "launches", "in-flight ms/scene"))
    for f, v in sorted(dur.items(), key=lambda kv: -kv[1]):
        print("%-22s %10.1f %16.3f" % (f, cnt[f] / scenes, v / scenes))
    print("%-22s %10.1f %16.3f  (sum of kernel durations per scene; %.2f x the %.3f ms of chip a scene gets)"
          % ("all", sum(cnt.values()) / scenes, sum(dur.values()) / scenes, sum(dur.values()) / scenes / ms_per_scene, ms_per_scene))
    if adir == "-":
        return
    # ---- part 2: the counters
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    n_scene = {}
    for d in (adir, bdir, cdir):
        rows = list(csv.DictReader(open(find(d, "counter_collection.csv"))))
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        ids = sorted({int(r["Dispatch_Id"]) for r in rows if family(r["Kernel_Name"]) == "head"})
        # whole scenes of the timed region only (its launch policy): the run ends with the one-in-flight side pass of
        # 2 x 4 + max(min(steps, 48), 24) scenes under the library's policy; the scenes are counted by their head launches
        p_steps = int(__import__("os").environ.get("PMC_STEPS", "56"))
        p_iso = 2 * 4 + max(min(p_steps, 48), 24)
        assert len(ids) >= p_steps + p_iso, (len(ids), p_steps, p_iso)
        lo, hi = ids[-(p_steps + p_iso) - 1], ids[-p_iso - 1]
        k = len([i for i in ids if lo < i <= hi])
        assert k == p_steps, (k, p_steps)
        for r in rows:
            if lo < int(r["Dispatch_Id"]) <= hi:
                per[family(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"]) / k
        n_scene[d] = k
    names = ["GRBM_GUI_ACTIVE", "SQ_BUSY_CU_CYCLES", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES",
             "FETCH_SIZE", "WRITE_SIZE"]
    print("\n== part 2: dispatch counters per scene (kernels serialised by the profiler; scenes averaged: %s)" % list(n_scene.values()))
    print("%-22s " % "family" + " ".join("%14s" % c[-14:] for c in names))
    tot = collections.defaultdict(float)
    for f, d in sorted(per.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
        print("%-22s " % f + " ".join("%14.4g" % d.get(c, 0.0) for c in names))
        for c in names:
            tot[c] += d.get(c, 0.0)
    print("%-22s " % "all" + " ".join("%14.4g" % tot[c] for c in names))
    # normalisation: the family whose kernels fill the chip (the vote tiles: >= 2 workgroups on every CU for the whole launch)
    v = per["hv_fwd_tiles"]
    full = v["SQ_BUSY_CU_CYCLES"] / v["GRBM_GUI_ACTIVE"]
    print("\nSQ_BUSY_CU_CYCLES / GRBM_GUI_ACTIVE of hv_fwd_tiles (every CU busy) = %.2f -> the unit of 'all 256 CUs busy'" % full)
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs by the profiler
    clk_cycles_per_scene = tot["GRBM_GUI_ACTIVE"] / 8.0
    print("serialised chip time per scene: GRBM_GUI_ACTIVE / 8 XCDs = %.3e cycles" % clk_cycles_per_scene)
    print("%-22s %12s %12s %14s" % ("family", "CU-busy share", "MFMA busy", "waves parked"))
    for f, d in sorted(per.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
        if d.get("GRBM_GUI_ACTIVE", 0) <= 0:
            continue
        cu = d["SQ_BUSY_CU_CYCLES"] / d["GRBM_GUI_ACTIVE"] / full
        mf = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024)
        park = d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"] if d["SQ_WAVE_CYCLES"] else 0.0
        print("%-22s %12.3f %12.3f %14.3f" % (f, cu, mf, park))
    cu_need = tot["SQ_BUSY_CU_CYCLES"] / full          # in units of (GRBM cycles summed over XCDs) with all CUs busy
    print("\nwork conservation (the two numbers of the bench line):")
    # chip time a scene gets in flight, in the same clock: serialised cycles x (in-flight ms per scene / serialised ms per scene)
    ser_ms = clk_cycles_per_scene / 2.4e6             # at the 2.4 GHz nominal clock (the profiled passes run ~1.9-2.0 GHz: an upper bound on speed)
    print("  serialised kernel time per scene ~ %.3f ms at 2.4 GHz (%.3f ms at 1.95 GHz); in flight a scene gets %.3f ms of chip"
          % (ser_ms, clk_cycles_per_scene / 1.95e6, ms_per_scene))
    cu_busy = cu_need / tot["GRBM_GUI_ACTIVE"]
    print("  CU-busy share of the serialised run: %.3f (CU time the kernels occupy / CU time of their launches)" % cu_busy)
    for clk in (2.4e6, 1.95e6):
        need_ms = cu_need / 8.0 / clk                  # ms of a chip with ALL CUs busy that a scene's kernels occupy
        print("  cu_busy_in_flight = %.3f ms of all-CU time per scene / %.3f ms per scene = %.3f  (clock %.2f GHz)"
              % (need_ms, ms_per_scene, need_ms / ms_per_scene, clk / 1e6))
    mf_all = tot["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0   # cycles of one SIMD's matrix pipe, averaged over the chip's 1024 SIMDs
    for clk in (2.4e6, 1.95e6):
        print("  mfma_busy_in_flight = %.4f ms of all-SIMD matrix time per scene / %.3f = %.3f  (clock %.2f GHz)"
              % (mf_all / clk, ms_per_scene, mf_all / clk / ms_per_scene, clk / 1e6))
    out = {"source": "profiles/in_flight_counters.sh (rocprofv3 --pmc dispatch counters of bench.py's timed-region scenes under the "
                     "seven-in-flight launch policy, kernels serialised by the profiler; rate from the kernel-trace run)",
           "scenes_per_s_under_tracer": rate, "ms_per_scene": ms_per_scene,
           "cu_busy_in_flight": cu_need / 8.0 / 2.4e6 / ms_per_scene, "mfma_busy_in_flight": mf_all / 2.4e6 / ms_per_scene,
           "clock_ghz_assumed": 2.4, "kernels_running_at_once": {str(k): round(v / wall, 4) for k, v in sorted(by_depth.items())},
           "mean_kernels_running": sum(k * v for k, v in by_depth.items()) / wall,
           "wall_share_with_at_least_1024_workgroups_running": by_wgs.get(">= 1024", 0.0) / wall,
           "fetch_mb_per_scene_raw": tot["FETCH_SIZE"] / 1024, "write_mb_per_scene": tot["WRITE_SIZE"] / 1024,
           "per_family": {f: {c: d.get(c, 0.0) for c in names} for f, d in per.items()}}
    jpath = __import__("os").environ.get("IN_FLIGHT_JSON")
    if jpath:
        json.dump(out, open(jpath, "w"), indent=1)
    print("  HBM-side traffic per scene: FETCH %.1f MB raw (<= %.1f with the gfx950 wide-read correction) + WRITE %.1f MB -> %.2f-%.2f TB/s at the in-flight rate"
          % (tot["FETCH_SIZE"] / 1024, 2 * tot["FETCH_SIZE"] / 1024, tot["WRITE_SIZE"] / 1024,
             (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / 1024 / 1e6 / (ms_per_scene * 1e-3),
             (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / 1024 / 1e6 / (ms_per_scene * 1e-3)))


if __name__ == "__main__":
    main()
