"""The C-ABI library builds for gfx950 without a GPU, loads, and exports exactly what
include/cv_hip.h declares.  Host-only entry points are exercised here; device entry points
are only checked for presence (no compute calls without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

import oracle
from canonicalvoting_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "cv_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cv_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built_lib):
    L = ctypes.CDLL(built_lib)
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(L, s), "libcvhip.so does not export %s" % s
        assert s in _lib.SIGNATURES, "no ctypes signature for %s" % s
    assert sorted(_lib.SIGNATURES) == syms
    header_version = int(re.search(r"#define\s+CV_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "cv_hip.h")).read()).group(1))
    assert _lib.lib().cv_abi_version() == header_version == _lib.ABI_VERSION       # library, header and ctypes signatures agree


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.CvError, match="no CPU fallback"):
        _lib.lib()


def test_grid_dims_host_matches_oracle(built_lib):
    L = _lib.lib()
    res = np.float32(0.03)
    rng = np.random.default_rng(0)
    for _ in range(500):
        lo = rng.integers(-200, 200, 3)
        ext = rng.integers(1, 400, 3)
        mn = (lo.astype(np.float32) * res).astype(np.float32)
        mx = ((lo + ext).astype(np.float32) * res).astype(np.float32)
        d = (ctypes.c_int * 3)()
        assert L.cv_hv_grid_dims_f32(mn.ctypes.data_as(_lib.c_float_p), mx.ctypes.data_as(_lib.c_float_p),
                                     ctypes.c_float(res), d) == 0
        od = (ctypes.c_int * 3)()
        oracle.lib().hv_oracle_grid_dims(mn.ctypes.data_as(_lib.c_float_p),
                                         mx.ctypes.data_as(_lib.c_float_p), ctypes.c_float(res), od)
        assert list(d) == list(od)
        assert all(dd in (e, e + 1) for dd, e in zip(d, ext))
    for n in (2, 4, 11, 13, 248, 285):      # SURVEY 8a V1 pitfall vectors
        mn = np.float32([-150 * res, 0, 0]); mx = np.float32([np.float32(-150 + n) * res, 1, 1])
        d = (ctypes.c_int * 3)()
        L.cv_hv_grid_dims_f32(mn.ctypes.data_as(_lib.c_float_p), mx.ctypes.data_as(_lib.c_float_p),
                              ctypes.c_float(res), d)
        assert d[0] == n


def test_bad_arguments_return_error_codes(built_lib):
    L = _lib.lib()
    d = (ctypes.c_int * 3)()
    assert L.cv_hv_grid_dims_f32(None, None, ctypes.c_float(0.03), d) == -22
    assert b"null" in L.cv_last_error()
    assert L.cv_nms_obb(None, None, 3, 0.3, None) == -22


def rand_boxes(rng, n):
    raw = np.array([[1, 1, 1], [1, 1, -1], [-1, 1, -1], [-1, 1, 1], [1, -1, 1], [1, -1, -1],
                    [-1, -1, -1], [-1, -1, 1]], np.float64)
    out = []
    for _ in range(n):
        t = rng.uniform(0, 2 * np.pi)
        Rm = np.array([[np.cos(t), 0, -np.sin(t)], [0, 1, 0], [np.sin(t), 0, np.cos(t)]])
        out.append((raw * rng.uniform(0.2, 0.9, 3)) @ Rm.T + rng.uniform(0, 2.5, 3) * [1, 0.2, 1])
    return np.array(out, np.float32)


def test_host_iou_and_nms_match_oracle(built_lib):
    from canonicalvoting_amd import decode
    rng = np.random.default_rng(1)
    boxes = rand_boxes(rng, 40)
    nz = 0
    for i in range(40):
        for j in range(40):
            a, b = decode.get_iou_obb(boxes[i], boxes[j]), oracle.iou_obb(boxes[i], boxes[j])
            assert abs(a - b) < 1e-12
            nz += a > 0
    assert nz > 100
    scores = rng.uniform(0, 1, 40).astype(np.float32)
    scores[5] = scores[6]
    for thr in (0.1, 0.3, 0.6):
        assert decode.nms(boxes, scores, thr) == oracle.nms(boxes, scores, thr)
    cls = rng.integers(0, 9, 40)
    a = decode.nms_per_class(boxes, scores, cls)
    b = oracle.nms_per_class(boxes, scores, cls)
    assert [(x[0], x[2]) for x in a] == [(x[0], x[2]) for x in b]
    assert all(np.array_equal(x[1], y[1]) for x, y in zip(a, b))
    assert decode.nms(boxes[:0], scores[:0], 0.3) == []


def test_host_side_layout_functions(built_lib):
    """pure host functions of the network executor / scene maps / tile plan: sizes and offsets (no GPU needed)"""
    import ctypes
    L = _lib.lib()
    rows = (ctypes.c_int64 * 5)(80000, 36822, 9929, 2349, 494)
    off = _lib.SceneMaps()
    words = L.cv_sp_scene_maps_words(rows, 80000, 5, 4, 16384, ctypes.byref(off))
    assert off.out == -1          # the caller's rows <- sorted rows map is cv_sp_sort_rows' inverse permutation
    spans = [(off.stem, 80000 * 125)] + [(off.down[i], rows[i + 1] * 8) for i in range(4)] + \
            [(off.k3[i], rows[i] * 27) for i in range(5)] + [(off.up[i], rows[3 - i] * 8) for i in range(4)] + \
            [(off.up_perm[i], rows[3 - i]) for i in range(4)] + [(off.scratch, (5 * 4 + 4) * 2048), (off.bitmap, 1 << 20)]
    for i in range(5):
        if rows[i] >= 16384:
            assert off.mask_perm[i] >= 0
            spans.append((off.mask_perm[i], 4 * rows[i] * (1 + 7)))      # orders + map rows in processing order
        else:
            assert off.mask_perm[i] == -1
    spans.sort()
    for (a, la), (b, _) in zip(spans[:-1], spans[1:]):
        assert a % 64 == 0 and a + la <= b                               # aligned, non-overlapping
    assert spans[-1][0] + spans[-1][1] <= words
    assert words <= L.cv_sp_scene_plan_words(80000, 5, 4, 16384)      # the pre-count sizing covers it
    # arena of a two-buffer program: one external, one level-1 buffer of 96 channels
    bufs = (_lib.NetBuf * 2)(_lib.NetBuf(-1, 3, 0), _lib.NetBuf(1, 96, 1))
    assert L.cv_net_arena_bytes(bufs, 2, rows, 5) >= 36822 * 96 * 4


def test_no_packed_fp32_instructions_in_device_code(built_lib, tmp_path):
    """gfx950: `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` returns wrong sums while another wave of the CU issues
    v_mfma_f32_32x32x16_{f16,bf16} / 16x16x32_f16 (profiles/r3/vote_concurrency_findings.txt): the
    library is built with the packed-fp32 target feature off (csrc/build.py), and no code object may carry such an
    instruction.  Disassembles the gfx950 image of every object file."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not found")
    obj_dir = os.path.join(os.path.dirname(built_lib), "obj")
    seen = 0
    for name in sorted(os.listdir(obj_dir)):
        if not name.endswith(".o"):
            continue
        local = shutil.copy(os.path.join(obj_dir, name), tmp_path / name)
        subprocess.run([objdump, "--offloading", str(local)], check=True, capture_output=True)
        images = [f for f in os.listdir(tmp_path) if f.startswith(name + ".") and "gfx950" in f]
        for img in images:              # (host-only translation units carry no device image)
            asm = subprocess.run([objdump, "-d", str(tmp_path / img)], check=True, capture_output=True, text=True).stdout
            packed = re.findall(r"\bv_pk_(?:add|mul|fma)_f32\b", asm)
            assert not packed, "%s: %d packed fp32 instructions" % (name, len(packed))
            seen += 1
    assert seen >= 4          # hv_vote, hv_decode, sparse_coords, sparse_conv


def test_compiled_hv_cuda_extension_loads_and_checks_its_inputs(built_lib):
    """the compiled pybind / torch extension `hv_cuda` (csrc/hv_cuda_ext.cpp = houghvoting/src/hv_cuda.cpp:30-77 over the
    C ABI): builds without a GPU, is a module named hv_cuda with forward / backward, and refuses CPU / non-contiguous
    tensors with the reference's messages (hv_cuda.cpp:26-28) - no compute call without a GPU"""
    import torch
    from canonicalvoting_amd import hv_cuda_ext
    from canonicalvoting_amd.csrc import build
    assert os.path.exists(build.build_ext())
    m = hv_cuda_ext.load()
    assert m.__name__ == "hv_cuda" and m.abi_version() == _lib.lib().cv_abi_version()
    assert m.__file__.endswith(".so") and os.path.dirname(m.__file__) == os.path.dirname(built_lib)      # in-tree, next to libcvhip.so
    p, s = torch.rand(10, 3), torch.rand(10)
    res, rots = torch.tensor(0.03), torch.tensor(120, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="points must be a CUDA tensor"):
        m.forward(p, p, p, s, res, rots)
    with pytest.raises(RuntimeError, match="grad_grid must be a CUDA tensor"):
        m.backward(torch.zeros(2, 2, 2), p, p, p, s, res, rots)
    import inspect
    doc = m.forward.__doc__
    for name in ("points", "xyz_labels", "scale_labels", "obj_labels", "res", "num_rots"):      # hv_cuda.cpp:30-36 order
        assert name in doc
    assert doc.index("points") < doc.index("xyz_labels") < doc.index("scale_labels") < doc.index("obj_labels") < doc.index("num_rots")
    assert inspect.ismodule(hv_cuda_ext.install()) and __import__("hv_cuda") is m
    import sys
    del sys.modules["hv_cuda"]


def test_kernel_selection_knobs(built_lib):
    """cv_sp_set_option / cv_sp_get_option (host only)"""
    import ctypes
    L = _lib.lib()
    assert L.cv_sp_set_option(b"no_such_knob", 1, None) != 0
    # cv_sp_get_option reads a knob without touching it (what the executor uses while other threads launch)
    for name in (b"hd_mask", b"hd_min_rows", b"hd_shape", b"zskip"):
        v, old = ctypes.c_longlong(-7), ctypes.c_longlong(-9)
        assert L.cv_sp_get_option(name, ctypes.byref(v)) == 0
        assert L.cv_sp_set_option(name, v.value, ctypes.byref(old)) == 0 and old.value == v.value
    assert L.cv_sp_get_option(b"no_such_knob", ctypes.byref(v)) != 0 and L.cv_sp_get_option(b"zskip", None) != 0


def test_in_flight_launch_sizing_policy(built_lib):
    """pipeline.configure_for_scenes_in_flight: the library defaults below four scenes in flight, the in-flight sizes from four on
    (host-side settings only: no GPU needed); cv_hv_set_part_records never goes below the 4096 the workspace bound assumes"""
    from canonicalvoting_amd import me as ME, pipeline
    L = _lib.lib()
    try:
        cfg = pipeline.configure_for_scenes_in_flight(7)
        assert cfg == {"conv_split_target": 256, "vote_part_records": 12288, "masked_min_rows": min(8192, ME.CoordinateManager.LIB_MASKED_MIN_ROWS)}
        assert ME.CoordinateManager.MASKED_MIN_ROWS == cfg["masked_min_rows"]
        assert L.cv_hv_set_part_records(100) == 12288 and L.cv_hv_set_part_records(0) == 4096      # (clamped up, then the default)
        assert L.cv_hv_set_part_records(20000) == 4096 and L.cv_hv_set_part_records(0) == 20000
        cfg = pipeline.configure_for_scenes_in_flight(1)
        assert cfg["conv_split_target"] == 0 and cfg["vote_part_records"] == 0
        assert ME.CoordinateManager.MASKED_MIN_ROWS == ME.CoordinateManager.LIB_MASKED_MIN_ROWS
        assert L.cv_hv_set_part_records(0) == 4096
    finally:
        pipeline.configure_for_scenes_in_flight(1)


def test_launch_policy_travels_with_the_call_not_the_process(built_lib):
    """pipeline.ScenePolicy / scene_policy: the three in-flight knobs as the CALLING THREAD's values for the duration of a
    block (cv_sp_set_split_target_thread, cv_hv_set_part_records_thread, ME.masked_min_rows()), restored on exit, invisible
    to other threads and to the process-wide values; cv_scene_desc carries the same three for the one-call path"""
    import threading
    from canonicalvoting_amd import me as ME, pipeline
    from canonicalvoting_amd.minkunet import MinkUNet34C
    L = _lib.lib()
    lib_rows = ME.CoordinateManager.LIB_MASKED_MIN_ROWS
    assert pipeline.policy_for_scenes_in_flight(3) == pipeline.ScenePolicy(0, 0, lib_rows)
    pol = pipeline.policy_for_scenes_in_flight(4)
    assert pol == pipeline.ScenePolicy(256, 12288, min(8192, lib_rows)) == pipeline.policy_for_scenes_in_flight(7)
    assert {"conv_split_target", "vote_part_records", "masked_min_rows"} <= {f[0] for f in _lib.SceneDesc._fields_}
    process_target = L.cv_sp_set_split_target(0)
    L.cv_sp_set_split_target(process_target)
    model = MinkUNet34C(3, 8)
    seen = {}
    with pipeline.scene_policy(pol):
        assert ME.masked_min_rows() == model.masked_min_rows() == pol.masked_min_rows
        assert L.cv_sp_set_split_target_thread(256) == 256 and L.cv_hv_set_part_records_thread(12288) == 12288
        with pipeline.scene_policy(pipeline.ScenePolicy(512, 100, 4096)):               # nests; part records clamp up to 4096
            assert ME.masked_min_rows() == 4096 and L.cv_sp_set_split_target_thread(512) == 512
            assert L.cv_hv_set_part_records_thread(4096) == 4096
        assert ME.masked_min_rows() == pol.masked_min_rows and L.cv_sp_set_split_target_thread(256) == 256

        def other():
            seen["rows"] = ME.masked_min_rows()
            seen["target"] = L.cv_sp_set_split_target_thread(0)
            seen["records"] = L.cv_hv_set_part_records_thread(0)
        t = threading.Thread(target=other)
        t.start()
        t.join()
        assert ME.CoordinateManager.MASKED_MIN_ROWS == lib_rows                          # nothing process-wide moved
        assert L.cv_sp_set_split_target(process_target) == process_target
    assert seen == {"rows": lib_rows, "target": 0, "records": 0}
    assert ME.masked_min_rows() == lib_rows and L.cv_sp_set_split_target_thread(0) == 0 and L.cv_hv_set_part_records_thread(0) == 0
    with pipeline.scene_policy(None):
        assert ME.masked_min_rows() == lib_rows
    model.MASKED_MIN_ROWS = 1234                                   # a pinned model value yields to a policy, not to the process
    assert model.masked_min_rows() == 1234
    with pipeline.scene_policy(pol):
        assert model.masked_min_rows() == pol.masked_min_rows


def test_a_changed_define_makes_the_object_stale(built_lib, monkeypatch):
    """csrc/build.py keeps the command line of every object beside it: another -D through CV_*_DEFS (or another HIPCC)
    is a rebuild of exactly that source, not a silent re-use (round 5 lost a table of ablations to that)."""
    from canonicalvoting_amd.csrc import build as b
    for k in ("CV_HV_DEFS", "CV_DEC_DEFS", "CV_SC_DEFS"):
        monkeypatch.delenv(k, raising=False)
    assert b.plan() == []
    monkeypatch.setenv("CV_SC_DEFS", "-DX=1")
    assert b.plan() == ["sparse_conv.hip"]
    monkeypatch.setenv("CV_HV_DEFS", "-DHV_TX=16 -DHV_TW=8")
    assert b.plan() == ["hv_vote.hip", "sparse_conv.hip"]
    monkeypatch.delenv("CV_SC_DEFS")
    monkeypatch.delenv("CV_HV_DEFS")
    assert b.plan() == []
    obj = os.path.join(b.OBJ_DIR, "cv_host.cpp.o")
    assert open(obj + ".cmd").read().split("\n")[0] == b.HIPCC and not b._stale(obj, [])
    assert b._stale(obj, [], ["another", "command"])


def test_scene_call_refuses_bad_descriptors_before_touching_the_gpu(built_lib):
    """cv_detect_scene_f32: null pointers and negative launch sizing come back as CV_EINVAL with a message (host-side checks only:
    no GPU needed); cv_scene_result is zeroed first, host_us included"""
    import ctypes
    L = _lib.lib()
    d, r = _lib.SceneDesc(), _lib.SceneResult()
    assert L.cv_detect_scene_f32(ctypes.byref(d), ctypes.byref(r), None) == -22 and b"scene descriptor" in L.cv_last_error()
    fake = ctypes.c_void_p(4096)          # never dereferenced: the sizing check comes before the first launch
    for f in ("d_coords4", "d_feats", "d_points", "ops", "bufs", "d_out_feats", "h_pinned", "d_ws", "h_boxes", "h_scores", "h_classes",
              "h_cand_idx", "h_verdict", "h_pick"):
        setattr(d, f, fake)
    d.n, d.n_ops, d.n_bufs, d.out_ld, d.out_channels, d.pinned_bytes, d.max_candidates = 100, 1, 1, 64, 64, 256, 8
    r.host_us[0] = 7.0
    for field in ("conv_split_target", "vote_part_records"):
        setattr(d, field, -1)
        assert L.cv_detect_scene_f32(ctypes.byref(d), ctypes.byref(r), None) == -22 and b"negative launch sizing" in L.cv_last_error()
        assert r.host_us[0] == 0.0 and r.n_cand == 0
        setattr(d, field, 0)
    assert L.cv_sp_set_split_target_thread(0) == 0 and L.cv_hv_set_part_records_thread(0) == 0      # nothing left behind
