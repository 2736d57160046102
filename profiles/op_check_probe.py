"""Which instruction class gives different results when waves issuing v_mfma_f32_32x32x16_f16 share its CU?
profiles/microbench/lds_hammer.hip: op_check folds a few thousand deterministic evaluations per class into per-thread
checksums; reference = a run with nothing else on the GPU; then the same launch next to each co-resident load."""
import ctypes, os, sys, threading
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
H = ctypes.CDLL(os.path.join(HERE, "microbench", "liblds_hammer.so"))
H.lds_hammer_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
H.op_check_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
BLOCKS, ITERS, NCLASS = 512, 3000, 24
CLASSES = ["fp32 division", "v_rcp_f32", "v_sqrt_f32", "floor / cvt i32", "fp32 mul-add chain", "f64 fma (fixed point)", "f64 division",
           "ds_bpermute", "ballot + mbcnt", "readlane", "int64 mul / add", "LDS indexed read", "LDS u64 atomic", "polynomial atan",
           "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_mul_f32 / v_add_f32",
           "v_pk_mul op_sel_hi:[0,1]", "v_pk_add x,x op_sel:[0,1] op_sel_hi:[1,0] neg:[0,1]", "v_pk_add neg_lo/hi:[0,1]",
           "v_pk_add op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mul op_sel:[1,0] op_sel_hi:[0,1]", "v_pk_fma op_sel_hi:[0,1,1]"]
ref = torch.zeros(NCLASS * BLOCKS * 512, dtype=torch.int64, device=dev)
mism = torch.zeros(NCLASS, dtype=torch.int32, device=dev)
sink = torch.zeros(65536, device=dev)
src = torch.rand(1 << 22, device=dev)
main = torch.cuda.Stream(dev)
H.op_check_launch(BLOCKS, ITERS, ref.data_ptr(), 1, mism.data_ptr(), ctypes.c_void_p(main.cuda_stream))
main.synchronize()
stop = False
def load(mode):
    st = torch.cuda.Stream(dev)
    while not stop:
        for _ in range(8):
            H.lds_hammer_launch(mode, 768, 200, sink.data_ptr(), src.data_ptr(), src.numel(), ctypes.c_void_p(st.cuda_stream))
        st.synchronize()
NAMES = {0: "nothing", 1: "LDS b128 traffic", 3: "v_mfma_f32_32x32x16_f16", 4: "v_mfma_f32_32x32x2_f32", 8: "v_mfma_f32_32x32x16_bf16",
         9: "v_mfma_f32_16x16x32_f16", 10: "v_mfma_f32_32x32x8_f16"}
for mode in (0, 10, 9, 8, 3):
    stop = False
    bg = [threading.Thread(target=load, args=(mode,)) for _ in range(4 if mode else 0)]
    [t.start() for t in bg]
    mism.zero_()
    torch.cuda.synchronize()
    for _ in range(20):
        H.op_check_launch(BLOCKS, ITERS, ref.data_ptr(), 0, mism.data_ptr(), ctypes.c_void_p(main.cuda_stream))
    main.synchronize()
    stop = True
    [t.join() for t in bg]
    torch.cuda.synchronize()
    m = mism.cpu().tolist()
    print("next to %-26s threads with a changed checksum (of %d x 20):" % (NAMES[mode], BLOCKS * 512),
          {CLASSES[k]: m[k] for k in range(NCLASS) if m[k]} or "none", flush=True)
