// Compiled drop-in for the reference's pybind / torch extension module `hv_cuda`
// (houghvoting/src/hv_cuda.cpp:74-77, built there by houghvoting/setup.py:5-10): the same two entry points with the same
// positional arguments and the same input checks, over the C ABI of libcvhip.so (include/cv_hip.h).
//
//   forward(points, xyz_labels, scale_labels, obj_labels, res, num_rots[, corners])
//       -> [grid_obj[X,Y,Z], grid_rot[X,Y,Z,2], grid_scale[X,Y,Z,3]]                   (hv_cuda.cpp:30-45)
//   backward(grad_grid, points, xyz_labels, scale_labels, obj_labels, res, num_rots[, corners])
//       -> [d_xyz_labels, d_scale_labels, d_obj_labels]                                 (hv_cuda.cpp:47-71)
//
// A maintainer of the reference drops the built module into houghvoting/ (or onto PYTHONPATH) and `import hv_cuda`
// (eval_joint.py:10) resolves to it; HVFunction / HoughVoting (eval_joint.py:24-57) are unchanged.  What differs from
// the CUDA extension, on purpose: kernels go to torch's CURRENT stream (the reference uses the legacy default stream,
// hv_cuda_kernel.cu:143,158,286), and the host waits ONCE per forward - the bounds of the points and the two 0-dim
// device scalars (res, num_rots: dereferenced on the device by the reference, :22-23,152-153) travel to pinned memory
// behind one stream synchronisation instead of twelve `.item()` round trips (:132-134,151).
#include <torch/extension.h>

#include <ATen/hip/HIPContext.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "cv_hip.h"

namespace {

// hv_cuda.cpp:26-28
#define CHECK_CUDA(x) TORCH_CHECK((x).is_cuda(), #x " must be a CUDA tensor")
#define CHECK_CONTIGUOUS(x) TORCH_CHECK((x).is_contiguous(), #x " must be contiguous")
#define CHECK_INPUT(x) \
    CHECK_CUDA(x);     \
    CHECK_CONTIGUOUS(x)

void cv_check(int rc, const char* what) {
    if (rc == CV_OK) return;
    const char* msg = cv_last_error();
    TORCH_CHECK(false, what, " failed (", rc, "): ", msg ? msg : "");
}

// pinned landing buffers (8 floats: min xyz, max xyz, res, num_rots bits), recycled: page-locked allocations cost ~1 ms
struct PinnedPool {
    std::mutex mu;
    std::vector<float*> free_list;
    float* take() {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!free_list.empty()) {
                float* p = free_list.back();
                free_list.pop_back();
                return p;
            }
        }
        float* p = nullptr;
        TORCH_CHECK(hipHostMalloc(reinterpret_cast<void**>(&p), 64, hipHostMallocDefault) == hipSuccess,
                    "hv_cuda: hipHostMalloc failed");
        return p;
    }
    void give(float* p) {
        std::lock_guard<std::mutex> lk(mu);
        free_list.push_back(p);
    }
};
PinnedPool g_pinned;
struct PinnedLease {
    float* p;
    PinnedLease() : p(g_pinned.take()) {}
    ~PinnedLease() { g_pinned.give(p); }
};

struct Checked {
    int64_t n;
    float res;
    int num_rots;
    float mn[3], mx[3];
};

// the reference's input checks (hv_cuda.cpp:36-41,56-62) + what its kernels silently assume; then ONE host wait for the
// scalars and (unless the caller supplies the grid box) the bounds of the points
Checked checked_inputs(const torch::Tensor& points, const torch::Tensor& xyz_labels, const torch::Tensor& scale_labels,
                       const torch::Tensor& obj_labels, const torch::Tensor& res, const torch::Tensor& num_rots,
                       bool need_bounds, hipStream_t st) {
    CHECK_INPUT(points);
    CHECK_INPUT(xyz_labels);
    CHECK_INPUT(scale_labels);
    CHECK_INPUT(obj_labels);
    CHECK_INPUT(res);
    CHECK_INPUT(num_rots);
    TORCH_CHECK(points.scalar_type() == torch::kFloat32, "points must be float32 (got ", points.scalar_type(), ")");
    TORCH_CHECK(xyz_labels.scalar_type() == torch::kFloat32, "xyz_labels must be float32 (got ", xyz_labels.scalar_type(), ")");
    TORCH_CHECK(scale_labels.scalar_type() == torch::kFloat32, "scale_labels must be float32 (got ", scale_labels.scalar_type(), ")");
    TORCH_CHECK(obj_labels.scalar_type() == torch::kFloat32, "obj_labels must be float32 (got ", obj_labels.scalar_type(), ")");
    const int64_t n = points.dim() == 2 ? points.size(0) : -1;
    TORCH_CHECK(n >= 0 && points.size(1) == 3 && xyz_labels.dim() == 2 && xyz_labels.size(0) == n && xyz_labels.size(1) == 3 &&
                    scale_labels.dim() == 2 && scale_labels.size(0) == n && scale_labels.size(1) == 3 &&
                    obj_labels.dim() == 1 && obj_labels.size(0) == n,
                "expected points/xyz_labels/scale_labels [N,3] and obj_labels [N]");
    // the reference fails inside torch::min on an empty tensor (hv_cuda_kernel.cu:129)
    TORCH_CHECK(n > 0, "hv_cuda: cannot vote with zero points");
    TORCH_CHECK(res.numel() == 1 && res.scalar_type() == torch::kFloat32, "res must be a 0-dim float32 tensor");
    TORCH_CHECK(num_rots.numel() == 1 && (num_rots.scalar_type() == torch::kInt32 || num_rots.scalar_type() == torch::kInt64),
                "num_rots must be a 0-dim int32 tensor");
    Checked c{};
    c.n = n;
    PinnedLease host;
    torch::Tensor ws;
    if (need_bounds) {
        ws = torch::empty({(int64_t)cv_hv_minmax_workspace_bytes()}, points.options().dtype(torch::kUInt8));
        cv_check(cv_hv_minmax_async_f32(points.data_ptr<float>(), n, host.p, ws.data_ptr(), (size_t)ws.numel(), st),
                 "cv_hv_minmax_async_f32");
    }
    TORCH_CHECK(hipMemcpyAsync(host.p + 6, res.data_ptr(), 4, hipMemcpyDeviceToHost, st) == hipSuccess, "hv_cuda: copy of res failed");
    TORCH_CHECK(hipMemcpyAsync(host.p + 8, num_rots.data_ptr(), num_rots.element_size(), hipMemcpyDeviceToHost, st) == hipSuccess,
                "hv_cuda: copy of num_rots failed");
    TORCH_CHECK(hipStreamSynchronize(st) == hipSuccess, "hv_cuda: stream synchronisation failed");
    c.res = host.p[6];
    c.num_rots = num_rots.scalar_type() == torch::kInt32 ? *reinterpret_cast<const int32_t*>(host.p + 8)
                                                         : (int)*reinterpret_cast<const int64_t*>(host.p + 8);
    for (int k = 0; k < 3; ++k) { c.mn[k] = host.p[k]; c.mx[k] = host.p[3 + k]; }
    return c;
}

// corners[2,3] of the SUN RGB-D caller (sunrgbd/brnetcanon.py:99,236-242): grid origin corners[0], extent from
// (corners[1] - corners[0]) / res
void box_from_corners(const torch::Tensor& corners, float mn[3], float mx[3]) {
    TORCH_CHECK(corners.numel() == 6, "corners must hold [2,3] values");
    const torch::Tensor c = corners.detach().to(torch::kCPU, torch::kFloat32).contiguous();
    const float* p = c.data_ptr<float>();
    for (int k = 0; k < 3; ++k) { mn[k] = p[k]; mx[k] = p[3 + k]; }
}

int g_algo = 0;

std::vector<torch::Tensor> hv_forward(torch::Tensor points, torch::Tensor xyz_labels, torch::Tensor scale_labels,
                                      torch::Tensor obj_labels, torch::Tensor res, torch::Tensor num_rots,
                                      c10::optional<torch::Tensor> corners) {
    CHECK_CUDA(points);
    // (PyTorch-ROCm presents HIP devices under the "cuda" device type: the guard and stream accessors that accept it are
    // the ...MasqueradingAsCUDA ones; this is what at::cuda::CUDAGuard / getCurrentCUDAStream hipify to)
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(points.device());
    hipStream_t st = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(points.device().index()).stream();
    Checked c = checked_inputs(points, xyz_labels, scale_labels, obj_labels, res, num_rots, !corners.has_value(), st);
    if (corners.has_value()) box_from_corners(*corners, c.mn, c.mx);
    int dims[3];
    cv_check(cv_hv_grid_dims_f32(c.mn, c.mx, c.res, dims), "cv_hv_grid_dims_f32");     // hv_cuda_kernel.cu:131-134 in fp32
    // fresh, writable, non-aliased outputs on the inputs' device (points.options(), :132-134): callers zero cells of
    // grid_obj in place (eval_joint.py:211,243)
    auto grid_obj = torch::empty({dims[0], dims[1], dims[2]}, points.options());
    auto grid_rot = torch::empty({dims[0], dims[1], dims[2], 2}, points.options());
    auto grid_scale = torch::empty({dims[0], dims[1], dims[2], 3}, points.options());
    const size_t wsb = cv_hv_forward_workspace_bytes(c.n, c.num_rots, dims, g_algo);
    auto ws = torch::empty({(int64_t)std::max<size_t>(wsb, 256)}, points.options().dtype(torch::kUInt8));
    cv_check(cv_hv_forward_f32(points.data_ptr<float>(), xyz_labels.data_ptr<float>(), scale_labels.data_ptr<float>(),
                               obj_labels.data_ptr<float>(), c.n, c.res, c.num_rots, c.mn, dims, grid_obj.data_ptr<float>(),
                               grid_rot.data_ptr<float>(), grid_scale.data_ptr<float>(), ws.data_ptr(), (size_t)ws.numel(),
                               g_algo, st),
             "cv_hv_forward_f32");
    return {grid_obj, grid_rot, grid_scale};
}

std::vector<torch::Tensor> hv_backward(torch::Tensor grad_grid, torch::Tensor points, torch::Tensor xyz_labels,
                                       torch::Tensor scale_labels, torch::Tensor obj_labels, torch::Tensor res,
                                       torch::Tensor num_rots, c10::optional<torch::Tensor> corners) {
    CHECK_INPUT(grad_grid);
    CHECK_CUDA(points);
    TORCH_CHECK(grad_grid.scalar_type() == torch::kFloat32 && grad_grid.dim() == 3, "grad_grid must be a float32 [X,Y,Z] tensor");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(points.device());
    hipStream_t st = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(points.device().index()).stream();
    // the grid origin is recomputed from the points (hv_cuda_kernel.cu:274-276), the sizes come from grad_grid (:200)
    Checked c = checked_inputs(points, xyz_labels, scale_labels, obj_labels, res, num_rots, !corners.has_value(), st);
    if (corners.has_value()) box_from_corners(*corners, c.mn, c.mx);
    const int dims[3] = {(int)grad_grid.size(0), (int)grad_grid.size(1), (int)grad_grid.size(2)};
    auto d_xyz = torch::empty_like(xyz_labels);
    auto d_scale = torch::empty_like(scale_labels);
    auto d_obj = torch::empty_like(obj_labels);
    cv_check(cv_hv_backward_f32(grad_grid.data_ptr<float>(), points.data_ptr<float>(), xyz_labels.data_ptr<float>(),
                                scale_labels.data_ptr<float>(), obj_labels.data_ptr<float>(), c.n, c.res, c.num_rots, c.mn,
                                dims, d_xyz.data_ptr<float>(), d_scale.data_ptr<float>(), d_obj.data_ptr<float>(), st),
             "cv_hv_backward_f32");
    return {d_xyz, d_scale, d_obj};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "canonical voting on MI355X (gfx950): compiled drop-in for the reference's hv_cuda extension";
    m.def("forward", &hv_forward, "hv forward (HIP, gfx950)", py::arg("points"), py::arg("xyz_labels"), py::arg("scale_labels"),
          py::arg("obj_labels"), py::arg("res"), py::arg("num_rots"), py::arg("corners") = py::none());
    m.def("backward", &hv_backward, "hv backward (HIP, gfx950)", py::arg("grad_grid"), py::arg("points"), py::arg("xyz_labels"),
          py::arg("scale_labels"), py::arg("obj_labels"), py::arg("res"), py::arg("num_rots"), py::arg("corners") = py::none());
    m.def("set_algorithm", [](int a) { g_algo = a; }, "0 auto, 1 direct global atomics, 2 LDS tiles (A/B measurements)");
    // the header this module was compiled against vs the libcvhip.so the loader found: a stale pair must not run
    TORCH_CHECK(cv_abi_version() == CV_ABI_VERSION, "hv_cuda: libcvhip.so has ABI version ", cv_abi_version(),
                ", this extension was compiled against ", CV_ABI_VERSION, " - rebuild (python -m canonicalvoting_amd.csrc.build --force)");
    m.def("abi_version", []() { return CV_ABI_VERSION; }, "CV_ABI_VERSION of the header this extension was compiled against");
    m.def("library_abi_version", []() { return cv_abi_version(); }, "cv_abi_version() of the loaded libcvhip.so");
}
