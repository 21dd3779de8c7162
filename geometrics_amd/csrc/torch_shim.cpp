// The compiled entry points the reference's python wrappers bind: pybind functions with the call shape of
//   cd.forward_cuda(xyz1, xyz2, dist1, dist2, idx1, idx2)            chamfer_distance/chamfer_distance.cpp:15-27, 36-38
//   tri.forward_cuda(xyz1, tri1, tri2, tri3, dist, point, index)     tri_distance/tri_distance.cpp:16-30, 34-36
// on top of the C ABI of libgeom_hip.so.  Caller-allocated outputs are filled in place; unlike the reference glue the
// tensors are validated (device, dtype, contiguity, shapes), the launch goes to torch's CURRENT stream of the input's
// device, and a failed launch raises instead of printing.  Host code only (no kernels): built with g++ against the
// torch headers by geometrics_amd/build.py; geometrics_amd.chamfer_distance / .tri_distance use the ctypes binding by
// default and these functions are what a maintainer of the reference would call from the unmodified wrappers
// (INTEGRATION.md section 3).
#include <torch/extension.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/hip/HIPGraphsC10Utils.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

#include "geom_hip.h"

namespace {

void check_points(const at::Tensor &t, const char *name, at::ScalarType dtype, int64_t dims, int64_t last)
{
    TORCH_CHECK(t.is_cuda(), name, " must live on a HIP device (geometrics_amd has no CPU path)");
    TORCH_CHECK(t.scalar_type() == dtype, name, " has the wrong dtype");
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
    TORCH_CHECK(t.dim() == dims, name, " must be ", dims, "-dimensional");
    TORCH_CHECK(last < 0 || t.size(-1) == last, name, " must have last dimension ", last);
}

void raise_on(int code, const char *what)
{
    TORCH_CHECK(code == 0, what, " failed: ", geom_strerror(code), " (code ", code, ")");
}

// Package-wide reference-quirk mode (SURVEY quirk register Q1/Q3): calls that pass no flags reproduce the shipped CUDA
// kernels' tail truncation (chamfer_distance.cu:31-33, tri_distance.cu:129,134).  Initialised from GEOM_REF_QUIRKS, switched
// at run time by set_reference_quirks() (geometrics_amd.set_reference_quirks() forwards to it).
std::atomic<int> g_reference_quirks{-1};

bool reference_quirks()
{
    int q = g_reference_quirks.load();
    if (q < 0) {
        const char *env = std::getenv("GEOM_REF_QUIRKS");
        q = (env && env[0] && std::strcmp(env, "0") != 0) ? 1 : 0;
        g_reference_quirks.store(q);
    }
    return q == 1;
}

unsigned resolve_flags(int64_t flags)
{
    return flags < 0 ? (reference_quirks() ? GEOM_FLAG_REF_TAIL_TRUNC : 0u) : (unsigned)flags;
}

void chamfer_forward_cuda(at::Tensor xyz1, at::Tensor xyz2, at::Tensor dist1, at::Tensor dist2, at::Tensor idx1,
                          at::Tensor idx2, int64_t flags)
{
    check_points(xyz1, "xyz1", at::kFloat, 3, 3);
    check_points(xyz2, "xyz2", at::kFloat, 3, 3);
    const int64_t b = xyz1.size(0), n = xyz1.size(1), m = xyz2.size(1);
    TORCH_CHECK(xyz2.size(0) == b, "batch sizes differ");
    check_points(dist1, "dist1", at::kFloat, 2, n);
    check_points(dist2, "dist2", at::kFloat, 2, m);
    check_points(idx1, "idx1", at::kInt, 2, n);
    check_points(idx2, "idx2", at::kInt, 2, m);
    TORCH_CHECK(dist1.size(0) == b && dist2.size(0) == b && idx1.size(0) == b && idx2.size(0) == b, "output batch sizes differ");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(xyz1.device());   // PyTorch-ROCm tensors carry device type "cuda"
    raise_on(geom_chamfer_nn_f32((int)b, (int)n, xyz1.data_ptr<float>(), (int)m, xyz2.data_ptr<float>(), dist1.data_ptr<float>(),
                                 idx1.data_ptr<int>(), dist2.data_ptr<float>(), idx2.data_ptr<int>(), resolve_flags(flags),
                                 c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()),
             "geom_chamfer_nn_f32");
}

// Visiting order for the two-level scan when all there is are corner tensors (no face list): one order per (triangle
// count, device), rebuilt every 256 calls on the device from the first mesh's centroids (30-bit Morton curve; a dozen
// small ATen launches, no host synchronisation), never during a stream capture.  A stale order costs speed only: the
// scan's keys carry the original triangle index.  Same policy as geometrics_amd.tri_distance.soup_order.
at::Tensor morton_order(const at::Tensor &centroids)
{
    auto c = at::nan_to_num(centroids, 0.0, 0.0, 0.0);
    auto lo = std::get<0>(c.min(0));
    auto span = (std::get<0>(c.max(0)) - lo).clamp_min(1e-30);
    auto q = ((c - lo) / span * 1023.0).to(at::kLong).clamp(0, 1023);
    auto spread = [](at::Tensor x) {
        x = (x | (x * 65536)) & 0x030000FF;
        x = (x | (x * 256)) & 0x0300F00F;
        x = (x | (x * 16)) & 0x030C30C3;
        return (x | (x * 4)) & 0x09249249;
    };
    auto code = spread(q.select(1, 0)) | (spread(q.select(1, 1)) * 2) | (spread(q.select(1, 2)) * 4);
    return at::argsort(code, /*stable=*/true, 0, false).to(at::kInt).contiguous();
}

struct SoupOrder {
    at::Tensor order;
    int calls = 0;
    bool captured = false;   // handed out during a stream capture: a graph holds its address, never replace it
};

// Returns the tensor BY VALUE (a reference count of its own): a concurrent refresh from another thread cannot free the
// memory under the caller's launch.  The cache itself is guarded by a mutex.
at::Tensor soup_order(const at::Tensor &tri1, const at::Tensor &tri2, const at::Tensor &tri3)
{
    static std::map<std::pair<int64_t, int>, SoupOrder> cache;
    static std::mutex guard;
    const int64_t m = tri1.size(1);
    if (tri1.size(0) == 0 || m < 64) return at::Tensor();
    const bool capturing = c10::hip::currentStreamCaptureStatusMayInitCtx() != c10::hip::CaptureStatus::None;
    {
        std::lock_guard<std::mutex> lock(guard);
        auto &slot = cache[{m, (int)tri1.get_device()}];
        if (slot.order.defined() && (slot.calls < 256 || capturing || slot.captured)) {
            ++slot.calls;
            slot.captured = slot.captured || capturing;
            return slot.order;
        }
    }
    if (capturing) return at::Tensor();
    at::NoGradGuard no_grad;
    at::Tensor fresh = morton_order((tri1[0] + tri2[0] + tri3[0]) * (1.0 / 3.0));
    std::lock_guard<std::mutex> lock(guard);
    auto &slot = cache[{m, (int)tri1.get_device()}];
    if (!slot.captured) {
        slot.order = fresh;
        slot.calls = 1;
    }
    return slot.order;
}

void tri_forward_cuda(at::Tensor xyz1, at::Tensor tri1, at::Tensor tri2, at::Tensor tri3, at::Tensor dist, at::Tensor point,
                      at::Tensor index, int64_t flags)
{
    check_points(xyz1, "xyz1", at::kFloat, 3, 3);
    check_points(tri1, "tri1", at::kFloat, 3, 3);
    check_points(tri2, "tri2", at::kFloat, 3, 3);
    check_points(tri3, "tri3", at::kFloat, 3, 3);
    const int64_t b = xyz1.size(0), n = xyz1.size(1), m = tri1.size(1);
    TORCH_CHECK(tri1.sizes() == tri2.sizes() && tri1.sizes() == tri3.sizes() && tri1.size(0) == b,
                "tri1 / tri2 / tri3 must share one [B,M,3] shape with xyz1's batch");
    check_points(dist, "dist", at::kFloat, 2, n);
    check_points(point, "point", at::kInt, 2, n);
    check_points(index, "index", at::kInt, 2, n);
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(xyz1.device());   // PyTorch-ROCm tensors carry device type "cuda"
    // the reference launcher's arguments (tri_distance.cpp:4-13) + what the fast scan needs: a visiting order (cached, see
    // above) and a scratch tensor for the per-triangle records (torch's caching allocator: free after the first call,
    // capture-safe)
    const at::Tensor order = soup_order(tri1, tri2, tri3);
    const size_t ws_bytes = geom_tri_distance_workspace_bytes((int)b, (int)n, (int)m);
    at::Tensor ws = at::empty({(int64_t)(ws_bytes / 4 + 4)}, xyz1.options());
    raise_on(geom_tri_distance_ws_f32((int)b, (int)n, xyz1.data_ptr<float>(), (int)m, tri1.data_ptr<float>(), tri2.data_ptr<float>(),
                                      tri3.data_ptr<float>(), order.defined() ? order.data_ptr<int>() : nullptr, dist.data_ptr<float>(),
                                      point.data_ptr<int>(), index.data_ptr<int>(), resolve_flags(flags), ws.data_ptr<float>(), ws_bytes,
                                      c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()),
             "geom_tri_distance_ws_f32");
}

} // namespace

PYBIND11_MODULE(geom_torch_shim, m)
{
    m.def("chamfer_forward_cuda", &chamfer_forward_cuda, "chamfer_distance.cpp:36-38 forward_cuda on libgeom_hip.so",
          py::arg("xyz1"), py::arg("xyz2"), py::arg("dist1"), py::arg("dist2"), py::arg("idx1"), py::arg("idx2"), py::arg("flags") = -1);
    m.def("tri_forward_cuda", &tri_forward_cuda, "tri_distance.cpp:34-36 forward_cuda on libgeom_hip.so", py::arg("xyz1"),
          py::arg("tri1"), py::arg("tri2"), py::arg("tri3"), py::arg("dist"), py::arg("point"), py::arg("index"), py::arg("flags") = -1);
    m.def("set_reference_quirks", [](bool on) { g_reference_quirks.store(on ? 1 : 0); },
          "calls without flags reproduce the shipped CUDA kernels' tail truncation (GEOM_FLAG_REF_TAIL_TRUNC)");
    m.def("reference_quirks", &reference_quirks);
}
