// PyTorch-ROCm extension module `_C`: the operator surface the reference's Python wrapper binds
// (submodules/diff-gaussian-rasterization/ext.cpp:15-20), implemented on top of the C ABI in
// include/grpg_rasterizer.h.  Same four functions, same argument order and return tuples as
// rasterize_points.h:18-88 / rasterize_points.cu:35-306:
//   rasterize_gaussians, rasterize_gaussians_backward, mark_visible, rasterize_gaussians_filter.
// This file is plumbing only (tensor allocation, pointer extraction, stream/device selection);
// it contains no rasterizer arithmetic and has no CPU path: CPU means3D -> error.
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <torch/extension.h>

#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/grpg_rasterizer.h"

namespace {

// Optional inputs arrive as EMPTY (usually CPU) tensors, diff_gaussian_rasterization/__init__.py:
// 207-217; the reference turns them into null data pointers.  Everything else must be fp32 on
// the same HIP device as means3D.
const float* fptr(const torch::Tensor& t, const torch::Tensor& like, const char* name,
                  torch::Tensor& keep) {
  if (!t.defined() || t.numel() == 0) return nullptr;
  TORCH_CHECK(t.scalar_type() == torch::kFloat32, name, " must be float32");
  TORCH_CHECK(t.device() == like.device(), name, " must be on ", like.device(), " (got ",
              t.device(), ")");
  keep = t.contiguous();
  return keep.data_ptr<float>();
}

char* resize_blob(size_t bytes, void* user) {   // replaces resizeFunctional, rasterize_points.cu:27-33
  auto* t = static_cast<torch::Tensor*>(user);
  // num_rendered changes a little from frame to frame; round the request up to 1/8-octave steps so
  // the caching allocator keeps handing back the same blocks instead of growing its pools
  // (a pool growth is a hipMalloc, i.e. a device-wide sync in the middle of the frame loop)
  if (bytes > (1u << 20)) {
    size_t step = 1;
    while ((step << 4) <= bytes) step <<= 1;   // step = 2^floor(log2(bytes)) / 8
    bytes = (bytes + step - 1) / step * step;
  }
  t->resize_({static_cast<long long>(bytes)});
  return reinterpret_cast<char*>(t->data_ptr());
}

void require_device(const torch::Tensor& means3D) {
  TORCH_CHECK(means3D.is_cuda(),
              "gaussianrpg_amd: means3D must live on a ROCm/HIP device (torch device 'cuda'); "
              "this rasterizer is MI355X-native and has no CPU path");
}

[[noreturn]] void raise_abi_error(const char* what, int rc) {
  throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) +
                           "): " + grpg_last_error());
}

}  // namespace

std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansImpl(const unsigned flags, int* ticket /* non-NULL: deferred frame */,
                   const torch::Tensor& background, const torch::Tensor& means3D,
                   const torch::Tensor& colors, const torch::Tensor& semantics,
                   const torch::Tensor& opacity, const torch::Tensor& scales,
                   const torch::Tensor& rotations, const float scale_modifier,
                   const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                   const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                   const int image_height, const int image_width, const torch::Tensor& sh,
                   const int degree, const torch::Tensor& campos, const bool prefiltered,
                   const bool debug) {
  if (means3D.ndimension() != 2 || means3D.size(1) != 3) {
    AT_ERROR("means3D must have dimensions (num_points, 3)");   // rasterize_points.cu:58-60
  }
  require_device(means3D);
  TORCH_CHECK(means3D.scalar_type() == torch::kFloat32, "means3D must be float32");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D.device());
  const int P = means3D.size(0);
  const int H = image_height, W = image_width;
  TORCH_CHECK(semantics.dim() == 2, "semantics must be [P,S]");
  const int S = semantics.size(1);

  auto float_opts = means3D.options().dtype(torch::kFloat32);
  // every element of these planes is written by the library: no zero-fill pass needed
  torch::Tensor out_color = torch::empty({GRPG_NUM_CHANNELS, H, W}, float_opts);
  torch::Tensor out_depth = torch::empty({1, H, W}, float_opts);
  torch::Tensor out_alpha = torch::empty({1, H, W}, float_opts);
  torch::Tensor out_semantic = torch::empty({S, H, W}, float_opts);
  torch::Tensor radii = torch::empty({P}, means3D.options().dtype(torch::kInt32));
  auto byte_opts = means3D.options().dtype(torch::kByte);
  torch::Tensor geomBuffer = torch::empty({0}, byte_opts);
  torch::Tensor binningBuffer = torch::empty({0}, byte_opts);
  torch::Tensor imgBuffer = torch::empty({0}, byte_opts);

  int M = 0;
  if (sh.size(0) != 0) M = sh.size(1);   // rasterize_points.cu:88-92

  torch::Tensor k_bg, k_means, k_sh, k_col, k_sem, k_op, k_sc, k_rot, k_cov, k_view, k_proj, k_cam;
  const float* p_bg = fptr(background, means3D, "background", k_bg);
  const float* p_means = fptr(means3D, means3D, "means3D", k_means);
  const float* p_sh = fptr(sh, means3D, "sh", k_sh);
  const float* p_col = fptr(colors, means3D, "colors_precomp", k_col);
  const float* p_sem = fptr(semantics, means3D, "semantics", k_sem);
  const float* p_op = fptr(opacity, means3D, "opacities", k_op);
  const float* p_sc = fptr(scales, means3D, "scales", k_sc);
  const float* p_rot = fptr(rotations, means3D, "rotations", k_rot);
  const float* p_cov = fptr(cov3D_precomp, means3D, "cov3D_precomp", k_cov);
  const float* p_view = fptr(viewmatrix, means3D, "viewmatrix", k_view);
  const float* p_proj = fptr(projmatrix, means3D, "projmatrix", k_proj);
  const float* p_cam = fptr(campos, means3D, "campos", k_cam);
  TORCH_CHECK(p_bg && p_view && p_proj && p_cam, "bg/viewmatrix/projmatrix/campos must be non-empty");

  hipStream_t stream = at::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  float* p_out_color = out_color.data_ptr<float>();
  float* p_out_depth = out_depth.data_ptr<float>();
  float* p_out_alpha = out_alpha.data_ptr<float>();
  float* p_out_sem = S > 0 ? out_semantic.data_ptr<float>() : nullptr;
  int* p_radii = P > 0 ? radii.data_ptr<int>() : nullptr;
  int rendered;
  {
    // The call blocks once on the stream (num_rendered read-back): let other Python threads drive
    // their own streams meanwhile.  The blob callbacks only touch ATen, never Python objects.
    pybind11::gil_scoped_release nogil;
    if (ticket)
      rendered = grpg_forward_deferred(
          resize_blob, &geomBuffer, resize_blob, &binningBuffer, resize_blob, &imgBuffer, P, degree,
          M, S, p_bg, W, H, p_means, p_sh, p_col, p_sem, p_op, p_sc, scale_modifier, p_rot, p_cov,
          p_view, p_proj, p_cam, tan_fovx, tan_fovy, prefiltered ? 1 : 0, p_out_color, p_out_depth,
          p_out_alpha, p_out_sem, p_radii, debug ? 1 : 0, (void*)stream, flags, ticket);
    else
      rendered = grpg_forward_flags(
          resize_blob, &geomBuffer, resize_blob, &binningBuffer, resize_blob, &imgBuffer, P, degree,
          M, S, p_bg, W, H, p_means, p_sh, p_col, p_sem, p_op, p_sc, scale_modifier, p_rot, p_cov,
          p_view, p_proj, p_cam, tan_fovx, tan_fovy, prefiltered ? 1 : 0, p_out_color, p_out_depth,
          p_out_alpha, p_out_sem, p_radii, debug ? 1 : 0, (void*)stream, flags);
  }
  if (rendered < 0) raise_abi_error("grpg_forward", rendered);
  return std::make_tuple(rendered, out_color, out_depth, out_alpha, out_semantic, radii,
                         geomBuffer, binningBuffer, imgBuffer);
}

#define GRPG_RASTERIZE_ARGS                                                                        \
  const torch::Tensor &background, const torch::Tensor &means3D, const torch::Tensor &colors,      \
      const torch::Tensor &semantics, const torch::Tensor &opacity, const torch::Tensor &scales,   \
      const torch::Tensor &rotations, const float scale_modifier, const torch::Tensor &cov3D_precomp, \
      const torch::Tensor &viewmatrix, const torch::Tensor &projmatrix, const float tan_fovx,      \
      const float tan_fovy, const int image_height, const int image_width, const torch::Tensor &sh, \
      const int degree, const torch::Tensor &campos, const bool prefiltered, const bool debug
#define GRPG_RASTERIZE_PASS                                                                        \
  background, means3D, colors, semantics, opacity, scales, rotations, scale_modifier, cov3D_precomp, \
      viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,   \
      prefiltered, debug

// the reference's entry point (rasterize_points.cu:35-124): blobs valid for the backward
auto RasterizeGaussians(GRPG_RASTERIZE_ARGS) { return RasterizeGaussiansImpl(0u, nullptr, GRPG_RASTERIZE_PASS); }
// same call when no backward can follow (additive): n_contrib is not produced
auto RasterizeGaussiansEval(GRPG_RASTERIZE_ARGS) {
  return RasterizeGaussiansImpl(GRPG_FORWARD_NO_BACKWARD, nullptr, GRPG_RASTERIZE_PASS);
}
// ... and without the per-frame wait for num_rendered (additive; grpg_forward_deferred): returns
// (ticket, color, depth, alpha, semantic, radii); the outputs may be consumed by further work on
// the stream, but are only KNOWN good once frame_status(ticket) says so
auto RasterizeGaussiansEvalDeferred(GRPG_RASTERIZE_ARGS) {
  int ticket = -1;
  auto r = RasterizeGaussiansImpl(GRPG_FORWARD_NO_BACKWARD, &ticket, GRPG_RASTERIZE_PASS);
  return std::make_tuple(ticket, std::get<1>(r), std::get<2>(r), std::get<3>(r), std::get<4>(r),
                         std::get<5>(r));
}
// Layered frame (additive; grpg_forward_layers): the composition + the background-only and objects-only
// planes of StreetGaussianRenderer.render_all from one pass.  layer_class: uint8 / bool [P], != 0 = object.
// returns (num_rendered, color, depth, alpha, radii, color_bg, alpha_bg, color_obj, alpha_obj)
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor, torch::Tensor>
RasterizeGaussiansLayers(const torch::Tensor& background, const torch::Tensor& layer_background,
                         const torch::Tensor& layer_class, const torch::Tensor& means3D,
                         const torch::Tensor& colors, const torch::Tensor& opacity, const torch::Tensor& scales,
                         const torch::Tensor& rotations, const float scale_modifier,
                         const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                         const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                         const int image_height, const int image_width, const torch::Tensor& sh,
                         const int degree, const torch::Tensor& campos, const bool debug) {
  if (means3D.ndimension() != 2 || means3D.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");
  require_device(means3D);
  TORCH_CHECK(means3D.scalar_type() == torch::kFloat32, "means3D must be float32");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D.device());
  const int P = means3D.size(0);
  const int H = image_height, W = image_width;
  TORCH_CHECK(layer_class.numel() == P && layer_class.device() == means3D.device() &&
                  (layer_class.scalar_type() == torch::kUInt8 || layer_class.scalar_type() == torch::kBool),
              "layer_class must be a uint8 / bool tensor of P elements on the device of means3D");
  const torch::Tensor k_cls = layer_class.contiguous();
  auto fo = means3D.options().dtype(torch::kFloat32);
  torch::Tensor out_color = torch::empty({GRPG_NUM_CHANNELS, H, W}, fo), out_depth = torch::empty({1, H, W}, fo);
  torch::Tensor out_alpha = torch::empty({1, H, W}, fo);
  torch::Tensor color_bg = torch::empty({GRPG_NUM_CHANNELS, H, W}, fo), alpha_bg = torch::empty({1, H, W}, fo);
  torch::Tensor color_obj = torch::empty({GRPG_NUM_CHANNELS, H, W}, fo), alpha_obj = torch::empty({1, H, W}, fo);
  torch::Tensor radii = torch::empty({P}, means3D.options().dtype(torch::kInt32));
  auto byte_opts = means3D.options().dtype(torch::kByte);
  torch::Tensor geomBuffer = torch::empty({0}, byte_opts), binningBuffer = torch::empty({0}, byte_opts);
  torch::Tensor imgBuffer = torch::empty({0}, byte_opts);
  int M = 0;
  if (sh.size(0) != 0) M = sh.size(1);
  torch::Tensor k_bg, k_lbg, k_means, k_sh, k_col, k_op, k_sc, k_rot, k_cov, k_view, k_proj, k_cam;
  const float* p_bg = fptr(background, means3D, "background", k_bg);
  const float* p_lbg = fptr(layer_background, means3D, "layer_background", k_lbg);
  const float* p_means = fptr(means3D, means3D, "means3D", k_means);
  const float* p_sh = fptr(sh, means3D, "sh", k_sh);
  const float* p_col = fptr(colors, means3D, "colors_precomp", k_col);
  const float* p_op = fptr(opacity, means3D, "opacities", k_op);
  const float* p_sc = fptr(scales, means3D, "scales", k_sc);
  const float* p_rot = fptr(rotations, means3D, "rotations", k_rot);
  const float* p_cov = fptr(cov3D_precomp, means3D, "cov3D_precomp", k_cov);
  const float* p_view = fptr(viewmatrix, means3D, "viewmatrix", k_view);
  const float* p_proj = fptr(projmatrix, means3D, "projmatrix", k_proj);
  const float* p_cam = fptr(campos, means3D, "campos", k_cam);
  TORCH_CHECK(p_bg && p_lbg && p_view && p_proj && p_cam, "bg/layer_bg/viewmatrix/projmatrix/campos must be non-empty");
  hipStream_t stream = at::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  int rendered;
  {
    pybind11::gil_scoped_release nogil;
    rendered = grpg_forward_layers(
        resize_blob, &geomBuffer, resize_blob, &binningBuffer, resize_blob, &imgBuffer, P, degree, M, p_bg, W, H,
        p_means, p_sh, p_col, p_op, p_sc, scale_modifier, p_rot, p_cov, p_view, p_proj, p_cam, tan_fovx, tan_fovy,
        P > 0 ? (const unsigned char*)k_cls.data_ptr() : nullptr, p_lbg, out_color.data_ptr<float>(),
        out_depth.data_ptr<float>(), out_alpha.data_ptr<float>(), color_bg.data_ptr<float>(),
        alpha_bg.data_ptr<float>(), color_obj.data_ptr<float>(), alpha_obj.data_ptr<float>(),
        P > 0 ? radii.data_ptr<int>() : nullptr, debug ? 1 : 0, (void*)stream);
  }
  if (rendered < 0) raise_abi_error("grpg_forward_layers", rendered);
  return std::make_tuple(rendered, out_color, out_depth, out_alpha, radii, color_bg, alpha_bg, color_obj, alpha_obj);
}

// ---- the frame epilogue (grpg_forward_frame / grpg_forward_composed_frame, ABI 6; host destination: ABI 7) ----
// sky_cube: [6,res,res,3] float32 on the device, or an empty tensor (no sky composite); ray_matrix: 9 floats, CPU
// (taken by value) or device (read by the kernel: no host round trip).
struct EpilogueArgs {
  grpg_frame_epilogue e;
  torch::Tensor k_cube, k_rm, rgb8;
};
static void make_epilogue(EpilogueArgs& a, const torch::Tensor& like, const torch::Tensor& sky_cube,
                          const torch::Tensor& ray_matrix, const float sky_fill, const bool clamp, const bool want_rgb8,
                          const bool truncate, const c10::optional<torch::Tensor>& out_rgb8, const int H, const int W) {
  a.e = grpg_frame_epilogue{nullptr, 0, nullptr, 0, sky_fill, clamp ? 1 : 0, nullptr, truncate ? 1 : 0, 0};
  if (sky_cube.defined() && sky_cube.numel() != 0) {
    TORCH_CHECK(sky_cube.dim() == 4 && sky_cube.size(0) == 6 && sky_cube.size(1) == sky_cube.size(2) &&
                    sky_cube.size(3) == 3 && sky_cube.scalar_type() == torch::kFloat32 && sky_cube.device() == like.device(),
                "sky_cube must be a float32 [6,res,res,3] tensor on the frame's device");
    TORCH_CHECK(ray_matrix.defined() && ray_matrix.numel() == 9 && ray_matrix.scalar_type() == torch::kFloat32,
                "ray_matrix must hold 9 float32 values (row-major R^T K^-1)");
    a.k_cube = sky_cube.contiguous();
    a.k_rm = ray_matrix.contiguous();
    if (a.k_rm.is_cuda()) TORCH_CHECK(a.k_rm.device() == like.device(), "ray_matrix on another device");
    a.e.sky_cube = a.k_cube.data_ptr<float>();
    a.e.sky_res = (int)a.k_cube.size(1);
    a.e.ray_matrix = a.k_rm.data_ptr<float>();
    a.e.ray_matrix_on_device = a.k_rm.is_cuda() ? 1 : 0;
  }
  if (want_rgb8) {
    if (out_rgb8.has_value() && out_rgb8->defined()) {
      // a PINNED host tensor is a destination too (ABI 7: the epilogue stores the bytes through the link while the
      // render runs; the caller synchronises with the stream before reading, as after a non_blocking copy)
      const bool on_host = out_rgb8->is_cpu();
      TORCH_CHECK(out_rgb8->scalar_type() == torch::kByte && out_rgb8->numel() == (int64_t)3 * H * W &&
                      out_rgb8->is_contiguous() &&
                      (on_host ? out_rgb8->is_pinned() : out_rgb8->device() == like.device()),
                  "out must be a contiguous uint8 tensor of H*W*3 elements on the frame's device, or in pinned host "
                  "memory (Tensor.pin_memory())");
      a.rgb8 = *out_rgb8;
      a.e.out_rgb8_on_host = on_host ? 1 : 0;
    } else {
      a.rgb8 = torch::empty({H, W, 3}, like.options().dtype(torch::kByte));
    }
    a.e.out_rgb8 = a.rgb8.data_ptr<unsigned char>();
  } else {
    a.rgb8 = torch::empty({0}, like.options().dtype(torch::kByte));
  }
}

// returns (num_rendered, rgb8, color, depth, alpha, radii, color_bg, alpha_bg, color_obj, alpha_obj); tensors that
// were not asked for are empty
typedef std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
                   torch::Tensor, torch::Tensor, torch::Tensor> FrameResult;

FrameResult RasterizeGaussiansFrame(
    const torch::Tensor& background, const torch::Tensor& layer_background, const torch::Tensor& layer_class,
    const torch::Tensor& means3D, const torch::Tensor& colors, const torch::Tensor& opacity,
    const torch::Tensor& scales, const torch::Tensor& rotations, const float scale_modifier,
    const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix,
    const float tan_fovx, const float tan_fovy, const int image_height, const int image_width,
    const torch::Tensor& sh, const int degree, const torch::Tensor& campos, const bool debug,
    const torch::Tensor& sky_cube, const torch::Tensor& ray_matrix, const float sky_fill, const bool clamp,
    const bool want_planes, const bool want_rgb8, const bool truncate, const c10::optional<torch::Tensor>& out_rgb8) {
  if (means3D.ndimension() != 2 || means3D.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");
  require_device(means3D);
  TORCH_CHECK(means3D.scalar_type() == torch::kFloat32, "means3D must be float32");
  TORCH_CHECK(want_planes || want_rgb8, "forward_frame: ask for the float planes, the rgb8 frame, or both");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D.device());
  const int P = means3D.size(0);
  const int H = image_height, W = image_width;
  const bool layered = layer_class.defined() && layer_class.numel() != 0;
  torch::Tensor k_cls;
  if (layered) {
    TORCH_CHECK(layer_class.numel() == P && layer_class.device() == means3D.device() &&
                    (layer_class.scalar_type() == torch::kUInt8 || layer_class.scalar_type() == torch::kBool),
                "layer_class must be a uint8 / bool tensor of P elements on the device of means3D");
    k_cls = layer_class.contiguous();
  }
  auto fo = means3D.options().dtype(torch::kFloat32);
  auto none = [&]() { return torch::empty({0}, fo); };
  torch::Tensor out_color = want_planes ? torch::empty({GRPG_NUM_CHANNELS, H, W}, fo) : none();
  torch::Tensor out_depth = want_planes ? torch::empty({1, H, W}, fo) : none();
  torch::Tensor out_alpha = want_planes ? torch::empty({1, H, W}, fo) : none();
  torch::Tensor color_bg = layered ? torch::empty({GRPG_NUM_CHANNELS, H, W}, fo) : none();
  torch::Tensor alpha_bg = layered ? torch::empty({1, H, W}, fo) : none();
  torch::Tensor color_obj = layered ? torch::empty({GRPG_NUM_CHANNELS, H, W}, fo) : none();
  torch::Tensor alpha_obj = layered ? torch::empty({1, H, W}, fo) : none();
  torch::Tensor radii = torch::empty({P}, means3D.options().dtype(torch::kInt32));
  auto byte_opts = means3D.options().dtype(torch::kByte);
  torch::Tensor geomBuffer = torch::empty({0}, byte_opts), binningBuffer = torch::empty({0}, byte_opts);
  torch::Tensor imgBuffer = torch::empty({0}, byte_opts);
  int M = 0;
  if (sh.size(0) != 0) M = sh.size(1);
  torch::Tensor k_bg, k_lbg, k_means, k_sh, k_col, k_op, k_sc, k_rot, k_cov, k_view, k_proj, k_cam;
  const float* p_bg = fptr(background, means3D, "background", k_bg);
  const float* p_lbg = fptr(layer_background, means3D, "layer_background", k_lbg);
  const float* p_means = fptr(means3D, means3D, "means3D", k_means);
  const float* p_sh = fptr(sh, means3D, "sh", k_sh);
  const float* p_col = fptr(colors, means3D, "colors_precomp", k_col);
  const float* p_op = fptr(opacity, means3D, "opacities", k_op);
  const float* p_sc = fptr(scales, means3D, "scales", k_sc);
  const float* p_rot = fptr(rotations, means3D, "rotations", k_rot);
  const float* p_cov = fptr(cov3D_precomp, means3D, "cov3D_precomp", k_cov);
  const float* p_view = fptr(viewmatrix, means3D, "viewmatrix", k_view);
  const float* p_proj = fptr(projmatrix, means3D, "projmatrix", k_proj);
  const float* p_cam = fptr(campos, means3D, "campos", k_cam);
  TORCH_CHECK(p_bg && p_view && p_proj && p_cam, "bg/viewmatrix/projmatrix/campos must be non-empty");
  TORCH_CHECK(!layered || p_lbg, "layer_background must be non-empty in a layered frame");
  EpilogueArgs ea;
  make_epilogue(ea, means3D, sky_cube, ray_matrix, sky_fill, clamp, want_rgb8, truncate, out_rgb8, H, W);
  hipStream_t stream = at::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  int rendered;
  {
    pybind11::gil_scoped_release nogil;
    rendered = grpg_forward_frame(
        resize_blob, &geomBuffer, resize_blob, &binningBuffer, resize_blob, &imgBuffer, P, degree, M, p_bg, W, H,
        p_means, p_sh, p_col, p_op, p_sc, scale_modifier, p_rot, p_cov, p_view, p_proj, p_cam, tan_fovx, tan_fovy,
        (layered && P > 0) ? (const unsigned char*)k_cls.data_ptr() : nullptr, p_lbg,
        want_planes ? out_color.data_ptr<float>() : nullptr, want_planes ? out_depth.data_ptr<float>() : nullptr,
        want_planes ? out_alpha.data_ptr<float>() : nullptr, layered ? color_bg.data_ptr<float>() : nullptr,
        layered ? alpha_bg.data_ptr<float>() : nullptr, layered ? color_obj.data_ptr<float>() : nullptr,
        layered ? alpha_obj.data_ptr<float>() : nullptr, P > 0 ? radii.data_ptr<int>() : nullptr, debug ? 1 : 0,
        (void*)stream, &ea.e);
  }
  if (rendered < 0) raise_abi_error("grpg_forward_frame", rendered);
  return std::make_tuple(rendered, ea.rgb8, out_color, out_depth, out_alpha, radii, color_bg, alpha_bg, color_obj,
                         alpha_obj);
}

// (ok, num_rendered): ok = 1 valid, 0 the frame must be rendered again, -1 not ready (wait = false)
std::tuple<int, int> FrameStatus(const int ticket, const bool wait) {
  int R = 0, rc;
  {
    pybind11::gil_scoped_release nogil;
    rc = grpg_frame_status(ticket, wait ? 1 : 0, &R);
  }
  if (rc == GRPG_OK) return std::make_tuple(1, R);
  if (rc == GRPG_ERR_CAPACITY) return std::make_tuple(0, 0);
  if (rc == GRPG_ERR_NOT_READY) return std::make_tuple(-1, 0);
  raise_abi_error("grpg_frame_status", rc);
  return std::make_tuple(0, 0);
}

// full_intermediates: the reference's binding ALWAYS returns dL_dcolors [P,3] and dL_dcov3D [P,6]
// (rasterize_points.cu:166-176,219) -- the per-Gaussian colour gradient and the 3D-covariance
// gradient, real values even when the colours came from SH / the covariance from scale + rotation.
// _C.rasterize_gaussians_backward (the reference's name) does the same.  The autograd Function
// never looks at them unless the corresponding optional input was given, so it calls
// _C.rasterize_gaussians_backward_lean, which does not materialise the unused ones ([0,3] / [0,6]).
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansBackwardImpl(const torch::Tensor& background, const torch::Tensor& means3D,
                           const torch::Tensor& radii, const torch::Tensor& colors,
                           const torch::Tensor& scales, const torch::Tensor& rotations,
                           const float scale_modifier, const torch::Tensor& cov3D_precomp,
                           const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix,
                           const float tan_fovx, const float tan_fovy,
                           const torch::Tensor& dL_dout_color, const torch::Tensor& dL_dout_depth,
                           const torch::Tensor& dL_dout_alpha,
                           const torch::Tensor& dL_dout_semantic, const torch::Tensor& sh,
                           const int degree, const torch::Tensor& campos,
                           const torch::Tensor& geomBuffer, const int R,
                           const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer,
                           const torch::Tensor& alphas, const torch::Tensor& semantics,
                           const bool debug, const bool full_intermediates) {
  require_device(means3D);
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D.device());
  const int P = means3D.size(0);
  const int H = dL_dout_color.size(1);   // rasterize_points.cu:156-158
  const int W = dL_dout_color.size(2);
  const int S = dL_dout_semantic.size(0);
  TORCH_CHECK(S <= GRPG_MAX_SEMANTIC_BACKWARD, "rasterize_gaussians_backward supports at most ",
              GRPG_MAX_SEMANTIC_BACKWARD, " semantic channels, got ", S);
  // rasterize_points.cu:159-163 takes M = 0 when sh has no rows; an EMPTY model (P = 0) that still
  // carries shs of shape [0, M, 3] then gets a [0, 0, 3] gradient autograd rejects -- keep M
  int M = 0;
  if (sh.dim() == 3) M = sh.size(1);

  auto o = means3D.options();
  // The reference zero-fills eleven gradient arrays per call (rasterize_points.cu:166-176) because
  // its kernels accumulate into them.  Here the blend backward accumulates into per-Gaussian
  // records inside the geometry blob and the preprocess backward WRITES every element of the
  // non-semantic arrays (zeros for culled Gaussians): they are allocated uninitialised, each as a
  // tensor of its own (autograd's AccumulateGrad can then adopt them as .grad without a copy, which
  // it cannot do with views into a pool -- 92 MB of clones per step at P = 1 M); only dL_dsemantic
  // (float atomics straight into it) is zero-filled.  Gradients of absent optional inputs
  // (colors_precomp, cov3D_precomp) and the two pure intermediates of the reference's binding
  // (dL_dconic, dL_ddepths) are not materialised at all: the library takes NULL for them.
  const bool want_colors = full_intermediates || colors.numel() != 0;
  const bool want_cov = full_intermediates || cov3D_precomp.numel() != 0;
  torch::Tensor dL_dmeans3D = torch::empty({P, 3}, o);
  torch::Tensor dL_dmeans2D = torch::empty({P, 3}, o);
  torch::Tensor dL_dcolors = torch::empty({want_colors ? P : 0, GRPG_NUM_CHANNELS}, o);
  torch::Tensor dL_dopacity = torch::empty({P, 1}, o);
  torch::Tensor dL_dcov3D = torch::empty({want_cov ? P : 0, 6}, o);
  torch::Tensor dL_dsh = torch::empty({P, M, 3}, o);
  torch::Tensor dL_dscales = torch::empty({P, 3}, o);
  torch::Tensor dL_drotations = torch::empty({P, 4}, o);
  torch::Tensor dL_dsemantic = torch::zeros({P, S}, o);

  if (P != 0) {
    torch::Tensor k[16];
    const float* p_bg = fptr(background, means3D, "background", k[0]);
    const float* p_means = fptr(means3D, means3D, "means3D", k[1]);
    const float* p_sh = fptr(sh, means3D, "sh", k[2]);
    const float* p_col = fptr(colors, means3D, "colors_precomp", k[3]);
    const float* p_sem = fptr(semantics, means3D, "semantics", k[4]);
    const float* p_alpha = fptr(alphas, means3D, "alphas", k[5]);
    const float* p_sc = fptr(scales, means3D, "scales", k[6]);
    const float* p_rot = fptr(rotations, means3D, "rotations", k[7]);
    const float* p_cov = fptr(cov3D_precomp, means3D, "cov3D_precomp", k[8]);
    const float* p_view = fptr(viewmatrix, means3D, "viewmatrix", k[9]);
    const float* p_proj = fptr(projmatrix, means3D, "projmatrix", k[10]);
    const float* p_cam = fptr(campos, means3D, "campos", k[11]);
    const float* g_col = fptr(dL_dout_color, means3D, "dL_dout_color", k[12]);
    const float* g_dep = fptr(dL_dout_depth, means3D, "dL_dout_depth", k[13]);
    const float* g_alp = fptr(dL_dout_alpha, means3D, "dL_dout_alpha", k[14]);
    const float* g_sem = fptr(dL_dout_semantic, means3D, "dL_dout_semantic", k[15]);
    TORCH_CHECK(radii.scalar_type() == torch::kInt32 && radii.device() == means3D.device(),
                "radii must be int32 on the device");
    torch::Tensor radii_c = radii.contiguous();
    torch::Tensor gb = geomBuffer.contiguous(), bb = binningBuffer.contiguous(),
                  ib = imageBuffer.contiguous();
    hipStream_t stream = at::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
    const int rc = grpg_backward(
        P, degree, M, R, S, p_bg, W, H, p_means, p_sh, p_col, p_sem, p_alpha, p_sc, scale_modifier,
        p_rot, p_cov, p_view, p_proj, p_cam, tan_fovx, tan_fovy, radii_c.data_ptr<int>(),
        reinterpret_cast<char*>(gb.data_ptr()), reinterpret_cast<char*>(bb.data_ptr()),
        reinterpret_cast<char*>(ib.data_ptr()), g_col, g_dep, g_alp, g_sem,
        dL_dmeans2D.data_ptr<float>(), /*dL_dconic=*/nullptr, dL_dopacity.data_ptr<float>(),
        want_colors ? dL_dcolors.data_ptr<float>() : nullptr, /*dL_ddepth=*/nullptr,
        dL_dmeans3D.data_ptr<float>(), want_cov ? dL_dcov3D.data_ptr<float>() : nullptr,
        M > 0 ? dL_dsh.data_ptr<float>() : nullptr,
        dL_dscales.data_ptr<float>(), dL_drotations.data_ptr<float>(),
        S > 0 ? dL_dsemantic.data_ptr<float>() : nullptr, debug ? 1 : 0, (void*)stream);
    if (rc != GRPG_OK) raise_abi_error("grpg_backward", rc);
  }
  return std::make_tuple(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh,
                         dL_dscales, dL_drotations, dL_dsemantic);
}

#define GRPG_BWD_PARAMS                                                                          \
  const torch::Tensor &background, const torch::Tensor &means3D, const torch::Tensor &radii,     \
      const torch::Tensor &colors, const torch::Tensor &scales, const torch::Tensor &rotations,  \
      const float scale_modifier, const torch::Tensor &cov3D_precomp,                            \
      const torch::Tensor &viewmatrix, const torch::Tensor &projmatrix, const float tan_fovx,    \
      const float tan_fovy, const torch::Tensor &dL_dout_color,                                  \
      const torch::Tensor &dL_dout_depth, const torch::Tensor &dL_dout_alpha,                    \
      const torch::Tensor &dL_dout_semantic, const torch::Tensor &sh, const int degree,          \
      const torch::Tensor &campos, const torch::Tensor &geomBuffer, const int R,                 \
      const torch::Tensor &binningBuffer, const torch::Tensor &imageBuffer,                      \
      const torch::Tensor &alphas, const torch::Tensor &semantics, const bool debug
#define GRPG_BWD_ARGS                                                                            \
  background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,           \
      viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth, dL_dout_alpha,    \
      dL_dout_semantic, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, alphas,    \
      semantics, debug
// the reference's entry point: shapes exactly as rasterize_points.cu:166-176,219
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansBackward(GRPG_BWD_PARAMS) { return RasterizeGaussiansBackwardImpl(GRPG_BWD_ARGS, true); }
// additive: what the autograd Function calls
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansBackwardLean(GRPG_BWD_PARAMS) { return RasterizeGaussiansBackwardImpl(GRPG_BWD_ARGS, false); }

torch::Tensor markVisible(torch::Tensor& means3D, torch::Tensor& viewmatrix,
                          torch::Tensor& projmatrix) {
  require_device(means3D);
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D.device());
  const int P = means3D.size(0);
  torch::Tensor present = torch::full({P}, false, means3D.options().dtype(at::kBool));
  if (P != 0) {
    torch::Tensor k[3];
    const float* p_means = fptr(means3D, means3D, "means3D", k[0]);
    const float* p_view = fptr(viewmatrix, means3D, "viewmatrix", k[1]);
    const float* p_proj = fptr(projmatrix, means3D, "projmatrix", k[2]);
    hipStream_t stream = at::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
    const int rc = grpg_mark_visible(P, p_means, p_view, p_proj,
                                     reinterpret_cast<unsigned char*>(present.data_ptr<bool>()),
                                     (void*)stream);
    if (rc != GRPG_OK) raise_abi_error("grpg_mark_visible", rc);
  }
  return present;
}

std::tuple<torch::Tensor, torch::Tensor> RasterizeGaussiansFilter(
    const torch::Tensor& means3D, const torch::Tensor& scales, const torch::Tensor& rotations,
    const float scale_modifier, const torch::Tensor& cov3D_precomp,
    const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx,
    const float tan_fovy, const int image_height, const int image_width, const bool prefiltered,
    const bool debug) {
  if (means3D.ndimension() != 2 || means3D.size(1) != 3) {
    AT_ERROR("means3D must have dimensions (num_points, 3)");
  }
  require_device(means3D);
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D.device());
  const int P = means3D.size(0);
  torch::Tensor radii = torch::full({P}, 0, means3D.options().dtype(torch::kInt32));
  torch::Tensor means2D = torch::full({P, 2}, 0, means3D.options());
  if (P != 0) {
    torch::Tensor k[6];
    const float* p_means = fptr(means3D, means3D, "means3D", k[0]);
    const float* p_sc = fptr(scales, means3D, "scales", k[1]);
    const float* p_rot = fptr(rotations, means3D, "rotations", k[2]);
    const float* p_cov = fptr(cov3D_precomp, means3D, "cov3D_precomp", k[3]);
    const float* p_view = fptr(viewmatrix, means3D, "viewmatrix", k[4]);
    const float* p_proj = fptr(projmatrix, means3D, "projmatrix", k[5]);
    hipStream_t stream = at::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
    const int rc = grpg_visible_filter(P, 0, image_width, image_height, p_means, p_sc,
                                       scale_modifier, p_rot, p_cov, p_view, p_proj, tan_fovx,
                                       tan_fovy, prefiltered ? 1 : 0, radii.data_ptr<int>(),
                                       means2D.data_ptr<float>(), debug ? 1 : 0, (void*)stream);
    if (rc != GRPG_OK) raise_abi_error("grpg_visible_filter", rc);
  }
  return std::make_tuple(radii, means2D);
}

// ----------------------------------------------------------------------------------------------
// Fused scene-graph composition (additive; include/grpg_rasterizer.h: grpg_forward_composed).
// Per model one tensor each of the RAW parameters, in the reference's concatenation order
// (background first, then the visible actors); `poses` is a CPU float tensor [nseg, 8] =
// (rigid flag, obj_rot w x y z, obj_trans x y z), `idft` a CPU float tensor [nseg, GRPG_MAX_FOURIER].
// ----------------------------------------------------------------------------------------------
namespace {

struct SegmentPack {
  std::vector<grpg_model_segment> segs;
  std::vector<torch::Tensor> keep;
  int64_t P = 0;
  int M = 0;
};

SegmentPack pack_segments(const std::vector<torch::Tensor>& xyz, const std::vector<torch::Tensor>& scaling,
                          const std::vector<torch::Tensor>& rotation,
                          const std::vector<torch::Tensor>& opacity,
                          const std::vector<torch::Tensor>& features_dc,
                          const std::vector<torch::Tensor>& features_rest,
                          const std::vector<torch::Tensor>& flip,
                          const torch::Tensor& poses, const torch::Tensor& idft) {
  const size_t n = xyz.size();
  TORCH_CHECK(flip.size() == n, "one flip mask (or an empty tensor) per model");
  TORCH_CHECK(n > 0 && n <= GRPG_MAX_SEGMENTS, "need 1..", GRPG_MAX_SEGMENTS, " models");
  TORCH_CHECK(scaling.size() == n && rotation.size() == n && opacity.size() == n &&
                  features_dc.size() == n && features_rest.size() == n,
              "one tensor per model in every parameter list");
  TORCH_CHECK(!poses.is_cuda() && poses.scalar_type() == torch::kFloat32 && poses.dim() == 2 &&
                  poses.size(0) == (int64_t)n && poses.size(1) == 8,
              "poses must be a CPU float tensor [num_models, 8]");
  TORCH_CHECK(!idft.is_cuda() && idft.scalar_type() == torch::kFloat32 && idft.dim() == 2 &&
                  idft.size(0) == (int64_t)n && idft.size(1) == GRPG_MAX_FOURIER,
              "idft must be a CPU float tensor [num_models, ", GRPG_MAX_FOURIER, "]");
  const torch::Tensor pc = poses.contiguous(), ic = idft.contiguous();
  SegmentPack pk;
  pk.segs.resize(n);
  const torch::Tensor& like = xyz[0];
  require_device(like);
  for (size_t i = 0; i < n; i++) {
    grpg_model_segment& g = pk.segs[i];
    const int64_t cnt = xyz[i].size(0);
    TORCH_CHECK(cnt > 0, "model ", i, " is empty: leave it out of the lists");
    TORCH_CHECK(xyz[i].dim() == 2 && xyz[i].size(1) == 3 && scaling[i].numel() == cnt * 3 &&
                    rotation[i].numel() == cnt * 4 && opacity[i].numel() == cnt,
                "model ", i, ": xyz [N,3], scaling [N,3], rotation [N,4], opacity [N,1]");
    TORCH_CHECK(features_dc[i].dim() == 3 && features_dc[i].size(0) == cnt && features_dc[i].size(2) == 3,
                "model ", i, ": features_dc must be [N, fourier_dim, 3]");
    const int F = features_dc[i].size(1);
    const int Mi = 1 + (features_rest[i].numel() > 0 ? (int)features_rest[i].size(1) : 0);
    if (i == 0) pk.M = Mi;
    TORCH_CHECK(Mi == pk.M, "all models must carry the same number of SH coefficients");
    torch::Tensor k[6];
    g.xyz = fptr(xyz[i], like, "xyz", k[0]);
    g.scaling = fptr(scaling[i], like, "scaling", k[1]);
    g.rotation = fptr(rotation[i], like, "rotation", k[2]);
    g.opacity = fptr(opacity[i], like, "opacity", k[3]);
    g.features_dc = fptr(features_dc[i], like, "features_dc", k[4]);
    g.features_rest = fptr(features_rest[i], like, "features_rest", k[5]);
    for (auto& t : k) pk.keep.push_back(t);
    g.flip = nullptr;
    if (flip[i].numel() != 0) {   // training symmetry prior: bool / uint8 [N] on the device
      TORCH_CHECK(flip[i].is_cuda() && flip[i].device() == like.device() && flip[i].numel() == cnt &&
                      (flip[i].scalar_type() == torch::kBool || flip[i].scalar_type() == torch::kUInt8),
                  "model ", i, ": flip must be a bool / uint8 device tensor [N]");
      torch::Tensor fc = flip[i].contiguous();
      pk.keep.push_back(fc);
      g.flip = reinterpret_cast<const unsigned char*>(fc.data_ptr());
    }
    g.count = (int)cnt;
    g.fourier_dim = F;
    const float* pr = pc.data_ptr<float>() + 8 * i;
    g.rigid = pr[0] != 0.0f ? 1 : 0;
    for (int q = 0; q < 4; q++) g.obj_rot[q] = pr[1 + q];
    for (int q = 0; q < 3; q++) g.obj_trans[q] = pr[5 + q];
    for (int q = 0; q < GRPG_MAX_FOURIER; q++) g.idft[q] = ic.data_ptr<float>()[GRPG_MAX_FOURIER * i + q];
    pk.P += cnt;
  }
  return pk;
}

}  // namespace

std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor, torch::Tensor>
RasterizeGaussiansComposed(const torch::Tensor& background, const std::vector<torch::Tensor>& xyz,
                           const std::vector<torch::Tensor>& scaling,
                           const std::vector<torch::Tensor>& rotation,
                           const std::vector<torch::Tensor>& opacity,
                           const std::vector<torch::Tensor>& features_dc,
                           const std::vector<torch::Tensor>& features_rest,
                           const std::vector<torch::Tensor>& flip,
                           const torch::Tensor& poses, const torch::Tensor& idft,
                           const float scale_modifier, const torch::Tensor& viewmatrix,
                           const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                           const int image_height, const int image_width, const int degree,
                           const torch::Tensor& campos, const bool debug, const bool for_backward) {
  SegmentPack pk = pack_segments(xyz, scaling, rotation, opacity, features_dc, features_rest, flip, poses, idft);
  const torch::Tensor& like = xyz[0];
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(like.device());
  const int H = image_height, W = image_width;
  auto float_opts = like.options().dtype(torch::kFloat32);
  torch::Tensor out_color = torch::empty({GRPG_NUM_CHANNELS, H, W}, float_opts);
  torch::Tensor out_depth = torch::empty({1, H, W}, float_opts);
  torch::Tensor out_alpha = torch::empty({1, H, W}, float_opts);
  torch::Tensor radii = torch::empty({pk.P}, like.options().dtype(torch::kInt32));
  auto byte_opts = like.options().dtype(torch::kByte);
  torch::Tensor geomBuffer = torch::empty({0}, byte_opts);
  torch::Tensor binningBuffer = torch::empty({0}, byte_opts);
  torch::Tensor imgBuffer = torch::empty({0}, byte_opts);
  torch::Tensor k_bg, k_view, k_proj, k_cam;
  const float* p_bg = fptr(background, like, "background", k_bg);
  const float* p_view = fptr(viewmatrix, like, "viewmatrix", k_view);
  const float* p_proj = fptr(projmatrix, like, "projmatrix", k_proj);
  const float* p_cam = fptr(campos, like, "campos", k_cam);
  TORCH_CHECK(p_bg && p_view && p_proj && p_cam, "bg/viewmatrix/projmatrix/campos must be non-empty");
  hipStream_t stream = at::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  int rendered;
  {
    pybind11::gil_scoped_release nogil;
    rendered = grpg_forward_composed_flags(
        resize_blob, &geomBuffer, resize_blob, &binningBuffer, resize_blob, &imgBuffer, pk.segs.data(),
        (int)pk.segs.size(), degree, pk.M, p_bg, W, H, scale_modifier, p_view, p_proj, p_cam, tan_fovx,
        tan_fovy, out_color.data_ptr<float>(), out_depth.data_ptr<float>(), out_alpha.data_ptr<float>(),
        radii.data_ptr<int>(), debug ? 1 : 0, (void*)stream, for_backward ? 0u : GRPG_FORWARD_NO_BACKWARD);
  }
  if (rendered < 0) raise_abi_error("grpg_forward_composed", rendered);
  return std::make_tuple(rendered, out_color, out_depth, out_alpha, radii, geomBuffer, binningBuffer,
                         imgBuffer);
}

// Composition + layered frame (grpg_forward_composed_layers): the reference's whole evaluation render of a frame --
// StreetGaussianModel's getters and StreetGaussianRenderer.render_all -- from the models' raw parameters.
// object_model: uint8 [n] on the host, != 0 = the model belongs to the object layer (empty: actors = objects).
// returns (num_rendered, color, depth, alpha, radii, color_bg, alpha_bg, color_obj, alpha_obj)
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor, torch::Tensor>
RasterizeGaussiansComposedLayers(const torch::Tensor& background, const torch::Tensor& layer_background,
                                 const torch::Tensor& object_model, const std::vector<torch::Tensor>& xyz,
                                 const std::vector<torch::Tensor>& scaling,
                                 const std::vector<torch::Tensor>& rotation,
                                 const std::vector<torch::Tensor>& opacity,
                                 const std::vector<torch::Tensor>& features_dc,
                                 const std::vector<torch::Tensor>& features_rest,
                                 const std::vector<torch::Tensor>& flip, const torch::Tensor& poses,
                                 const torch::Tensor& idft, const float scale_modifier,
                                 const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix,
                                 const float tan_fovx, const float tan_fovy, const int image_height,
                                 const int image_width, const int degree, const torch::Tensor& campos,
                                 const bool debug) {
  SegmentPack pk = pack_segments(xyz, scaling, rotation, opacity, features_dc, features_rest, flip, poses, idft);
  const torch::Tensor& like = xyz[0];
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(like.device());
  const int H = image_height, W = image_width;
  const unsigned char* p_cls = nullptr;
  torch::Tensor k_cls;
  if (object_model.numel() != 0) {
    TORCH_CHECK(object_model.numel() == (int64_t)pk.segs.size() && !object_model.is_cuda() &&
                    (object_model.scalar_type() == torch::kUInt8 || object_model.scalar_type() == torch::kBool),
                "object_model must be a uint8 / bool HOST tensor with one element per model");
    k_cls = object_model.contiguous();
    p_cls = (const unsigned char*)k_cls.data_ptr();
  }
  auto fo = like.options().dtype(torch::kFloat32);
  torch::Tensor out_color = torch::empty({GRPG_NUM_CHANNELS, H, W}, fo), out_depth = torch::empty({1, H, W}, fo);
  torch::Tensor out_alpha = torch::empty({1, H, W}, fo);
  torch::Tensor color_bg = torch::empty({GRPG_NUM_CHANNELS, H, W}, fo), alpha_bg = torch::empty({1, H, W}, fo);
  torch::Tensor color_obj = torch::empty({GRPG_NUM_CHANNELS, H, W}, fo), alpha_obj = torch::empty({1, H, W}, fo);
  torch::Tensor radii = torch::empty({pk.P}, like.options().dtype(torch::kInt32));
  auto byte_opts = like.options().dtype(torch::kByte);
  torch::Tensor geomBuffer = torch::empty({0}, byte_opts), binningBuffer = torch::empty({0}, byte_opts);
  torch::Tensor imgBuffer = torch::empty({0}, byte_opts);
  torch::Tensor k_bg, k_lbg, k_view, k_proj, k_cam;
  const float* p_bg = fptr(background, like, "background", k_bg);
  const float* p_lbg = fptr(layer_background, like, "layer_background", k_lbg);
  const float* p_view = fptr(viewmatrix, like, "viewmatrix", k_view);
  const float* p_proj = fptr(projmatrix, like, "projmatrix", k_proj);
  const float* p_cam = fptr(campos, like, "campos", k_cam);
  TORCH_CHECK(p_bg && p_lbg && p_view && p_proj && p_cam, "bg/layer_bg/viewmatrix/projmatrix/campos must be non-empty");
  hipStream_t stream = at::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  int rendered;
  {
    pybind11::gil_scoped_release nogil;
    rendered = grpg_forward_composed_layers(
        resize_blob, &geomBuffer, resize_blob, &binningBuffer, resize_blob, &imgBuffer, pk.segs.data(),
        (int)pk.segs.size(), p_cls, degree, pk.M, p_bg, p_lbg, W, H, scale_modifier, p_view, p_proj, p_cam, tan_fovx,
        tan_fovy, out_color.data_ptr<float>(), out_depth.data_ptr<float>(), out_alpha.data_ptr<float>(),
        color_bg.data_ptr<float>(), alpha_bg.data_ptr<float>(), color_obj.data_ptr<float>(),
        alpha_obj.data_ptr<float>(), radii.data_ptr<int>(), debug ? 1 : 0, (void*)stream);
  }
  if (rendered < 0) raise_abi_error("grpg_forward_composed_layers", rendered);
  return std::make_tuple(rendered, out_color, out_depth, out_alpha, radii, color_bg, alpha_bg, color_obj, alpha_obj);
}

// The scene-graph frame as one call (grpg_forward_composed_frame): composition + op (+ layers) + sky + clamp + rgb8.
FrameResult RasterizeGaussiansComposedFrame(
    const torch::Tensor& background, const torch::Tensor& layer_background, const torch::Tensor& object_model,
    const bool layered, const std::vector<torch::Tensor>& xyz, const std::vector<torch::Tensor>& scaling,
    const std::vector<torch::Tensor>& rotation, const std::vector<torch::Tensor>& opacity,
    const std::vector<torch::Tensor>& features_dc, const std::vector<torch::Tensor>& features_rest,
    const std::vector<torch::Tensor>& flip, const torch::Tensor& poses, const torch::Tensor& idft,
    const float scale_modifier, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix,
    const float tan_fovx, const float tan_fovy, const int image_height, const int image_width, const int degree,
    const torch::Tensor& campos, const bool debug, const torch::Tensor& sky_cube, const torch::Tensor& ray_matrix,
    const float sky_fill, const bool clamp, const bool want_planes, const bool want_rgb8, const bool truncate,
    const c10::optional<torch::Tensor>& out_rgb8) {
  SegmentPack pk = pack_segments(xyz, scaling, rotation, opacity, features_dc, features_rest, flip, poses, idft);
  const torch::Tensor& like = xyz[0];
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(like.device());
  TORCH_CHECK(want_planes || want_rgb8, "forward_frame: ask for the float planes, the rgb8 frame, or both");
  const int H = image_height, W = image_width;
  const unsigned char* p_cls = nullptr;
  torch::Tensor k_cls;
  if (layered && object_model.numel() != 0) {
    TORCH_CHECK(object_model.numel() == (int64_t)pk.segs.size() && !object_model.is_cuda() &&
                    (object_model.scalar_type() == torch::kUInt8 || object_model.scalar_type() == torch::kBool),
                "object_model must be a uint8 / bool HOST tensor with one element per model");
    k_cls = object_model.contiguous();
    p_cls = (const unsigned char*)k_cls.data_ptr();
  }
  auto fo = like.options().dtype(torch::kFloat32);
  auto none = [&]() { return torch::empty({0}, fo); };
  torch::Tensor out_color = want_planes ? torch::empty({GRPG_NUM_CHANNELS, H, W}, fo) : none();
  torch::Tensor out_depth = want_planes ? torch::empty({1, H, W}, fo) : none();
  torch::Tensor out_alpha = want_planes ? torch::empty({1, H, W}, fo) : none();
  torch::Tensor color_bg = layered ? torch::empty({GRPG_NUM_CHANNELS, H, W}, fo) : none();
  torch::Tensor alpha_bg = layered ? torch::empty({1, H, W}, fo) : none();
  torch::Tensor color_obj = layered ? torch::empty({GRPG_NUM_CHANNELS, H, W}, fo) : none();
  torch::Tensor alpha_obj = layered ? torch::empty({1, H, W}, fo) : none();
  torch::Tensor radii = torch::empty({pk.P}, like.options().dtype(torch::kInt32));
  auto byte_opts = like.options().dtype(torch::kByte);
  torch::Tensor geomBuffer = torch::empty({0}, byte_opts), binningBuffer = torch::empty({0}, byte_opts);
  torch::Tensor imgBuffer = torch::empty({0}, byte_opts);
  torch::Tensor k_bg, k_lbg, k_view, k_proj, k_cam;
  const float* p_bg = fptr(background, like, "background", k_bg);
  const float* p_lbg = fptr(layer_background, like, "layer_background", k_lbg);
  const float* p_view = fptr(viewmatrix, like, "viewmatrix", k_view);
  const float* p_proj = fptr(projmatrix, like, "projmatrix", k_proj);
  const float* p_cam = fptr(campos, like, "campos", k_cam);
  TORCH_CHECK(p_bg && p_view && p_proj && p_cam, "bg/viewmatrix/projmatrix/campos must be non-empty");
  TORCH_CHECK(!layered || p_lbg, "layer_background must be non-empty in a layered frame");
  EpilogueArgs ea;
  make_epilogue(ea, like, sky_cube, ray_matrix, sky_fill, clamp, want_rgb8, truncate, out_rgb8, H, W);
  hipStream_t stream = at::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  int rendered;
  {
    pybind11::gil_scoped_release nogil;
    rendered = grpg_forward_composed_frame(
        resize_blob, &geomBuffer, resize_blob, &binningBuffer, resize_blob, &imgBuffer, pk.segs.data(),
        (int)pk.segs.size(), p_cls, degree, pk.M, p_bg, p_lbg, W, H, scale_modifier, p_view, p_proj, p_cam, tan_fovx,
        tan_fovy, want_planes ? out_color.data_ptr<float>() : nullptr,
        want_planes ? out_depth.data_ptr<float>() : nullptr, want_planes ? out_alpha.data_ptr<float>() : nullptr,
        layered ? color_bg.data_ptr<float>() : nullptr, layered ? alpha_bg.data_ptr<float>() : nullptr,
        layered ? color_obj.data_ptr<float>() : nullptr, layered ? alpha_obj.data_ptr<float>() : nullptr,
        radii.data_ptr<int>(), debug ? 1 : 0, (void*)stream, &ea.e);
  }
  if (rendered < 0) raise_abi_error("grpg_forward_composed_frame", rendered);
  return std::make_tuple(rendered, ea.rgb8, out_color, out_depth, out_alpha, radii, color_bg, alpha_bg, color_obj,
                         alpha_obj);
}

// Training backward of the fused composition (grpg_backward_composed): gradients with respect to
// every model's RAW parameter tensors, means2D [P,3] (densification statistic) and the poses [n,8].
std::tuple<std::vector<torch::Tensor>, std::vector<torch::Tensor>, std::vector<torch::Tensor>,
           std::vector<torch::Tensor>, std::vector<torch::Tensor>, std::vector<torch::Tensor>,
           torch::Tensor, torch::Tensor>
RasterizeGaussiansComposedBackward(
    const torch::Tensor& background, const std::vector<torch::Tensor>& xyz,
    const std::vector<torch::Tensor>& scaling, const std::vector<torch::Tensor>& rotation,
    const std::vector<torch::Tensor>& opacity, const std::vector<torch::Tensor>& features_dc,
    const std::vector<torch::Tensor>& features_rest, const std::vector<torch::Tensor>& flip,
    const torch::Tensor& poses, const torch::Tensor& idft,
    const float scale_modifier, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix,
    const float tan_fovx, const float tan_fovy, const int degree, const torch::Tensor& campos,
    const torch::Tensor& radii, const torch::Tensor& alphas, const torch::Tensor& geomBuffer, const int R,
    const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer,
    const torch::Tensor& dL_dout_color, const torch::Tensor& dL_dout_depth,
    const torch::Tensor& dL_dout_alpha, const bool debug) {
  SegmentPack pk = pack_segments(xyz, scaling, rotation, opacity, features_dc, features_rest, flip, poses, idft);
  const torch::Tensor& like = xyz[0];
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(like.device());
  const int H = dL_dout_color.size(1), W = dL_dout_color.size(2);
  const size_t n = xyz.size();
  auto o = like.options().dtype(torch::kFloat32);
  std::vector<torch::Tensor> g_xyz(n), g_scaling(n), g_rotation(n), g_opacity(n), g_fdc(n), g_frest(n);
  std::vector<grpg_model_segment_grad> gs(n);
  for (size_t i = 0; i < n; i++) {   // every element is written by the kernel: no zero-fill
    g_xyz[i] = torch::empty(xyz[i].sizes(), o);
    g_scaling[i] = torch::empty(scaling[i].sizes(), o);
    g_rotation[i] = torch::empty(rotation[i].sizes(), o);
    g_opacity[i] = torch::empty(opacity[i].sizes(), o);
    g_fdc[i] = torch::empty(features_dc[i].sizes(), o);
    g_frest[i] = torch::empty(features_rest[i].sizes(), o);
    gs[i] = grpg_model_segment_grad{g_xyz[i].data_ptr<float>(), g_scaling[i].data_ptr<float>(),
                                    g_rotation[i].data_ptr<float>(), g_opacity[i].data_ptr<float>(),
                                    g_fdc[i].data_ptr<float>(),
                                    g_frest[i].numel() ? g_frest[i].data_ptr<float>() : nullptr};
  }
  torch::Tensor dL_dmeans2D = torch::empty({pk.P, 3}, o);
  torch::Tensor dL_dposes = torch::empty({(int64_t)n, 8}, o);
  torch::Tensor k[7];
  const float* p_bg = fptr(background, like, "background", k[0]);
  const float* p_view = fptr(viewmatrix, like, "viewmatrix", k[1]);
  const float* p_proj = fptr(projmatrix, like, "projmatrix", k[2]);
  const float* p_cam = fptr(campos, like, "campos", k[3]);
  const float* g_col = fptr(dL_dout_color, like, "dL_dout_color", k[4]);
  const float* g_dep = fptr(dL_dout_depth, like, "dL_dout_depth", k[5]);
  const float* g_alp = fptr(dL_dout_alpha, like, "dL_dout_alpha", k[6]);
  torch::Tensor k_alpha;
  const float* p_alpha = fptr(alphas, like, "alphas", k_alpha);
  TORCH_CHECK(radii.scalar_type() == torch::kInt32 && radii.device() == like.device(),
              "radii must be int32 on the device");
  torch::Tensor radii_c = radii.contiguous();
  torch::Tensor gb = geomBuffer.contiguous(), bb = binningBuffer.contiguous(), ib = imageBuffer.contiguous();
  hipStream_t stream = at::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  const int rc = grpg_backward_composed(
      pk.segs.data(), gs.data(), (int)n, degree, pk.M, R, p_bg, W, H, scale_modifier, p_view, p_proj, p_cam,
      tan_fovx, tan_fovy, radii_c.data_ptr<int>(), p_alpha, reinterpret_cast<char*>(gb.data_ptr()),
      reinterpret_cast<char*>(bb.data_ptr()), reinterpret_cast<char*>(ib.data_ptr()), g_col, g_dep, g_alp,
      dL_dmeans2D.data_ptr<float>(), dL_dposes.data_ptr<float>(), debug ? 1 : 0, (void*)stream);
  if (rc != GRPG_OK) raise_abi_error("grpg_backward_composed", rc);
  return std::make_tuple(g_xyz, g_scaling, g_rotation, g_opacity, g_fdc, g_frest, dL_dmeans2D, dL_dposes);
}

// (means3D [P,3], scales [P,3], rotations [P,4], opacity [P,1], shs [P,M,3]): what the reference's
// get_xyz / get_scaling / get_rotation / get_opacity / get_features return for the same models
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
Compose(const std::vector<torch::Tensor>& xyz, const std::vector<torch::Tensor>& scaling,
        const std::vector<torch::Tensor>& rotation, const std::vector<torch::Tensor>& opacity,
        const std::vector<torch::Tensor>& features_dc, const std::vector<torch::Tensor>& features_rest,
        const std::vector<torch::Tensor>& flip, const torch::Tensor& poses, const torch::Tensor& idft) {
  SegmentPack pk = pack_segments(xyz, scaling, rotation, opacity, features_dc, features_rest, flip, poses, idft);
  const torch::Tensor& like = xyz[0];
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(like.device());
  auto o = like.options().dtype(torch::kFloat32);
  torch::Tensor means = torch::empty({pk.P, 3}, o), scales = torch::empty({pk.P, 3}, o),
                rots = torch::empty({pk.P, 4}, o), opac = torch::empty({pk.P, 1}, o),
                shs = torch::empty({pk.P, pk.M, 3}, o);
  hipStream_t stream = at::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  const int rc = grpg_compose(pk.segs.data(), (int)pk.segs.size(), pk.M, means.data_ptr<float>(),
                              scales.data_ptr<float>(), rots.data_ptr<float>(), opac.data_ptr<float>(),
                              shs.data_ptr<float>(), (void*)stream);
  if (rc != GRPG_OK) raise_abi_error("grpg_compose", rc);
  return std::make_tuple(means, scales, rots, opac, shs);
}

// ----------------------------------------------------------------------------------------------
// Sky cube map (additive; include/grpg_rasterizer.h: grpg_sky_composite / grpg_sky_backward).
// ray_matrix: CPU float tensor [3,3] = R^T K^-1.
// ----------------------------------------------------------------------------------------------
// ray_matrix [3,3] float32 = R^T K^-1, on the CPU or on the cube map's device (the latter costs no
// host read of the camera); mask (bool / uint8 [H,W]) and jitter (float32 [2,H,W]) are the train-mode
// extras of grpg_sky_composite_ex.
static const unsigned char* sky_mask_ptr(const c10::optional<torch::Tensor>& mask_opt, torch::Tensor& hold,
                                         const int height, const int width) {
  if (!mask_opt.has_value() || !mask_opt->defined()) return nullptr;
  TORCH_CHECK(mask_opt->is_cuda() && (mask_opt->scalar_type() == torch::kBool ||
                                      mask_opt->scalar_type() == torch::kUInt8) &&
                  mask_opt->numel() == (int64_t)height * width,
              "sky mask must be a bool / uint8 device tensor [H,W]");
  hold = mask_opt->contiguous();
  return (const unsigned char*)hold.data_ptr();
}
static const float* sky_jitter_ptr(const c10::optional<torch::Tensor>& jit_opt, torch::Tensor& hold,
                                   const int height, const int width) {
  if (!jit_opt.has_value() || !jit_opt->defined()) return nullptr;
  TORCH_CHECK(jit_opt->is_cuda() && jit_opt->scalar_type() == torch::kFloat32 &&
                  jit_opt->numel() == 2 * (int64_t)height * width,
              "sky jitter must be a float32 device tensor [2,H,W]");
  hold = jit_opt->contiguous();
  return hold.data_ptr<float>();
}
static void check_ray_matrix(const torch::Tensor& ray_matrix, const torch::Tensor& cube) {
  TORCH_CHECK(ray_matrix.scalar_type() == torch::kFloat32 && ray_matrix.numel() == 9 &&
                  (!ray_matrix.is_cuda() || ray_matrix.device() == cube.device()),
              "ray_matrix must be a float32 tensor [3,3] on the CPU or on the cube map's device");
}

std::tuple<torch::Tensor, torch::Tensor>
SkyComposite(const torch::Tensor& cube, const torch::Tensor& ray_matrix, const float fill,
             const bool clamp_out, const c10::optional<torch::Tensor>& rgb_opt,
             const c10::optional<torch::Tensor>& acc_opt, const int height, const int width,
             const bool want_sky, const c10::optional<torch::Tensor>& mask_opt,
             const c10::optional<torch::Tensor>& jitter_opt) {
  TORCH_CHECK(cube.is_cuda() && cube.scalar_type() == torch::kFloat32 && cube.dim() == 4 &&
                  cube.size(0) == 6 && cube.size(1) == cube.size(2) && cube.size(3) == 3,
              "sky cube map must be a float32 device tensor [6,res,res,3]");
  check_ray_matrix(ray_matrix, cube);
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(cube.device());
  torch::Tensor cu = cube.contiguous(), rm = ray_matrix.contiguous(), rgb, acc, out, sky, mask, jit;
  const float* p_rgb = nullptr;
  const float* p_acc = nullptr;
  if (rgb_opt.has_value() && rgb_opt->defined()) {
    TORCH_CHECK(rgb_opt->is_cuda() && rgb_opt->scalar_type() == torch::kFloat32 && rgb_opt->dim() == 3 &&
                    rgb_opt->size(0) == 3 && rgb_opt->size(1) == height && rgb_opt->size(2) == width,
                "rgb must be a float32 device tensor [3,H,W]");
    rgb = rgb_opt->contiguous();
    p_rgb = rgb.data_ptr<float>();
    out = torch::empty_like(rgb);
  }
  if (acc_opt.has_value() && acc_opt->defined()) {
    TORCH_CHECK(acc_opt->is_cuda() && acc_opt->scalar_type() == torch::kFloat32 &&
                    acc_opt->numel() == (int64_t)height * width, "acc must be a float32 device tensor [1,H,W]");
    acc = acc_opt->contiguous();
    p_acc = acc.data_ptr<float>();
  }
  const unsigned char* p_mask = sky_mask_ptr(mask_opt, mask, height, width);
  const float* p_jit = sky_jitter_ptr(jitter_opt, jit, height, width);
  if (want_sky || !p_rgb) sky = torch::empty({3, height, width}, cu.options());
  hipStream_t stream = at::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  const int rc = grpg_sky_composite_ex(cu.data_ptr<float>(), (int)cu.size(1), rm.data_ptr<float>(),
                                       rm.is_cuda() ? 1 : 0, fill, clamp_out ? 1 : 0, width, height, p_rgb,
                                       p_acc, p_mask, p_jit, p_rgb ? out.data_ptr<float>() : nullptr,
                                       sky.defined() ? sky.data_ptr<float>() : nullptr, (void*)stream);
  if (rc != GRPG_OK) raise_abi_error("grpg_sky_composite", rc);
  return std::make_tuple(out, sky);
}

std::tuple<torch::Tensor, torch::Tensor>
SkyBackward(const torch::Tensor& cube, const torch::Tensor& ray_matrix, const float fill,
            const c10::optional<torch::Tensor>& acc_opt, const torch::Tensor& grad_rgb,
            const c10::optional<torch::Tensor>& mask_opt, const c10::optional<torch::Tensor>& jitter_opt) {
  TORCH_CHECK(cube.is_cuda() && cube.scalar_type() == torch::kFloat32 && grad_rgb.is_cuda() &&
                  grad_rgb.scalar_type() == torch::kFloat32 && grad_rgb.dim() == 3 && grad_rgb.size(0) == 3,
              "cube and grad_rgb [3,H,W] must be float32 device tensors");
  check_ray_matrix(ray_matrix, cube);
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(cube.device());
  const int H = grad_rgb.size(1), W = grad_rgb.size(2);
  torch::Tensor cu = cube.contiguous(), rm = ray_matrix.contiguous(), g = grad_rgb.contiguous(), acc, mask, jit;
  const float* p_acc = nullptr;
  if (acc_opt.has_value() && acc_opt->defined()) {
    TORCH_CHECK(acc_opt->is_cuda() && acc_opt->scalar_type() == torch::kFloat32 &&
                    acc_opt->numel() == (int64_t)H * W, "acc must be a float32 device tensor [1,H,W]");
    acc = acc_opt->contiguous();
    p_acc = acc.data_ptr<float>();
  }
  const unsigned char* p_mask = sky_mask_ptr(mask_opt, mask, H, W);
  const float* p_jit = sky_jitter_ptr(jitter_opt, jit, H, W);
  torch::Tensor grad_cube = torch::zeros_like(cu);
  torch::Tensor grad_acc = torch::empty({1, H, W}, cu.options());
  hipStream_t stream = at::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  const int rc = grpg_sky_backward_ex(cu.data_ptr<float>(), (int)cu.size(1), rm.data_ptr<float>(),
                                      rm.is_cuda() ? 1 : 0, fill, W, H, p_acc, p_mask, p_jit,
                                      g.data_ptr<float>(), grad_cube.data_ptr<float>(),
                                      grad_acc.data_ptr<float>(), (void*)stream);
  if (rc != GRPG_OK) raise_abi_error("grpg_sky_backward", rc);
  return std::make_tuple(grad_cube, grad_acc);
}

// Parity/debug accessor (no reference counterpart): decode the private blobs of a
// forward call into the reference's intermediate arrays.
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
DebugExport(const torch::Tensor& geomBuffer, const torch::Tensor& binningBuffer,
            const torch::Tensor& imgBuffer, const int P, const int R, const int image_height,
            const int image_width) {
  TORCH_CHECK(geomBuffer.is_cuda(), "buffers must be device tensors");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(geomBuffer.device());
  const int gx = (image_width + GRPG_TILE_X - 1) / GRPG_TILE_X;
  const int gy = (image_height + GRPG_TILE_Y - 1) / GRPG_TILE_Y;
  auto o = geomBuffer.options();
  torch::Tensor keys = torch::zeros({R}, o.dtype(torch::kInt64));
  torch::Tensor plist = torch::zeros({R}, o.dtype(torch::kInt32));
  torch::Tensor ranges = torch::zeros({gx * gy, 2}, o.dtype(torch::kInt32));
  torch::Tensor ncontrib = torch::zeros({image_height, image_width}, o.dtype(torch::kInt32));
  torch::Tensor means2D = torch::zeros({P, 2}, o.dtype(torch::kFloat32));
  torch::Tensor depths = torch::zeros({P}, o.dtype(torch::kFloat32));
  torch::Tensor conic = torch::zeros({P, 4}, o.dtype(torch::kFloat32));
  torch::Tensor rgb = torch::zeros({P, 3}, o.dtype(torch::kFloat32));
  torch::Tensor tiles = torch::zeros({P}, o.dtype(torch::kInt32));
  hipStream_t stream = at::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  const int rc = grpg_debug_export(
      P, R, image_width, image_height, reinterpret_cast<const char*>(geomBuffer.data_ptr()),
      reinterpret_cast<const char*>(binningBuffer.data_ptr()),
      reinterpret_cast<const char*>(imgBuffer.data_ptr()),
      reinterpret_cast<uint64_t*>(keys.data_ptr<int64_t>()),
      reinterpret_cast<uint32_t*>(plist.data_ptr<int>()),
      reinterpret_cast<uint32_t*>(ranges.data_ptr<int>()),
      reinterpret_cast<uint32_t*>(ncontrib.data_ptr<int>()), means2D.data_ptr<float>(),
      depths.data_ptr<float>(), conic.data_ptr<float>(), rgb.data_ptr<float>(),
      reinterpret_cast<uint32_t*>(tiles.data_ptr<int>()), (void*)stream);
  if (rc != GRPG_OK) raise_abi_error("grpg_debug_export", rc);
  return std::make_tuple(keys, plist, ranges, ncontrib, means2D, depths, conic, rgb, tiles);
}

// out: optional preallocated uint8 tensor of the same shape (e.g. a slot of the gather buffer), so
// that the packed frame is written in place instead of being copied there afterwards
torch::Tensor PackU8(const torch::Tensor& color, const c10::optional<torch::Tensor>& out_opt) {
  TORCH_CHECK(color.is_cuda() && color.scalar_type() == torch::kFloat32, "pack_u8: float32 device tensor");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(color.device());
  torch::Tensor src = color.contiguous();
  torch::Tensor out;
  if (out_opt.has_value() && out_opt->defined()) {
    out = *out_opt;
    TORCH_CHECK(out.is_cuda() && out.scalar_type() == torch::kUInt8 && out.is_contiguous() &&
                    out.numel() == src.numel(), "pack_u8: out must be a contiguous uint8 device tensor of the same size");
  } else {
    out = torch::empty(src.sizes(), src.options().dtype(torch::kUInt8));
  }
  hipStream_t stream = at::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  const int rc = grpg_pack_rgb_u8(src.data_ptr<float>(), out.data_ptr<uint8_t>(), (size_t)src.numel(),
                                  (void*)stream);
  if (rc != GRPG_OK) raise_abi_error("grpg_pack_rgb_u8", rc);
  return out;
}

// float [3,H,W] -> uint8 [H,W,3] (rgb8); truncate=true reproduces the simulator's astype(uint8)
torch::Tensor PackHWC(const torch::Tensor& color, const bool truncate) {
  TORCH_CHECK(color.is_cuda() && color.scalar_type() == torch::kFloat32 && color.dim() == 3 &&
                  color.size(0) == 3, "pack_hwc: float32 device tensor [3,H,W]");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(color.device());
  torch::Tensor src = color.contiguous();
  const int H = src.size(1), W = src.size(2);
  torch::Tensor out = torch::empty({H, W, 3}, src.options().dtype(torch::kUInt8));
  hipStream_t stream = at::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  const int rc = grpg_pack_rgb_u8_hwc(src.data_ptr<float>(), out.data_ptr<uint8_t>(), H, W,
                                      truncate ? 1 : 0, (void*)stream);
  if (rc != GRPG_OK) raise_abi_error("grpg_pack_rgb_u8_hwc", rc);
  return out;
}

std::tuple<std::vector<float>, int> StageTiming() {
  std::vector<float> ms(GRPG_NUM_STAGES, 0.f);
  int calls = 0;
  const int rc = grpg_get_stage_timing(ms.data(), &calls);
  if (rc != GRPG_OK) raise_abi_error("grpg_get_stage_timing", rc);
  return std::make_tuple(ms, calls);
}

std::tuple<float, float, int> BackwardTiming() {
  float blend = 0.f, pre = 0.f;
  int calls = 0;
  const int rc = grpg_get_backward_timing(&blend, &pre, &calls);
  if (rc != GRPG_OK) raise_abi_error("grpg_get_backward_timing", rc);
  return std::make_tuple(blend, pre, calls);
}

// simple_knn._C.distCUDA2 (submodules/simple-knn/spatial.cu:14-25): [P,3] points -> [P] mean
// squared distance to the 3 nearest other points.
torch::Tensor distCUDA2(const torch::Tensor& points) {
  require_device(points);
  TORCH_CHECK(points.dim() == 2 && points.size(1) == 3, "points must be [P,3]");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(points.device());
  const int P = points.size(0);
  auto float_opts = points.options().dtype(torch::kFloat32);
  torch::Tensor pts = points.to(torch::kFloat32).contiguous();
  torch::Tensor means = torch::zeros({P}, float_opts);   // the reference returns torch::full(0)
  torch::Tensor workspace = torch::empty({0}, points.options().dtype(torch::kByte));
  hipStream_t stream = at::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  const int rc = grpg_knn_mean_dist2(P, pts.data_ptr<float>(), means.data_ptr<float>(), resize_blob,
                                     &workspace, (void*)stream);
  if (rc != GRPG_OK) raise_abi_error("grpg_knn_mean_dist2", rc);
  return means;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("rasterize_gaussians", &RasterizeGaussians);
  m.def("rasterize_gaussians_eval", &RasterizeGaussiansEval);
  m.def("rasterize_gaussians_eval_deferred", &RasterizeGaussiansEvalDeferred);
  m.def("rasterize_gaussians_layers", &RasterizeGaussiansLayers);
  m.def("frame_status", &FrameStatus, pybind11::arg("ticket"), pybind11::arg("wait") = true);
  m.def("distCUDA2", &distCUDA2);
  m.def("rasterize_gaussians_backward", &RasterizeGaussiansBackward);
  m.def("rasterize_gaussians_backward_lean", &RasterizeGaussiansBackwardLean);
  m.def("mark_visible", &markVisible);
  m.def("rasterize_gaussians_filter", &RasterizeGaussiansFilter);
  // additions (not in the reference module)
  m.def("debug_export", &DebugExport);
  m.def("rasterize_gaussians_composed", &RasterizeGaussiansComposed, pybind11::arg("bg"), pybind11::arg("xyz"),
        pybind11::arg("scaling"), pybind11::arg("rotation"), pybind11::arg("opacity"),
        pybind11::arg("features_dc"), pybind11::arg("features_rest"), pybind11::arg("flip"), pybind11::arg("poses"),
        pybind11::arg("idft"), pybind11::arg("scale_modifier"), pybind11::arg("viewmatrix"),
        pybind11::arg("projmatrix"), pybind11::arg("tan_fovx"), pybind11::arg("tan_fovy"),
        pybind11::arg("image_height"), pybind11::arg("image_width"), pybind11::arg("degree"),
        pybind11::arg("campos"), pybind11::arg("debug"), pybind11::arg("for_backward") = false);
  m.def("rasterize_gaussians_composed_backward", &RasterizeGaussiansComposedBackward);
  m.def("rasterize_gaussians_composed_layers", &RasterizeGaussiansComposedLayers);
  m.def("rasterize_gaussians_frame", &RasterizeGaussiansFrame);
  m.def("rasterize_gaussians_composed_frame", &RasterizeGaussiansComposedFrame);
  m.def("compose", &Compose);
  m.def("sky_composite", &SkyComposite, pybind11::arg("cube"), pybind11::arg("ray_matrix"),
        pybind11::arg("fill"), pybind11::arg("clamp_out"), pybind11::arg("rgb"), pybind11::arg("acc"),
        pybind11::arg("height"), pybind11::arg("width"), pybind11::arg("want_sky"),
        pybind11::arg("mask") = pybind11::none(), pybind11::arg("jitter") = pybind11::none());
  m.def("sky_backward", &SkyBackward, pybind11::arg("cube"), pybind11::arg("ray_matrix"),
        pybind11::arg("fill"), pybind11::arg("acc"), pybind11::arg("grad_rgb"),
        pybind11::arg("mask") = pybind11::none(), pybind11::arg("jitter") = pybind11::none());
  m.def("pack_u8", &PackU8, pybind11::arg("color"), pybind11::arg("out") = pybind11::none());
  m.def("pack_hwc", &PackHWC);
  m.def("get_backward_timing", &BackwardTiming);   // (blend ms sum, preprocess ms sum, calls)
  m.def("set_stage_timing", [](int mode) { grpg_set_stage_timing(mode); });   // 0 off, 1 all, 2 render only
  m.def("stage_timing", &StageTiming);
  m.def("abi_version", []() { return grpg_abi_version(); });
  // 0 = speculative binning capacity + deferred num_rendered wait (default), 1 = exact (reference-like)
  m.def("set_binning_mode", [](int mode) {
    if (grpg_set_binning_mode(mode) != GRPG_OK) raise_abi_error("grpg_set_binning_mode", -1);
  });
  m.def("reset_capacity_hints", []() { grpg_reset_capacity_hints(); });
  // (P, width, height, instances, coarse pairs): exact capacities for the next forward of that shape
  m.def("set_capacity_hint", [](int P, int W, int H, unsigned R, unsigned Rc) {
    const int rc = grpg_set_capacity_hint(P, W, H, R, Rc);
    if (rc != GRPG_OK) raise_abi_error("grpg_set_capacity_hint", rc);
  });
  m.def("get_binning_algorithm", []() { return grpg_get_binning_algorithm(); });
  m.def("set_binning_algorithm", [](int alg) {
    if (grpg_set_binning_algorithm(alg) != GRPG_OK) raise_abi_error("grpg_set_binning_algorithm", -1);
  });
}
