// olsr_torch.cpp — the reference's `_C` extension surface as a compiled torch extension over the C-ABI.
//
// Binds the same five functions as DGR/ext.cpp:15-21 (DGR = submodules/diff-gaussian-rasterization of the
// reference) with the positional signatures and return tuples of DGR/rasterize_points.h:17-152, followed by the
// three knobs the reference fixes at compile time (tile, bwd_mode, binning — CR/config.h:15-18).  Nothing is
// computed here: tensors are checked, made contiguous, allocated (torch owns device memory and the current HIP
// stream) and their pointers handed to olsr_forward / olsr_backward / olsr_mark_visible of include/olsr.h, which
// libolsr.so implements.  Host-only C++: built with g++ by online_lang_splatting_amd/build.py.
#include <torch/extension.h>

#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>

#include <algorithm>
#include <atomic>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/olsr.h"

namespace {

using torch::Tensor;

void check(int rc) {
  if (rc != OLSR_OK) throw std::runtime_error("olsr error " + std::to_string(rc) + ": " + olsr_last_error());
}

// contiguous fp32 tensor, or an undefined one for an absent input (0 elements -> nullptr)
Tensor prep(const Tensor &t, const char *what) {
  if (!t.defined() || t.numel() == 0) return Tensor();
  TORCH_CHECK(t.is_cuda(), what, " must live on the GPU (got a CPU tensor): this rasterizer has no CPU path");
  return t.scalar_type() == torch::kFloat32 ? t.contiguous() : t.contiguous().to(torch::kFloat32);
}

const float *fp(const Tensor &t) { return t.defined() && t.numel() > 0 ? t.data_ptr<float>() : nullptr; }
float *fpw(Tensor &t) { return t.defined() && t.numel() > 0 ? t.data_ptr<float>() : nullptr; }

// resizeFunctional, DGR/rasterize_points.cu:27-33
void *resize(void *user, size_t nbytes) {
  Tensor *t = static_cast<Tensor *>(user);
  t->resize_({static_cast<int64_t>(nbytes)});
  return t->data_ptr();
}

struct Scene {
  olsr_scene s;
  std::vector<Tensor> keep;  // the contiguous copies the pointers in `s` refer to
};

void fill_scene(Scene &sc, int F, const Tensor &bg, const Tensor &means3D, const Tensor &colors, const Tensor &language,
                const Tensor &opacity, const Tensor &scales, const Tensor &rotations, float scale_modifier,
                const Tensor &cov3D_precomp, const Tensor &viewmatrix, const Tensor &projmatrix,
                const Tensor &projmatrix_raw, float tan_fovx, float tan_fovy, int H, int W, const Tensor &sh, int degree,
                const Tensor &campos, bool prefiltered, bool debug, int tile, int bwd_mode, int binning, int flags) {
  const Tensor *in[13] = {&bg,        &means3D,       &sh,         &colors,     &language,       &opacity, &scales,
                          &rotations, &cov3D_precomp, &viewmatrix, &projmatrix, &projmatrix_raw, &campos};
  static const char *names[13] = {"bg",        "means3D",       "sh",         "colors_precomp", "language_precomp",
                                  "opacities", "scales",        "rotations",  "cov3D_precomp",  "viewmatrix",
                                  "projmatrix", "projmatrix_raw", "campos"};
  sc.keep.resize(13);
  for (int i = 0; i < 13; ++i) sc.keep[i] = prep(*in[i], names[i]);
  olsr_scene &s = sc.s;
  s = olsr_scene{};
  s.P = static_cast<int32_t>(means3D.size(0));
  s.D = degree;
  s.M = sc.keep[2].defined() ? static_cast<int32_t>(sc.keep[2].size(1)) : 0;
  s.F = F;
  s.width = W;
  s.height = H;
  s.tile = tile;
  s.prefiltered = prefiltered;
  s.debug = debug;
  s.bwd_mode = bwd_mode;
  s.tan_fovx = tan_fovx;
  s.tan_fovy = tan_fovy;
  s.scale_modifier = scale_modifier;
  s.binning = binning;
  s.flags = flags;
  s.background = fp(sc.keep[0]);
  s.means3D = fp(sc.keep[1]);
  s.shs = fp(sc.keep[2]);
  s.colors_precomp = fp(sc.keep[3]);
  s.language_precomp = fp(sc.keep[4]);
  s.opacities = fp(sc.keep[5]);
  s.scales = fp(sc.keep[6]);
  s.rotations = fp(sc.keep[7]);
  s.cov3D_precomp = fp(sc.keep[8]);
  s.viewmatrix = fp(sc.keep[9]);
  s.projmatrix = fp(sc.keep[10]);
  s.projmatrix_raw = fp(sc.keep[11]);
  s.cam_pos = fp(sc.keep[12]);
}

void *stream_of(const Tensor &t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

// RasterizeGaussiansCUDA / RasterizeLanguageGaussiansCUDA, DGR/rasterize_points.cu:35-123,125-241 (F == 0: the
// RGB-only rasterizer; the language tensor is then ignored and the returned language image is empty)
std::tuple<int, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> forward(
    int F, const Tensor &bg, const Tensor &means3D, const Tensor &colors, const Tensor &language, const Tensor &opacity,
    const Tensor &scales, const Tensor &rotations, float scale_modifier, const Tensor &cov3D_precomp,
    const Tensor &viewmatrix, const Tensor &projmatrix, const Tensor &projmatrix_raw, float tan_fovx, float tan_fovy,
    int image_height, int image_width, const Tensor &sh, int degree, const Tensor &campos, bool prefiltered, bool debug,
    int tile, int bwd_mode, int binning, int flags) {
  TORCH_CHECK(means3D.dim() == 2 && means3D.size(1) == 3,
              "means3D must have dimensions (num_points, 3)");  // DGR/rasterize_points.cu:159-161
  TORCH_CHECK(means3D.is_cuda(), "means3D must live on the GPU: this rasterizer has no CPU path");
  const c10::DeviceGuard guard(means3D.device());
  const int64_t P = means3D.size(0), H = image_height, W = image_width;
  Scene sc;
  fill_scene(sc, F, bg, means3D, colors, language, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
             projmatrix, projmatrix_raw, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered,
             debug, tile, bwd_mode, binning, flags);
  auto f32 = means3D.options().dtype(torch::kFloat32);
  auto i32 = means3D.options().dtype(torch::kInt32);
  auto u8 = means3D.options().dtype(torch::kUInt8);
  // every output is fully written by the library (no torch::full zero-fill, unlike DGR/rasterize_points.cu:170-175)
  Tensor out_color = torch::empty({3, H, W}, f32), out_lang = torch::empty({F, H, W}, f32);
  Tensor out_depth = torch::empty({1, H, W}, f32), out_opacity = torch::empty({1, H, W}, f32);
  Tensor radii = torch::empty({P}, i32), n_touched = torch::empty({P}, i32);
  Tensor geom = torch::empty({0}, u8), binb = torch::empty({0}, u8), img = torch::empty({0}, u8);
  int32_t R = 0;
  // The GIL is released for the call: the host waits inside until the GPU has reached this frame's instance count (behind
  // whatever the stream still holds), and other Python threads should run meanwhile.  The allocation callbacks resize C++
  // tensors that no Python object refers to yet (plain ATen calls: no interpreter state involved).
  int rc;
  {
    pybind11::gil_scoped_release nogil;
    rc = olsr_forward(&sc.s, resize, &geom, resize, &binb, resize, &img, out_color.data_ptr<float>(), fpw(out_lang),
                      out_depth.data_ptr<float>(), out_opacity.data_ptr<float>(), P ? radii.data_ptr<int32_t>() : nullptr,
                      P ? n_touched.data_ptr<int32_t>() : nullptr, &R, stream_of(means3D));
  }
  check(rc);
  return std::make_tuple(static_cast<int>(R), out_color, out_lang, radii, geom, binb, img, out_depth, out_opacity,
                         n_touched);
}

// gradient rows per instance of the frame whose count this process read last ([1]: packed survivor waves), and how often a
// guessed scratch size had to be followed by an exact second backward (tests / diagnostics)
std::atomic<float> g_rows_per_instance[2] = {{0.f}, {0.f}};
std::atomic<int> g_rows_redone{0};

// RasterizeGaussiansBackwardCUDA / RasterizeLanguageGaussiansBackwardCUDA, DGR/rasterize_points.cu:243-331,333-455.
// Returns {dL_dmeans2D, dL_dcolors, dL_dlanguage, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales,
// dL_drotations, dL_dtau, dL_dtau_sum, dL_dconic, dL_ddepths}; the last two (the reference's internal buffers,
// DGR/rasterize_points.cu:390-391) are undefined unless want_internal.
std::vector<Tensor> backward(int F, const Tensor &bg, const Tensor &means3D, const Tensor &radii, const Tensor &colors,
                             const Tensor &language, const Tensor &scales, const Tensor &rotations, float scale_modifier,
                             const Tensor &cov3D_precomp, const Tensor &viewmatrix, const Tensor &projmatrix,
                             const Tensor &projmatrix_raw, float tan_fovx, float tan_fovy, const Tensor &dL_dout_color,
                             const Tensor &dL_dout_language, const Tensor &dL_dout_depth, const Tensor &sh, int degree,
                             const Tensor &campos, const Tensor &geomBuffer, int R, const Tensor &binningBuffer,
                             const Tensor &imageBuffer, bool debug, bool want_internal, int tile, int bwd_mode,
                             int binning, int rows_token) {
  TORCH_CHECK(means3D.is_cuda(), "means3D must live on the GPU: this rasterizer has no CPU path");
  TORCH_CHECK(dL_dout_color.dim() == 3, "dL_dout_color must be [3, H, W]");
  const c10::DeviceGuard guard(means3D.device());
  const int64_t P = means3D.size(0);
  const int H = static_cast<int>(dL_dout_color.size(1)), W = static_cast<int>(dL_dout_color.size(2));
  Scene sc;
  fill_scene(sc, F, bg, means3D, colors, language, Tensor(), scales, rotations, scale_modifier, cov3D_precomp,
             viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy, H, W, sh, degree, campos, false, debug, tile,
             bwd_mode, binning, 0);
  const int64_t M = sc.s.M;
  auto f32 = means3D.options().dtype(torch::kFloat32);
  // written exactly once per row by the library: no torch::zeros (DGR/rasterize_points.cu:386-398)
  std::vector<Tensor> g(13);
  g[0] = torch::empty({P, 3}, f32);      // dL_dmeans2D
  g[1] = torch::empty({P, 3}, f32);      // dL_dcolors
  g[2] = torch::empty({P, F}, f32);      // dL_dlanguage
  g[3] = torch::empty({P, 1}, f32);      // dL_dopacity
  g[4] = torch::empty({P, 3}, f32);      // dL_dmeans3D
  g[5] = torch::empty({P, 6}, f32);      // dL_dcov3D
  g[6] = torch::empty({P, M, 3}, f32);   // dL_dsh
  g[7] = torch::empty({P, 3}, f32);      // dL_dscales
  g[8] = torch::empty({P, 4}, f32);      // dL_drotations
  g[9] = torch::empty({P, 6}, f32);      // dL_dtau
  g[10] = torch::empty({6}, f32);        // dL_dtau_sum
  if (want_internal) {
    g[11] = torch::empty({P, 2, 2}, f32);  // dL_dconic
    g[12] = torch::empty({P, 1}, f32);     // dL_ddepths
  }
  Tensor dc = prep(dL_dout_color, "dL_dout_color"), dl = prep(dL_dout_language, "dL_dout_language"),
         dd = prep(dL_dout_depth, "dL_dout_depth");
  Tensor rad = radii.contiguous();
  // Row scratch: one partial-gradient row per live (instance, 64-pixel slot) pair (reference mode of 15x15 tiles: per
  // packed survivor wave).  The caching allocator reuses blocks stream-ordered, so the tensor may die at return.
  // The forward posts the frame's exact row count to the host; a caller that is ahead of the GPU waits for it (the GPU is
  // busy with the forward meanwhile) when the bound L <= slots * R would cost more than 64 MB (olsr_backward_rows).
  // Round 5: a training loop reaches this point while the forward is still executing.  Waiting for its posted count and only
  // THEN launching left the GPU idle between the forward's last kernel and the backward's first (13 us per frame at config 3).
  // Now: the count if it is there; else a GUESS — 1.5 x the rows per instance of the frame this process verified last — the
  // backward is launched with it, and while the GPU works the count is awaited and compared: a guess that was too small (that
  // backward wrote zeros everywhere and said so) is followed by a second, exact backward on the same stream, which overwrites
  // every output.  The result is always the exact one; the redo is the price of a scene whose rows per instance jumped by half.
  const bool packed = (bwd_mode == OLSR_BWD_REFERENCE && tile == 15);
  const int64_t bound = static_cast<int64_t>(R > 0 ? R : 0) * (packed ? 2 : 4);
  int64_t rows = olsr_live_rows(rows_token, packed ? 1 : 0);
  bool guessed = false;
  if (rows < 0 || rows > bound) {
    const float ratio = g_rows_per_instance[packed ? 1 : 0].load(std::memory_order_relaxed);
    if (olsr_live_rows_overwritten(rows_token)) {
      rows = bound;  // (the slot belongs to a later forward: no count will ever arrive — no guess, no wait, no second backward)
    } else if (rows_token > 0 && ratio > 0.f && R > 0 && olsr_backward_scratch_bytes(bound, F) > (static_cast<size_t>(64) << 20)) {
      // (rounded up to a multiple of 128 Ki rows — 14 MB at F = 15: a size that changed a little from frame to frame, with the
      //  decaying ratio, made the caching allocator cut a new block every frame, and every few frames that is a hipMalloc)
      rows = static_cast<int64_t>(1.5 * static_cast<double>(ratio) * R) + 65536;
      rows = std::min<int64_t>(bound, (rows + 131071) / 131072 * 131072);
      guessed = true;
    } else {
      pybind11::gil_scoped_release nogil;
      rows = olsr_backward_rows(rows_token, packed ? 1 : 0, R, F);  // (waits when the bound would cost more than 64 MB)
    }
  }
  auto launch = [&](int64_t nrows) {
    Tensor scratch = torch::empty({static_cast<int64_t>(olsr_backward_scratch_bytes(nrows, F))},
                                  means3D.options().dtype(torch::kUInt8));
    int rc;
    {
      pybind11::gil_scoped_release nogil;  // pure launches, no callback into Python
      rc = olsr_backward(&sc.s, P ? rad.data_ptr<int32_t>() : nullptr, geomBuffer.data_ptr(), R, binningBuffer.data_ptr(),
                         imageBuffer.data_ptr(), nullptr, nullptr, scratch.data_ptr(), nrows, fp(dc), fp(dl), fp(dd),
                         fpw(g[0]), fpw(g[11]), fpw(g[3]), fpw(g[1]), fpw(g[2]), fpw(g[12]), fpw(g[4]), fpw(g[5]),
                         fpw(g[6]), fpw(g[7]), fpw(g[8]), fpw(g[9]), fpw(g[10]), nullptr, nullptr, stream_of(means3D));
    }
    check(rc);
  };
  launch(rows);
  int64_t exact = guessed ? -1 : ((rows < bound || bound == 0) ? rows : -1);
  if (guessed) {
    {
      pybind11::gil_scoped_release nogil;  // (the GPU is busy with the forward and the backward just queued)
      exact = olsr_live_rows_wait(rows_token, packed ? 1 : 0, 20000);
    }
    if (exact < 0 || exact > rows) {
      g_rows_redone.fetch_add(1, std::memory_order_relaxed);
      launch((exact < 0 || exact > bound) ? bound : exact);
    }
  }
  if (exact >= 0 && R > 0) {
    // a slowly decaying maximum: the views of a mapping window differ in rows per instance (the arc views of the benchmark
    // by 60 %), and a guess sized from the lightest of them would be redone for every heavier one
    const float now = static_cast<float>(exact) / static_cast<float>(R);
    const float old = g_rows_per_instance[packed ? 1 : 0].load(std::memory_order_relaxed);
    g_rows_per_instance[packed ? 1 : 0].store(std::max(now, 0.9f * old + 0.1f * now), std::memory_order_relaxed);
  }
  return g;
}

// markVisible, DGR/rasterize_points.cu:457-476
Tensor mark_visible(const Tensor &means3D, const Tensor &viewmatrix, const Tensor &projmatrix) {
  TORCH_CHECK(means3D.is_cuda(), "means3D must live on the GPU: this rasterizer has no CPU path");
  const c10::DeviceGuard guard(means3D.device());
  const int64_t P = means3D.size(0);
  Tensor present = torch::zeros({P}, means3D.options().dtype(torch::kBool));
  if (P) {
    Tensor m = prep(means3D, "means3D"), v = prep(viewmatrix, "viewmatrix"), p = prep(projmatrix, "projmatrix");
    check(olsr_mark_visible(static_cast<int32_t>(P), fp(m), fp(v), fp(p),
                            reinterpret_cast<uint8_t *>(present.data_ptr<bool>()), stream_of(means3D)));
  }
  return present;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "compiled torch binding of libolsr.so (same functions as the reference's diff_gaussian_rasterization._C)";
  m.def("forward", &forward);
  m.def("backward", &backward);
  m.def("mark_visible", &mark_visible);
  m.def("last_forward_token", []() { return static_cast<int>(olsr_last_forward_token()); });
  m.def("debug_rows_ratio", [](bool packed, float ratio) {  // ratio < 0: read only.  Returns (ratio in force, backwards redone)
    if (ratio >= 0.f) g_rows_per_instance[packed ? 1 : 0].store(ratio);
    return std::make_tuple(g_rows_per_instance[packed ? 1 : 0].load(), g_rows_redone.load());
  });
  m.def("version", []() { return std::string(olsr_version()); });
}
