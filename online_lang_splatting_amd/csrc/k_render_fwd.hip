// k_render_fwd.hip — forward front-to-back alpha compositing of RGB + depth + F language channels.
//
// Replaces renderCUDA / language_renderCUDA (CR/forward.cu:515-644, 377-513).
//
// MI355X mapping: ONE wave64 per logical tile.  The reference's TILE x TILE pixel block
// (15x15 = 225 thread ranks, rank = ty*TILE + tx) is folded onto 64 lanes x 4 pixel slots
// (rank = slot*64 + lane), so
//   * the wave never needs a workgroup barrier (the reference syncs 225 threads twice per batch),
//   * the wave-uniform splat data (xy, conic, opacity, features) is read from LDS once per
//     splat and reused for 4 pixels,
//   * each lane carries 4 independent transmittance chains (ILP for the exp / fma latency),
//   * tile-wide decisions (all pixels done, "did any pixel use this splat") are 64-bit ballots.
// Splat data is staged through LDS in batches of 64 with one coalesced gather per lane.
// Per-pixel arithmetic keeps the reference's operation order (see olsr_device.h).
//
// Besides the image outputs the kernel records, per (tile, splat) instance, whether any pixel
// of the tile blended it (flags[], indexed by emission position): the backward composite only
// visits those instances, and in REFERENCE mode that flag is exactly the tile-wide
// "skip_counter != BLOCK_SIZE" predicate of CR/backward.cu:1087-1093.
#include "olsr_device.h"
#include "olsr_kernels.h"

namespace olsr {

template <int TILE, int F>
__global__ __launch_bounds__(64) void render_fwd_kernel(
    const u32* __restrict__ ranges, const u32* __restrict__ point_list, const u32* __restrict__ src, int W, int H,
    int gx, int ntiles, const float* __restrict__ means2D, const float* __restrict__ conic_opacity,
    const float* __restrict__ depths, const float* __restrict__ colors, const float* __restrict__ lang,
    const float* __restrict__ bg, float* __restrict__ final_T, u32* __restrict__ n_contrib,
    float* __restrict__ out_color, float* __restrict__ out_lang, float* __restrict__ out_depth,
    float* __restrict__ out_opacity, int32_t* __restrict__ n_touched, uint8_t* __restrict__ flags) {
  constexpr int BS = TILE * TILE;
  constexpr int SLOTS = (BS + 63) / 64;
  constexpr int FR = feat_row(F);
  constexpr int NA = 4 + F;  // r g b depth lang[F]

  __shared__ float2 s_xy[64];
  __shared__ float4 s_co[64];
  __shared__ __attribute__((aligned(16))) float s_feat[64 * FR];
  __shared__ u32 s_id[64];
  __shared__ u32 s_src[64];

  const int tile_id = xcd_remap((int)blockIdx.x, ntiles);
  const int lane = threadIdx.x;
  const int bx = tile_id % gx, by = tile_id / gx;
  const u32 r0 = ranges[2 * tile_id], r1 = ranges[2 * tile_id + 1];
  const int n = (int)(r1 - r0);

  float pixfx[SLOTS], pixfy[SLOTS], T[SLOTS];
  bool inside[SLOTS], done[SLOTS];
  u32 last_contributor[SLOTS];
  u32 pix_id[SLOTS];
  float acc[SLOTS][NA];
#pragma unroll
  for (int q = 0; q < SLOTS; ++q) {
    const int rank = q * 64 + lane;
    const int px = bx * TILE + rank % TILE, py = by * TILE + rank / TILE;
    inside[q] = (rank < BS) && (px < W) && (py < H);
    done[q] = !inside[q];
    pixfx[q] = (float)px;
    pixfy[q] = (float)py;
    pix_id[q] = (u32)W * (u32)py + (u32)px;
    T[q] = 1.0f;
    last_contributor[q] = 0;
#pragma unroll
    for (int k = 0; k < NA; ++k) acc[q][k] = 0.0f;
  }

  for (int base = 0; base < n; base += 64) {
    bool all_done = true;
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) all_done = all_done && done[q];
    if (wave_all(all_done)) break;

    const int cnt = min(64, n - base);
    __syncthreads();  // single-wave workgroup: orders this batch's LDS writes after the last batch's reads
    if (lane < cnt) {
      const u32 sp = r0 + (u32)base + (u32)lane;
      const u32 gid = point_list[sp];
      s_id[lane] = gid;
      s_src[lane] = src[sp];
      s_xy[lane] = reinterpret_cast<const float2*>(means2D)[gid];
      s_co[lane] = reinterpret_cast<const float4*>(conic_opacity)[gid];
      float* fr = &s_feat[lane * FR];
      fr[0] = colors[3 * (size_t)gid + 0];
      fr[1] = colors[3 * (size_t)gid + 1];
      fr[2] = colors[3 * (size_t)gid + 2];
      fr[3] = depths[gid];
#pragma unroll
      for (int ch = 0; ch < F; ++ch) fr[4 + ch] = lang[(size_t)gid * F + ch];
    }
    __syncthreads();

    u64 active_mask = 0ull;  // bit j: some pixel of the tile blended splat j of this batch
    u32 touch = 0;           // lane j: #pixels with test_T > 0.5 for splat j
    for (int j = 0; j < cnt; ++j) {
      const float2 xy = s_xy[j];
      const float4 co = s_co[j];
      const float* fr = &s_feat[j * FR];
      u64 contrib_any = 0ull;
      u32 touch_cnt = 0;
      bool live = false;
#pragma unroll
      for (int q = 0; q < SLOTS; ++q) {
        bool contrib = false, touched = false;
        if (!done[q]) {
          const float dx = xy.x - pixfx[q], dy = xy.y - pixfy[q];
          const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
          if (!(power > 0.0f)) {
            const float alpha = fminf_ref(0.99f, co.w * pinned_expf(power));
            if (!(alpha < 1.0f / 255.0f)) {
              const float test_T = T[q] * (1 - alpha);
              if (test_T < 0.0001f) {
                done[q] = true;
              } else {
#pragma unroll
                for (int k = 0; k < NA; ++k) acc[q][k] += fr[k] * alpha * T[q];
                touched = test_T > 0.5f;
                T[q] = test_T;
                last_contributor[q] = (u32)(base + j + 1);
                contrib = true;
              }
            }
          }
        }
        contrib_any |= ballot(contrib);
        touch_cnt += (u32)__popcll(ballot(touched));
        live = live || !done[q];
      }
      if (contrib_any) active_mask |= (1ull << j);
      if (touch_cnt) touch = (lane == j) ? touch_cnt : touch;
      if (!wave_any(live)) break;
    }

    if (lane < cnt) {
      if ((active_mask >> lane) & 1ull) flags[s_src[lane]] = 1;
      if (touch) atomicAdd(&n_touched[s_id[lane]], (int)touch);
    }
  }

  const size_t HW = (size_t)H * W;
#pragma unroll
  for (int q = 0; q < SLOTS; ++q) {
    if (inside[q]) {
      const u32 p = pix_id[q];
      final_T[p] = T[q];
      n_contrib[p] = last_contributor[q];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) out_color[ch * HW + p] = acc[q][ch] + T[q] * bg[ch];
      out_depth[p] = acc[q][3];
      out_opacity[p] = 1 - T[q];
      if constexpr (F > 0) {
#pragma unroll
        for (int ch = 0; ch < F; ++ch) out_lang[ch * HW + p] = acc[q][4 + ch];
      }
    }
  }
}

template <int TILE, int F>
static void launch_fwd_t(const olsr_scene& s, const FrameDims& d, const GeometryState& g, const BinningState& b,
                         const u32* src, const ImageState& im, float* out_color, float* out_language,
                         float* out_depth, float* out_opacity, int32_t* n_touched, hipStream_t st) {
  const float* colors = s.colors_precomp ? s.colors_precomp : g.rgb;
  render_fwd_kernel<TILE, F><<<d.ntiles, 64, 0, st>>>(im.ranges, b.point_list, src, d.W, d.H, d.gx, d.ntiles, g.means2D,
                                                      g.conic_opacity, g.depths, colors, s.language_precomp,
                                                      s.background, im.final_T, im.n_contrib, out_color, out_language,
                                                      out_depth, out_opacity, n_touched, b.flags);
}

template <int TILE>
static void launch_fwd_f(const olsr_scene& s, const FrameDims& d, const GeometryState& g, const BinningState& b,
                         const u32* src, const ImageState& im, float* oc, float* ol, float* od, float* oo, int32_t* nt,
                         hipStream_t st) {
  switch (s.F) {
    case 0: launch_fwd_t<TILE, 0>(s, d, g, b, src, im, oc, ol, od, oo, nt, st); break;
    case 3: launch_fwd_t<TILE, 3>(s, d, g, b, src, im, oc, ol, od, oo, nt, st); break;
    case 15: launch_fwd_t<TILE, 15>(s, d, g, b, src, im, oc, ol, od, oo, nt, st); break;
    case 16: launch_fwd_t<TILE, 16>(s, d, g, b, src, im, oc, ol, od, oo, nt, st); break;
    case 32: launch_fwd_t<TILE, 32>(s, d, g, b, src, im, oc, ol, od, oo, nt, st); break;
    default: break;
  }
}

void launch_render_forward(const olsr_scene& s, const FrameDims& d, const GeometryState& g, const BinningState& b,
                           const ImageState& im, float* out_color, float* out_language, float* out_depth,
                           float* out_opacity, int32_t* n_touched, hipStream_t st) {
  if (d.tile == 15)
    launch_fwd_f<15>(s, d, g, b, b.src, im, out_color, out_language, out_depth, out_opacity, n_touched, st);
  else
    launch_fwd_f<16>(s, d, g, b, b.src, im, out_color, out_language, out_depth, out_opacity, n_touched, st);
}

}  // namespace olsr
