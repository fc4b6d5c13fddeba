// k_render_bwd.hip — backward of the alpha compositing: per-pixel gradients -> per-(tile, splat)
// partial-gradient rows.
//
// Replaces renderCUDA / language_render_cuda (backward) with render_cuda_reduce_sum
// (CR/backward.cu:706-930, 932-1201, 684-702).
//
// The reference spends, per splat per tile, 3 block barriers + a shared-memory atomic per
// skipping thread + an 8-barrier shared-memory tree over 225 threads + up to 25 global float
// atomics.  MI355X mapping (same as the forward: one workgroup per tile, 4 wave64s, one pixel
// per lane, wave w = the forward's slot w — one quadrant of the tile, slot_rank in olsr_device.h):
//   * no barrier per splat: the forward composite recorded, per (tile, splat) instance, which
//     slots blended it (flags[] bit w).  A wave skips instances its slot never touched with a
//     uniform branch, and the four waves of a tile only meet when the next batch of 128 list
//     entries is staged into LDS;
//   * the per-splat reduction never leaves the wave and never touches LDS: gfx950's
//     v_permlane32_swap / v_permlane16_swap fold the lane dimension for two values per swap,
//     DPP row rotations finish inside rows of 16 (10 instructions per 4 values).  The totals are
//     not gathered into consecutive lanes: the lane that already holds total j stores row element
//     j (12 lanes for the 10 reference-mode values), so the row costs one store and no permute;
//   * the kernel is VALU-issue bound, so the per-visit instruction count is what is tuned: the
//     skip-dependent values are three masked factors (G, dL_dalpha, alpha*T) instead of ~30
//     zero-initialised registers, the language dot product runs on packed fp32 FMAs, 1/(1-alpha)
//     is v_rcp + one Newton step, and the reference-mode language row (tile rank 0 only) is ONE
//     scalar broadcast times a per-lane constant;
//   * no float atomics: rows are compacted in emission order (rowbase = exclusive scan of
//     popcount(flags)), so the per-Gaussian reduction (k_preprocess_bwd.hip) streams one dense,
//     contiguous run of rows per Gaussian and the gradients are bit-reproducible from run to run.
//
// MODE = OLSR_BWD_REFERENCE reproduces the shipped reference bit-compatibly in structure:
//   - only the ranks that survive the 225-lane integer-halving tree contribute to
//     mean2D / conic / opacity / colour / depth (ref_survives<15>),
//   - language gradients are taken from rank 0 only,
//   - the language recursion (accum_rec_F, last_language_feature) runs for every pixel of the
//     tile whenever the tile as a whole does not skip the splat.
// MODE = OLSR_BWD_EXACT sums all pixels and guards the language recursion like colour.
#include <type_traits>

#include "olsr_device.h"
#include "olsr_kernels.h"

namespace olsr {

// Which thread ranks reach element 0 of render_cuda_reduce_sum's tree when g.size() == TILE*TILE:
// all 256 for 16x16 tiles, the 128 of ref15_survives (olsr_device.h) for 15x15 tiles.
template <int TILE>
__device__ __forceinline__ bool ref_survives(int rank) {
  if constexpr (TILE == 16) {
    return true;
  } else {
    static_assert(TILE == 15, "closed form derived for 15x15 and 16x16 tiles only");
    return ref15_survives(rank);
  }
}

// lanes (lane & 15) that hold a total of one of the first `ntri` merged triples of groups (row_sums3, see the kernel)
__device__ __forceinline__ constexpr bool triple_lane(int lg, int ntri) {
  return ((lg & 12) == 4 || (lg & 12) == 12 || (lg & 12) == 0) && (lg & 3) < ntri;
}

__device__ __forceinline__ constexpr bool own_lanes_clear(int first, int end, int ntri) {
  for (int lg = first; lg < end; ++lg)
    if (triple_lane(lg, ntri)) return false;
  return true;
}

constexpr int BWD_BATCH = 128;
#ifndef OLSR_BWD_LDS_REDUCE
#define OLSR_BWD_LDS_REDUCE 1  // fold the lanes of the per-splat sums through LDS (olsr_device.h) instead of permlane swaps
#endif
#ifndef OLSR_BWD_MFMA_REDUCE
#define OLSR_BWD_MFMA_REDUCE 0  // 1: sum the ten per-splat values over the wave with MFMAs (round-4 experiment, slower: see the kernel)
#endif
typedef float bwd_f32x4 __attribute__((ext_vector_type(4)));
#ifndef OLSR_BWD_SCALAR_MAX_F
#define OLSR_BWD_SCALAR_MAX_F 32
#endif
#ifndef OLSR_BWD_SCALAR_VALUE
#define OLSR_BWD_SCALAR_VALUE 1  // the value path on scalar fp32 (build with -fno-slp-vectorize); 0: packed pairs (rounds 1-3)
#endif
// packed fp32 is half rate on gfx950: two lanes' work in twice the time, plus the moves that pair the operands up
struct bv2s {
  float x, y;
};
// (value path only: its products and sums may contract to FMAs like the packed form's did — the flag travels with the
//  operations when they are inlined; the decision path never touches this type)
#pragma clang fp contract(fast)
__device__ __forceinline__ bv2s operator+(bv2s a, bv2s b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ bv2s operator-(bv2s a, bv2s b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ bv2s operator*(bv2s a, bv2s b) { return {a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ bv2s operator*(bv2s a, float b) { return {a.x * b, a.y * b}; }
__device__ __forceinline__ bv2s operator-(bv2s a) { return {-a.x, -a.y}; }
__device__ __forceinline__ bv2s& operator+=(bv2s& a, bv2s b) {
  a.x += b.x;
  a.y += b.y;
  return a;
}
#pragma clang fp contract(off)
// scalar for every F (config 3: 0.1907 -> 0.1763 ms, exact mode 1 108 -> 1 155 fps, tracking iteration 0.548 -> 0.538 ms;
// config 5, F = 32: 0.525 -> 0.506 ms).  OLSR_BWD_SCALAR_VALUE=0 / OLSR_BWD_SCALAR_MAX_F restore the packed pairs.
template <int F>
struct bwd_pair {
  typedef typename std::conditional<(OLSR_BWD_SCALAR_VALUE != 0) && (F <= OLSR_BWD_SCALAR_MAX_F), bv2s, v2f>::type type;
};

// PACKED (reference mode, 15x15 tiles): the workgroup is the 128 survivors of the reference's reduction
// tree in two full waves (ref15_rank_of_packed); the 97 other pixels of the tile are not evaluated at all —
// nothing they compute reaches an output of the reference's backward.
#ifndef OLSR_BWD_MIN_WAVES
#define OLSR_BWD_MIN_WAVES 5  // waves per SIMD the packed reference-mode instantiations with F <= 16 are compiled for
#endif
template <int TILE, int F, int MODE, bool PACKED>
__global__ __launch_bounds__(PACKED ? 128 : 256, (PACKED && F <= 16) ? OLSR_BWD_MIN_WAVES : 1) void render_bwd_kernel(
    const u32* __restrict__ ranges, const u32* __restrict__ inst_gid, const u32* __restrict__ src,
    const uint8_t* __restrict__ flags, const u32* __restrict__ rowbase, const int32_t* __restrict__ counters,
    const u32* __restrict__ tile_order, int W, int H, int gx, int ntiles, const float* __restrict__ bg,
    const float* __restrict__ means2D, const float* __restrict__ conic_opacity, const float* __restrict__ colors,
    const float* __restrict__ lang, const float* __restrict__ depths, const float* __restrict__ final_Ts,
    const u32* __restrict__ n_contrib, const float* __restrict__ dL_dpixels, const float* __restrict__ dL_dpixels_lang,
    const float* __restrict__ dL_dpixels_depth, float* __restrict__ rows, int rows_stamp) {
#ifdef OLSR_COMPOSITE_VGPR_FLOOR
  asm volatile("; vgpr floor" ::: OLSR_COMPOSITE_VGPR_FLOOR);  // (experiment: fewer resident waves, room for other frames' kernels)
#endif
  typedef typename bwd_pair<F>::type bv2;
  constexpr int BS = TILE * TILE;
  constexpr int FR = feat_row(F);
  constexpr int ROW = grad_row(F);
  constexpr bool REF = (MODE == OLSR_BWD_REFERENCE);
  constexpr int NV = REF ? 10 : 10 + F;  // values that go through the wave reduction
  constexpr int NG4_ = NV / 4;           // groups of four values reduced together (wave_reduce4)
  constexpr int REM = (NV % 4 == 3) ? 0 : NV % 4;  // 1-2 left-over values: wave_reduce2
  constexpr int NG4 = NG4_ + ((NV % 4 == 3) ? 1 : 0);  // (three left over: a four-tree with one zero)
  constexpr int NVP = 4 * NG4 + (REM ? 2 : 0);    // sum[] entries incl. zero padding
  constexpr int FX = (F > 0) ? F : 1;
  constexpr int F2 = (F + 1) / 2;        // packed pairs of language channels
  constexpr int F2X = (F2 > 0) ? F2 : 1;
  constexpr int B = BWD_BATCH;
  constexpr int NT = PACKED ? 128 : 256;  // threads
  constexpr int NWV = NT / 64;            // waves
  static_assert(ROW <= 64, "one lane per row element");
  static_assert(!PACKED || (REF && TILE == 15), "survivor packing is the reference mode of 15x15 tiles");

  __shared__ float2 s_xy[B];
  __shared__ float4 s_co[B];
  __shared__ __attribute__((aligned(16))) float s_feat[B * FR];
  __shared__ u32 s_row[B];   // first compact row of the instance
  __shared__ uint8_t s_flag[B];
  __shared__ int s_kmax[NWV];
  // Where the lanes of the per-splat sums are folded through LDS (olsr_device.h): the reference mode's ten values with at
  // most 16 language channels.  Measured at the other instantiations, the extra LDS costs a resident workgroup (F = 32:
  // 0.567 -> 0.602 ms) or the seven exchanges of the exact mode's 25 values cost more than the swaps (-1.7 %).
  constexpr bool LDSR = (OLSR_BWD_LDS_REDUCE != 0) && REF && (F <= 16);
  // The reference mode's three folded registers (two groups of four values + the pair) share one in-row reduction
  // (row_sums3: 7 DPP operations instead of 12), whichever way they were folded ...
  constexpr bool MERGED = NG4 == 2 && REM == 2;  // (NV = 10: the reference mode, and the exact mode without language)
  // ... and so do, three at a time, the groups of four of the other instantiations (exact mode, F = 15: six groups = two
  // triples + one value on its own).  Triple j leaves its totals in the lanes (lane & 15) == 4 + j, 12 + j and j.
  constexpr int NTRI = MERGED ? 0 : NG4 / 3;
  static_assert(NTRI <= 4 && NG4 + 1 <= 16 && own_lanes_clear(3 * NTRI, NG4 + (REM ? 1 : 0), NTRI),
                "the lanes of the groups reduced on their own must stay clear of the triples' lanes");
  // The ten-value sums (NV == 10) on the MATRIX pipe (round 4): the cross-lane sum of a value is a
  // product with a constant matrix, and v_mfma_f32_16x16x4_f32 is exact fp32.  Value m goes in as the A operand (lane l holds
  // A[i = l % 16][k = l / 16]) against the selector B_m[k][j] = (j == m): D[i][m] += sum_k v_m[16 k + i], so ten chained MFMAs
  // leave, in the lanes of column m, the sixteen partial sums of value m (four registers x four rows of lanes).  Three adds
  // fold the registers — for all ten values at once, every column holds another value — and an eleventh MFMA against a ones
  // operand folds the four rows: every lane of column m then holds total m, and lanes 0..9 store row elements 0..9 as ONE
  // contiguous 40-byte store.  VALU work of the reduction: 3 adds, against 9 adds + 7 DPP + 2 selects + 8 LDS instructions
  // of the LDS fold.  MEASURED (config 3, profiles/r4_experiments.json): 0.188 -> 0.262 ms, the tracking iteration's F = 0
  // backward 0.143 -> 0.249 ms; parity suite green.  The f32-input MFMA runs at the f32 VECTOR rate (64 FLOP / clk / SIMD,
  // MI355X_MICROARCH.md) — 11 x 32 cycles per visit that behave like VALU time, not like time on an idle pipe, plus a chain
  // of ten dependent accumulations (40 cycles each) that an in-order wave sits out.  Kept behind the macro, off.
  // (A non-finite value would poison all ten sums of its visit (0 x inf), not only its own.)
  constexpr bool MRED = (OLSR_BWD_MFMA_REDUCE != 0) && MERGED;
  __shared__ __attribute__((aligned(16))) float s_red[(LDSR && !MRED) ? NWV * 256 : 4];  // 1 KB per wave: the exchange

  // workgroup b runs on XCD b % 8 and takes the (b / 8)-th heaviest tile of that XCD's chunk
  const int tile_id = (int)tile_order[xcd_remap((int)blockIdx.x, ntiles)];
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bx = tile_id % gx, by = tile_id / gx;
  const u32 r0 = ranges[2 * tile_id], r1 = ranges[2 * tile_id + 1];
  if (r1 <= r0) return;
  if (frame_unusable(counters, rows_stamp)) return;  // row scratch too small / synchronisation error / rows compacted for another scratch (reported to the caller): write nothing
  const size_t HW = (size_t)H * W;

  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const bool has_bg = (bg0 != 0.0f) || (bg1 != 0.0f) || (bg2 != 0.0f);
  const float ddelx_dx = 0.5f * W;
  const float ddely_dy = 0.5f * H;

  // ---- per-pixel state (one pixel per lane; thread rank = ty*TILE + tx as in the reference) ----
  // Language channels are carried in "dot form": the reference keeps accum_rec_F[F] and
  // last_language_feature[F] per thread only to evaluate sum_ch (f - accum_rec_F)[ch] * dL_dF[ch];
  // with A = <accum_rec_F, dL_dF> and D = <f, dL_dF> the recursion
  //   accum_rec_F <- last_alpha * last_F + (1 - last_alpha) * accum_rec_F     (CR/backward.cu:1132)
  // becomes A <- last_alpha * D_last + (1 - last_alpha) * A and the contribution is D - A.
  // Algebraically identical, 2 registers per pixel instead of 2F.
  const int rank = PACKED ? ref15_rank_of_packed(tid) : slot_rank<TILE>(tid);  // (unpacked: the forward's slots)
  const int px = bx * TILE + rank % TILE, py = by * TILE + rank / TILE;
  const bool inside = (rank < BS) && (px < W) && (py < H);
  const bool surv = (REF && !PACKED) ? ref_survives<TILE>(rank) : true;
  const float pixfx = (float)px, pixfy = (float)py;
  const size_t pix = (size_t)W * py + px;
  const float T_final = inside ? final_Ts[pix] : 0.f;
  float T = T_final;
  const int last_contributor = inside ? (int)n_contrib[pix] : 0;
  float last_alpha = 0.f;
  // Colour / depth "behind" a splat: the reference carries (last_alpha, last_color, accum_rec) and forms
  // accum_rec <- last_alpha * last_color + (1 - last_alpha) * accum_rec when it reaches the next splat
  // (CR/backward.cu:1110-1123).  The same quantity in running form: Z = colour composited from everything behind,
  // used as accum_rec and then advanced, Z <- Z + alpha * (c - Z).  Algebraically identical, 4 state registers
  // instead of 8, one operation less per channel (value path: no decision depends on it).
  bv2 Z01 = {0.f, 0.f}, Z23 = {0.f, 0.f};  // {r, g} and {b, depth} composited from everything behind
  float dLc[3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) dLc[ch] = inside ? dL_dpixels[ch * HW + pix] : 0.f;
  float bg_dot = 0.f;
  bg_dot += bg0 * dLc[0];
  bg_dot += bg1 * dLc[1];
  bg_dot += bg2 * dLc[2];
  // the background's share of dL_dalpha, (-T_final / (1 - alpha)) <bg, dL_dpixel> (CR/backward.cu:1119-1125), as ONE fma per visit
  // against the visit's 1 / (1 - alpha): a per-pixel constant, zero without a background
  const float bg_term = has_bg ? -T_final * bg_dot : 0.f;
  const float dLd = (inside && dL_dpixels_depth != nullptr) ? dL_dpixels_depth[pix] : 0.f;  // (NULL: no depth term in the loss)
  const bv2 dL01 = {dLc[0], dLc[1]}, dL23 = {dLc[2], dLd};  // the pixel's cotangents, in the same pairs
  float A_f = 0.f, D_last = 0.f, dLf[FX];
  bool seen_mine = false;  // wave-uniform: an entry of this wave's own has been visited
#pragma unroll
  for (int ch = 0; ch < FX; ++ch) dLf[ch] = (F > 0 && inside) ? dL_dpixels_lang[ch * HW + pix] : 0.f;
  bv2 dLf2[F2X];  // the same cotangents in pairs, for packed fp32 math
#pragma unroll
  for (int k2 = 0; k2 < F2X; ++k2) {
    dLf2[k2].x = (2 * k2 < F) ? dLf[(2 * k2 < F) ? 2 * k2 : 0] : 0.f;
    dLf2[k2].y = (2 * k2 + 1 < F) ? dLf[(2 * k2 + 1 < F) ? 2 * k2 + 1 : 0] : 0.f;
  }

  // ---- which row element this lane stores ----------------------------------------------------
  // wave_reduce4 leaves the totals of group g in the four 16-lane rows of its result (row r holds
  // value perm(r)), wave_reduce2 in rows 1 and 3.  The lane with (lane & 15) == g stores group g's
  // total of its row; the remaining row elements (reference mode: the language channels) are dealt
  // to the free lanes in ascending order.
  const int lg = lane & 15, lr = lane >> 4;
  int role = -1;
  float sel[MRED ? 10 : 1];  // MRED: the selectors B_m (1 in the lanes of column m)
#pragma unroll
  for (int m = 0; m < (MRED ? 10 : 1); ++m) sel[m] = (lg == m) ? 1.0f : 0.0f;
  if constexpr (MRED) {
    role = (lane < NV) ? lane : -1;  // every lane of column m ends with total m; row 0's lanes store
  } else if constexpr (MERGED) {
    // row_sums3: (lane & 15) == 4 -> group 0, == 12 -> group 1, == 0 -> the pair (rows 1 and 3 store it)
    if (lg == 4) role = (((lr & 1) << 1) | (lr >> 1));
    if (lg == 12) role = 4 + (((lr & 1) << 1) | (lr >> 1));
    // (the pair: the LDS fold leaves a in rows 0-1 and b in rows 2-3, the swap fold a in row 0 and b in row 2)
    if (lg == 0 && (lr & 1) == (LDSR ? 1 : 0)) role = 8 + (lr >> 1);
  } else {
    const int perm = ((lr & 1) << 1) | (lr >> 1);
#pragma unroll
    for (int j = 0; j < NTRI; ++j) {
      if (lg == 4 + j) role = 4 * (3 * j) + perm;
      if (lg == 12 + j) role = 4 * (3 * j + 1) + perm;
      if (lg == j) role = 4 * (3 * j + 2) + perm;
    }
    // groups on their own and the left-over pair: the lanes 3 NTRI .. (clear of 0..NTRI-1, 4.., 12.. — checked below)
    if (lg >= 3 * NTRI && lg < NG4 && !triple_lane(lg, NTRI)) role = 4 * lg + perm;
    if (REM > 0 && lg == NG4 && (lr & 1)) role = 4 * NG4 + (lr >> 1);
  }
  if (role >= NV) role = -1;
  const bool in_hi8 = (lane & 8) != 0, in_0to3 = (lane & 12) == 0;
  bool lang_lane = false;
  float dLf0_lane = 0.f;
  if constexpr (REF && F > 0) {
    const u64 freem = ballot(role < 0);
    const int nth = (int)__popcll(freem & ((1ull << lane) - 1ull));
    if (role < 0 && nth < F) {
      role = 10 + nth;
      lang_lane = true;
    }
#pragma unroll
    for (int ch = 0; ch < F; ++ch) {
      const float v = lane_read(dLf[ch], 0);  // the tile's rank-0 pixel (wave 0, lane 0)
      if (lang_lane && role == 10 + ch) dLf0_lane = v;
    }
    if (w != 0) dLf0_lane = 0.f;
  }

  // entries at list positions >= max(last_contributor) are skipped by every pixel of the tile
  int kmax = last_contributor;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) kmax = max(kmax, __shfl_xor(kmax, m));
  if (lane == 0) s_kmax[w] = kmax;
  __syncthreads();
  kmax = s_kmax[0];
#pragma unroll
  for (int k = 1; k < NWV; ++k) kmax = max(kmax, s_kmax[k]);

  for (int kstart = kmax - 1; kstart >= 0; kstart -= B) {
    const int cnt = min(B, kstart + 1);
    __syncthreads();
    {
      const int e = tid & (B - 1);
      if (e < cnt) {
        const u32 sp = r0 + (u32)(kstart - e);
        const u32 u = src[sp];        // emission index of the instance
        const u32 gid = inst_gid[u];  // its Gaussian
        if (PACKED || tid < B) {
          s_row[e] = rowbase[u];
          s_flag[e] = flags[u];
          s_xy[e] = reinterpret_cast<const float2*>(means2D)[gid];
          s_co[e] = reinterpret_cast<const float4*>(conic_opacity)[gid];
        }
        if (PACKED || tid >= B) {
          float* fr = &s_feat[e * FR];
          fr[0] = colors[3 * (size_t)gid + 0];
          fr[1] = colors[3 * (size_t)gid + 1];
          fr[2] = colors[3 * (size_t)gid + 2];
          fr[3] = depths[gid];
#pragma unroll
          for (int ch = 0; ch < F; ++ch) fr[4 + ch] = lang[(size_t)gid * F + ch];
#pragma unroll
          for (int ch = 4 + F; ch < FR; ++ch) fr[ch] = 0.f;  // the packed dot product reads the padding
        }
      }
    }
    __syncthreads();

    // From here to the next batch the four waves run independently: no barrier per splat.
    // The flags of the staged entries become wave-uniform 64-bit masks (one ballot per half batch): which entries the
    // tile as a whole does not skip (bits 0-3: the forward's slots) and which ones THIS wave blended (bits 4-5: the
    // packed survivor waves).  The loop then walks set bits with scalar instructions; an entry nobody blended — most
    // of every list — costs nothing (it used to cost a dependent LDS read and a v_readfirstlane each).
    static_assert(B == 128, "two 64-bit masks per batch");
    u64 m_any[2], m_mine[2], m_w0[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = lane + 64 * h;
      const u32 f = (e < cnt) ? s_flag[e] : 0u;
      m_any[h] = ballot((f & 15u) != 0u);
      m_mine[h] = ballot(((f >> (PACKED ? 4 + w : w)) & 1u) != 0u);
      m_w0[h] = ballot(((f >> 4) & 1u) != 0u);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
    // reference mode with language channels: the recursion also advances on entries another wave blended
    u64 todo = (REF && F > 0) ? m_any[h] : m_mine[h];
    while (todo != 0ull) {
      const int bit = (int)__builtin_ctzll(todo);
      todo &= todo - 1ull;
      const int i = 64 * h + bit;
      const bool mine = (m_mine[h] >> bit) & 1ull;  // did any pixel of THIS wave blend it in the forward?
      // Until this wave meets the first entry one of its pixels blended, every lane still has last_alpha = 0: the
      // unguarded recursion A <- 0 * D_last + 1 * A leaves A where it is, and the D_last it records is overwritten by
      // that first entry before anything reads it — such visits (entries only OTHER waves of the tile blended, behind
      // this wave's last contributors) are skipped outright, bit for bit the same result.
      if (REF && !mine && !seen_mine) continue;
      seen_mine = true;
      const float* fr = &s_feat[i * FR];
      float D_cur = 0.f;
      if constexpr (F > 0) {
#pragma clang fp contract(fast)
        const bv2* fr2 = reinterpret_cast<const bv2*>(fr + 4);
        bv2 acc2 = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < F2; ++k) acc2 += fr2[k] * dLf2[k];
        D_cur = acc2.x + acc2.y;
      }
      if (REF && !mine) {
        // every pixel of this slot skips, but the tile does not: only the unguarded language
        // recursion advances (CR/backward.cu:1127-1139)
        if constexpr (F > 0) {
#pragma clang fp contract(fast)
          A_f = last_alpha * D_last + (1.f - last_alpha) * A_f;
          D_last = D_cur;
        }
        continue;
      }
      const int k = kstart - i;  // == `contributor` after its decrement (CR/backward.cu:999,1073)
      const float2 xy = s_xy[i];
      const float4 co = s_co[i];

      // -- decision path: bit-for-bit the forward's arithmetic (no contraction)
      bool skip = !inside;
      skip |= (k >= last_contributor);
      const float dx = xy.x - pixfx, dy = xy.y - pixfy;
      const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
      skip |= power > 0.0f;
      const float G = pinned_expf(power);
      const float alpha = fminf_ref(0.99f, co.w * G);
      skip |= alpha < 1.0f / 255.0f;

      // -- value path: FMA contraction allowed (rounding differs from the oracle by ~1 ulp per
      //    operation; no decision depends on these values).  Everything a skipping pixel must not
      //    contribute hangs off three factors that are zero for it.
      // The state update is branch-free: every lane evaluates it, a skipping lane keeps its old state through a
      // select.  (An `if (!skip)` block cost ~28 register copies per visit around the exec-masked region, in a kernel
      // that is bound by VALU issue.)
      float f_dcd, f_dLa;  // alpha * T (dchannel_dcolor) and dL_dalpha; zero for a skipping pixel
      float sum[NVP];
      {
#pragma clang fp contract(fast)
        if constexpr (REF && F > 0) {  // unguarded (CR/backward.cu:1132-1133)
          A_f = last_alpha * D_last + (1.f - last_alpha) * A_f;
          D_last = D_cur;
        }
        // Everything a skipping pixel must not do hangs off two masked factors: alpha_eff = 0 leaves Z unchanged and
        // zeroes alpha * T, inv_eff = 1 leaves T unchanged.  Colour and depth run on packed fp32 pairs — an
        // instruction holds the SIMD's issue for one quad-cycle whether it is packed or not.
        const float one_m_alpha = 1.f - alpha;  // in [0.01, 1] for every pixel that does not skip
        float inv = __builtin_amdgcn_rcpf(one_m_alpha);
        inv = __builtin_fmaf(__builtin_fmaf(-one_m_alpha, inv, 1.0f), inv, inv);
        const float alpha_eff = skip ? 0.f : alpha;
        const float inv_eff = skip ? 1.f : inv;
        T = T * inv_eff;
        const float4 f4 = *reinterpret_cast<const float4*>(fr);  // r g b depth of the splat (wave-uniform LDS read)
        const bv2 d01 = bv2{f4.x, f4.y} - Z01, d23 = bv2{f4.z, f4.w} - Z23;
        const bv2 dd = d01 * dL01 + d23 * dL23;
        float dL_dalpha = dd.x + dd.y;
        const bv2 a2 = {alpha_eff, alpha_eff};
        Z01 = Z01 + a2 * d01;
        Z23 = Z23 + a2 * d23;
        if constexpr (F > 0) {
          if constexpr (!REF) {  // guarded like colour
            const float A_new = last_alpha * D_last + (1.f - last_alpha) * A_f;
            dL_dalpha += D_cur - A_new;
            A_f = skip ? A_f : A_new;
            D_last = skip ? D_last : D_cur;
          } else {
            dL_dalpha += D_cur - A_f;
          }
        }
        dL_dalpha *= T;
        dL_dalpha += bg_term * inv;  // (-T_final / (1 - alpha)) <bg, dL_dpixel>, CR/backward.cu:1119-1125; one fma, 0 without a background
        last_alpha = skip ? last_alpha : alpha;
        f_dcd = alpha_eff * T;
        f_dLa = skip ? 0.f : dL_dalpha;
        // reference mode: only the ranks that survive its 225-lane tree contribute to these ten sums
        const float Gm = (skip || !surv) ? 0.f : G;  // (G of a skipping pixel may be huge: keep it out of products)
        const float s_dLa = surv ? f_dLa : 0.f;
        const float s_dcd = surv ? f_dcd : 0.f;
        const float dL_dG = co.w * s_dLa;
        const bv2 dxy = {dx, dy};
        const bv2 gd = dxy * bv2{Gm, Gm};                                        // {gdx, gdy}
        const bv2 dGd = -(gd * bv2{co.x, co.z}) - bv2{gd.y, gd.x} * bv2{co.y, co.y};  // {dG_ddelx, dG_ddely}
        const bv2 s01 = (dGd * bv2{dL_dG, dL_dG}) * bv2{ddelx_dx, ddely_dy};
        const float h = -0.5f * dL_dG;
        const bv2 s23 = dxy * bv2{gd.x * h, gd.x * h};                           // -0.5 gdx {dx, dy} dL_dG
        const bv2 s67 = dL01 * bv2{s_dcd, s_dcd}, s89 = dL23 * bv2{s_dcd, s_dcd};
        sum[0] = s01.x;
        sum[1] = s01.y;
        sum[2] = s23.x;
        sum[3] = s23.y;
        sum[4] = (gd.y * h) * dy;
        sum[5] = Gm * s_dLa;
        sum[6] = s67.x;
        sum[7] = s67.y;
        sum[8] = s89.x;
        sum[9] = s89.y;
        if constexpr (!REF && F > 0) {
#pragma unroll
          for (int k2 = 0; k2 < F2; ++k2) {
            const bv2 pr = dLf2[k2] * s_dcd;
            sum[10 + 2 * k2] = pr.x;
            if (10 + 2 * k2 + 1 < NVP) sum[10 + 2 * k2 + 1] = pr.y;
          }
        }
#pragma unroll
        for (int v = NV; v < NVP; ++v) sum[v] = 0.f;
      }

      // Wave reduction (olsr_device.h): four values per permlane-swap tree, the 1-2 left over in a
      // two-value tree.  Total j ends up in the lanes whose role is j (role_of below); that lane stores it.
      float rowval = 0.f;
      if constexpr (MRED) {
        bwd_f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 10; ++m) d4 = __builtin_amdgcn_mfma_f32_16x16x4f32(sum[m], sel[m], d4, 0, 0, 0);
        const float part = (d4[0] + d4[1]) + (d4[2] + d4[3]);
        const bwd_f32x4 t4 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, part, bwd_f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        rowval = t4[0];
      } else if constexpr (MERGED && LDSR) {
        float* wl = &s_red[w * 256];
        const float t0 = wave_fold4_lds(sum[0], sum[1], sum[2], sum[3], wl);
        const float t1 = wave_fold4_lds(sum[4], sum[5], sum[6], sum[7], wl);
        const float t2 = wave_fold2_lds(sum[8], sum[9], wl);
        rowval = row_sums3(t0, t1, t2, in_hi8, in_0to3);
      } else if constexpr (MERGED) {
        const float t0 = wave_fold4(sum[0], sum[1], sum[2], sum[3]);
        const float t1 = wave_fold4(sum[4], sum[5], sum[6], sum[7]);
        const float t2 = wave_fold2(sum[8], sum[9]);
        rowval = row_sums3(t0, t1, t2, in_hi8, in_0to3);
      } else if constexpr (LDSR) {
        float* wl = &s_red[w * 256];
#pragma unroll
        for (int g = 0; g < NG4; ++g) {
          const float red = wave_reduce4_lds(sum[4 * g], sum[4 * g + 1], sum[4 * g + 2], sum[4 * g + 3], wl);
          rowval = (lg == g) ? red : rowval;
        }
        if constexpr (REM > 0) {
          const float red = wave_reduce2_lds(sum[4 * NG4], REM > 1 ? sum[4 * NG4 + 1] : 0.f, wl);
          rowval = (lg == NG4) ? red : rowval;
        }
      } else {
#pragma unroll
        for (int j = 0; j < NTRI; ++j) {
          const int g = 3 * j;
          const float t0 = wave_fold4(sum[4 * g], sum[4 * g + 1], sum[4 * g + 2], sum[4 * g + 3]);
          const float t1 = wave_fold4(sum[4 * g + 4], sum[4 * g + 5], sum[4 * g + 6], sum[4 * g + 7]);
          const float t2 = wave_fold4(sum[4 * g + 8], sum[4 * g + 9], sum[4 * g + 10], sum[4 * g + 11]);
          const float red = row_sums3(t0, t1, t2, in_hi8, in_0to3);
          rowval = ((lg & 3) == j) ? red : rowval;  // (own-group lanes below overwrite theirs)
        }
#pragma unroll
        for (int g = 3 * NTRI; g < NG4; ++g) {
          const float red = wave_reduce4(sum[4 * g], sum[4 * g + 1], sum[4 * g + 2], sum[4 * g + 3]);
          rowval = (lg == g) ? red : rowval;
        }
        if constexpr (REM > 0) {
          const float red = wave_reduce2(sum[4 * NG4], REM > 1 ? sum[4 * NG4 + 1] : 0.f);
          rowval = (lg == NG4) ? red : rowval;
        }
      }
      if constexpr (REF && F > 0) {
        // language gradients come from tile rank 0 only (lane 0 of wave 0): one broadcast, times the
        // lane's own dL_dF[role - 10] of that pixel (zero in every other wave)
        const float dcd0 = lane_read(f_dcd, 0);
        rowval = lang_lane ? dcd0 * dLf0_lane : rowval;
      }
      // compact row index: rows of an instance are consecutive, one per set slot bit
      u32 before;  // rows of earlier waves
      if constexpr (PACKED) {
        before = (u32)w & (u32)((m_w0[h] >> bit) & 1ull);
      } else {
        const u32 fl = (u32)__builtin_amdgcn_readfirstlane((int)s_flag[i]);
        before = (u32)__popc(fl & ((1u << w) - 1u));
      }
      float* rowp = rows + ((size_t)(u32)__builtin_amdgcn_readfirstlane((int)s_row[i]) + before) * ROW;  // scalar
      if (role >= 0) rowp[role] = rowval;
    }
    }
  }
}

template <int TILE, int F, int MODE>
static void launch_bwd_t(const olsr_scene& s, const FrameDims& d, const GeometryState& g, const BinningState& b,
                         const ImageState& im, const float* dc, const float* dl, const float* dd, float* rows,
                         hipStream_t st) {
  const float* colors = s.colors_precomp ? s.colors_precomp : g.rgb;
  constexpr bool PACKED = (MODE == OLSR_BWD_REFERENCE && TILE == 15);
  render_bwd_kernel<TILE, F, MODE, PACKED><<<d.ntiles, PACKED ? 128 : 256, 0, st>>>(
      im.ranges, b.inst_gid, b.src, b.flags, b.rowbase, g.counters, im.tile_order, d.W, d.H, d.gx, d.ntiles,
      s.background,
      g.means2D, g.conic_opacity, colors, s.language_precomp, g.depths, im.final_T, im.n_contrib, dc, dl, dd, rows,
      rows_stamp_of(s.backward_row_capacity));
}

// F_rows: the language channels the rows carry — s.F, or 0 when the caller handed no language cotangent (the
// tracking loss, utils/slam_utils.py:92-121): the RGB instantiation then runs on the language forward's state
template <int TILE, int MODE>
static void launch_bwd_f(const olsr_scene& s, int F_rows, const FrameDims& d, const GeometryState& g, const BinningState& b,
                         const ImageState& im, const float* dc, const float* dl, const float* dd, float* rows,
                         hipStream_t st) {
  switch (F_rows) {
    case 0: launch_bwd_t<TILE, 0, MODE>(s, d, g, b, im, dc, dl, dd, rows, st); break;
    case 3: launch_bwd_t<TILE, 3, MODE>(s, d, g, b, im, dc, dl, dd, rows, st); break;
    case 15: launch_bwd_t<TILE, 15, MODE>(s, d, g, b, im, dc, dl, dd, rows, st); break;
    case 16: launch_bwd_t<TILE, 16, MODE>(s, d, g, b, im, dc, dl, dd, rows, st); break;
    case 32: launch_bwd_t<TILE, 32, MODE>(s, d, g, b, im, dc, dl, dd, rows, st); break;
    default: break;
  }
}

#ifndef OLSR_BWD_TU_MODE
#error "compile with -DOLSR_BWD_TU_MODE=0 (reference) or 1 (exact)"
#endif

#if OLSR_BWD_TU_MODE == 0
void launch_render_backward_reference(const olsr_scene& s, int F_rows, const FrameDims& d, const GeometryState& g,
                                      const BinningState& b, const ImageState& im, const float* dc, const float* dl,
                                      const float* dd, float* rows, hipStream_t st) {
  if (d.tile == 15)
    launch_bwd_f<15, OLSR_BWD_REFERENCE>(s, F_rows, d, g, b, im, dc, dl, dd, rows, st);
  else
    launch_bwd_f<16, OLSR_BWD_REFERENCE>(s, F_rows, d, g, b, im, dc, dl, dd, rows, st);
}
#else
void launch_render_backward_exact(const olsr_scene& s, int F_rows, const FrameDims& d, const GeometryState& g,
                                  const BinningState& b, const ImageState& im, const float* dc, const float* dl,
                                  const float* dd, float* rows, hipStream_t st) {
  if (d.tile == 15)
    launch_bwd_f<15, OLSR_BWD_EXACT>(s, F_rows, d, g, b, im, dc, dl, dd, rows, st);
  else
    launch_bwd_f<16, OLSR_BWD_EXACT>(s, F_rows, d, g, b, im, dc, dl, dd, rows, st);
}
#endif

}  // namespace olsr
