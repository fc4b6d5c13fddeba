// k_render_fwd.hip — forward front-to-back alpha compositing of RGB + depth + F language channels.
//
// Replaces renderCUDA / language_renderCUDA (CR/forward.cu:515-644, 377-513).
//
// MI355X mapping: one workgroup per logical TILE x TILE tile, 4 wave64s, one pixel per lane
// (thread rank = ty*TILE + tx as in the reference; wave w owns one QUADRANT of the tile — slot_rank, olsr_device.h:
// 64 / 56 / 56 / 49 pixels of a 15x15 tile; until late in round 4 the ranks 64w..64w+63, a strip).  What differs from the reference:
//   * splat data is staged through LDS in batches of 128 with ONE coalesced gather per thread
//     (half the threads fetch geometry, half fetch the colour/depth/language row), and the
//     per-splat colour + language features are read from LDS as wave-uniform broadcasts — the
//     reference re-gathers them from global memory for every contributing (pixel, splat) pair;
//   * inside a batch the four waves run free: no barrier per splat, a wave leaves the batch as
//     soon as its 64 pixels are saturated (64-bit ballots), the workgroup stops staging when
//     all four are;
//   * the kernel is VALU-issue bound, so its inner loop is written for instruction count: the lane
//     predicates are 64-bit wave masks in scalar registers (v_cmp + s_and, "any lane" = s_cmp, the
//     accumulation under inverse_ballot), two list entries are evaluated per iteration with their
//     geometry / power / exp on packed fp32, the accumulation is v_pk_mul_f32 + v_pk_fma_f32;
//   * n_touched is counted with per-wave popcounts (each wave owns 16 bits of the splat's LDS
//     record: no LDS atomics) and ONE global integer atomic per (splat, batch);
//   * tiles are launched heaviest-first inside each XCD's chunk when the caller hands back the
//     order measured on its previous frame (tile_order_inout of olsr_forward_async);
//   * the kernel records, per (tile, splat) instance, WHICH 64-pixel slots blended it
//     (flags[] bits 0-3, indexed by emission position; bits 4-5 say the same for the two packed waves of
//     the reference-mode backward, see ref15_survives in olsr_device.h).  The backward composite visits only those
//     (instance, slot) pairs, and "flags != 0" is exactly the tile-wide
//     skip_counter != BLOCK_SIZE predicate of CR/backward.cu:1087-1093.
// Per-pixel arithmetic keeps the reference's operation order (see olsr_device.h), so the
// images are bit-identical to the CPU oracle.
#include "olsr_device.h"
#include "olsr_kernels.h"
#include "olsr_loss_device.h"

namespace olsr {

constexpr int FWD_BATCH = 128;
#ifndef OLSR_FWD_WAVES
#define OLSR_FWD_WAVES 7  // waves per SIMD the default accumulation is compiled for at F <= 16 (8: spills, measured slower)
#endif
#ifndef OLSR_FWD_LOSS1_WAVES
#define OLSR_FWD_LOSS1_WAVES 6  // ... and the mapping-loss epilogue at F = 15 / 16: 7 left 8 - 12 bytes of scratch per lane in the
                                // epilogue; 6 has none and measures the same (12-view mapping iteration, room 3.325 / 3.331 ms,
                                // volume 6.41 / 6.44 ms, scripts/probe/mapping_time.py)
#endif
#ifndef OLSR_FWD_ACC2_WAVES
#define OLSR_FWD_ACC2_WAVES 7  // waves per SIMD the weight-accumulation variant is compiled for
#endif

#ifdef OLSR_FWD_STATS
// experiment build only (scripts/build_variant.sh ... -DOLSR_FWD_STATS): how many (entry, wave) pairs the loop looks at,
// how many pass the wave-level reach test, how many blend, and how many lanes blend — read with olsr_debug_fwd_stats
__device__ unsigned long long g_fwd_stats[8];
#define FWD_STAT(i, v) st_##i += (unsigned)(v)
#else
#define FWD_STAT(i, v)
#endif

// MFMA (OLSR_FLAG_FWD_ACCUM_MFMA): the accumulation acc[64 px x (4 + F)] += w[64 px x K] feat[K x (4 + F)], w = alpha T, runs on
// the matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32, a k-ordered fma chain) in groups of K = 4 blending entries per
// wave, beside the VALU that evaluates alpha and the transmittance; every decision stays on the VALU, bit for bit.
// The product is then rounded as fma(alpha T, f, C) instead of the reference's fma(f alpha, T, C): images agree with the
// oracle to ~1e-7 relative instead of bit for bit (final_T, n_contrib, radii, n_touched, flags stay bit-identical).
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ACC: 0 = the reference's rounding fma(f alpha, T, C) on the vector ALU (bit-identical to the oracle, default);
//      1 = matrix cores (OLSR_FLAG_FWD_ACCUM_MFMA); 2 = w = alpha T once per pixel, then ONE fma(w, f, C) per channel on
//      the vector ALU (OLSR_FLAG_FWD_ACCUM_WEIGHT: half the lane operations of the accumulation, the MFMA variant's rounding)
// LOSS: 0 = images only; 1 / 2 = the mapping / tracking loss evaluated in the epilogue (olsr_forward_async_loss): the pixel's
//      colour, depth, language features and transmittance are still in registers, so the cotangents the backward consumes and
//      the tile's partial loss sums are produced here instead of by a kernel that re-reads the images (csrc/olsr_loss_device.h).
// CUT:  the per-tile depth cut-off bookkeeping (include/olsr.h) — its own instantiation: carried as a run-time test it cost the
//       plain kernel 4 % (0.1588 -> 0.1647 ms at config 3, two more live scalars across the entry loop).
template <int TILE, int F, int ACC, int LOSS, bool CUT>
__global__ __launch_bounds__(256, (F <= 16 ? (ACC == 1 ? 5 : (ACC == 2 ? OLSR_FWD_ACC2_WAVES : ((LOSS == 1 && F >= 15) ? OLSR_FWD_LOSS1_WAVES : OLSR_FWD_WAVES))) : (ACC == 1 ? 4 : 5))) void render_fwd_kernel(
    const u32* ranges, u32* ranges_rw, const u32* __restrict__ inst_gid, const u32* __restrict__ src, int W, int H,
    int gx, int ntiles, const float* __restrict__ means2D, const float* __restrict__ conic_opacity,
    const float* __restrict__ depths, const float* __restrict__ colors, const float* __restrict__ lang,
    const float* __restrict__ bg, float* __restrict__ final_T, u32* __restrict__ n_contrib,
    float* __restrict__ out_color, float* __restrict__ out_lang, float* __restrict__ out_depth,
    float* __restrict__ out_opacity, int32_t* __restrict__ n_touched, uint8_t* __restrict__ flags,
    u32* __restrict__ tile_work, const u32* __restrict__ order_hint, const int32_t* __restrict__ counters,
    const u32* __restrict__ hint_slot, float* __restrict__ depth_cut, int32_t* __restrict__ cut_miss,
    uint8_t* __restrict__ blended, const FusedLossArgs fl) {
#ifdef OLSR_COMPOSITE_VGPR_FLOOR
  asm volatile("; vgpr floor" ::: OLSR_COMPOSITE_VGPR_FLOOR);  // (experiment: fewer resident waves, room for other frames' kernels)
#endif
  // a radix pass of this frame lost a predecessor's counts (olsr_state.h, counters[8]): the lists are garbage and must not be
  // used as indices — render nothing; the tile-order kernel behind this one reports OLSR_STATUS_SYNC_ERROR
  if (counters[8] != 0) return;
  constexpr int BS = TILE * TILE;
  constexpr int FR = feat_row(F);
  constexpr int NA = 4 + F;  // r g b depth lang[F]
  constexpr int B = FWD_BATCH;
  constexpr bool MFMA = (ACC == 1);
  constexpr int NC = (NA + 15) / 16;  // MFMA: channel blocks of 16

  // geometry of two consecutive list entries side by side, so that 16-byte LDS reads land as register pairs for
  // packed fp32 math: {x0 x1 y0 y1} {thr0 thr1 a0 a1} {b0 b1 c0 c1} {op0 op1 - -}
  __shared__ float4 s_pair[(B / 2) * 4];
  __shared__ __attribute__((aligned(16))) float s_feat[B * FR + 16];  // (+16: the MFMA B operand reads whole 16-channel blocks)
  __shared__ u32 s_id[B];
  __shared__ u32 s_src[B];
  // per staged splat, 16 bits per wave: bit 15 = the wave's slot blended it, bits 0-6 = how many of its pixels
  // counted it as touched (<= 64), bits 8-9 = which packed survivor wave of the reference-mode backward a
  // blending pixel belongs to (15x15 tiles); each wave writes only its own 16 bits
  __shared__ uint2 s_hit[B];
  __shared__ u32 s_work, s_work2;
  __shared__ int s_stop, s_undone;  // per-tile depth cut-offs: the deepest list position a wave stopped at; waves not saturated

  // workgroup b runs on XCD b % 8; with a hint it takes the (b / 8)-th heaviest tile of that XCD's chunk as
  // measured on the caller's previous frame, else the (b / 8)-th tile of the chunk
  int tile_id = xcd_remap((int)blockIdx.x, ntiles);
  // (hint_slot: the synchronising entry keeps several orders per stream, one per view it has seen; word 0 names this frame's)
  if (order_hint != nullptr)
  {
    // (a hint is caller memory: an entry that is no tile id — an uninitialised or stale buffer — must not become an address;
    //  such a frame renders some tiles twice and others not at all, which the caller's bug earns, but it stays in bounds)
    const u32 hinted = order_hint[(hint_slot != nullptr ? (size_t)hint_slot[0] * (size_t)ntiles : (size_t)0) + (size_t)tile_id];
    if (hinted < (u32)ntiles) tile_id = (int)hinted;
  }
  const int tid = threadIdx.x;
  const int w = tid >> 6;
  const int bx = tile_id % gx, by = tile_id / gx;
  u32 r0 = ranges[2 * tile_id], r1 = ranges[2 * tile_id + 1];
  if (r1 <= r0) {  // a tile without instances still holds the initial {UINT_MAX, 0}: leave {0, 0} behind like the
    r0 = 0;        // reference's cudaMemset + identifyTileRanges (CR/rasterizer_impl.cu:485-493)
    r1 = 0;
    if (threadIdx.x == 0) {
      ranges_rw[2 * tile_id] = 0;
      ranges_rw[2 * tile_id + 1] = 0;
    }
  }
  const int n = (int)(r1 - r0);

  const int rank = slot_rank<TILE>(tid);  // wave w = one quadrant of the tile (olsr_device.h)
  const int px = bx * TILE + rank % TILE, py = by * TILE + rank / TILE;
  const bool inside = (rank < BS) && (px < W) && (py < H);
  const float pixfx = (float)px, pixfy = (float)py;
  bool done = !inside;
  // backward wave (0 / 1) of this pixel among the survivors of the reference's tree, 2 = not a survivor
  int cls = 2;
  if constexpr (TILE == 15) {
    if (rank < BS && ref15_survives(rank)) cls = ref15_packed_of_rank(rank) >> 6;
  }
  const u64 cls_m0 = ballot(cls == 0), cls_m1 = ballot(cls == 1);  // wave-uniform lane masks (scalar registers)
  u64 done_m = ballot(done);  // lanes that are outside the image or saturated
  const bool lane0 = (tid & 63) == 0;
  if (tid == 0) {
    s_work = 0;
    s_work2 = 0;
    s_stop = -1;
    s_undone = 0;
  }
  int my_stop = -1;   // (wave-uniform) list position of the entry at which this wave's last pixel saturated
  int last_base = 0;  // (uniform) first list position of the batch staged last
  // (the tile's live-pair counts are summed per flush with wave ballots straight into LDS: a per-thread counter kept across
  //  the loop cost a register the accumulation variants do not have)
#ifdef OLSR_FWD_STATS
  unsigned st_0 = 0, st_1 = 0, st_2 = 0, st_3 = 0, st_4 = 0;
#endif
  float T = 1.0f;
  u32 last_contributor = 0;
  // r g b depth lang[F] in pairs: the accumulation runs on packed fp32 (v_pk_mul_f32 / v_pk_fma_f32)
  constexpr int NA2 = (NA + 1) / 2;
  v2f acc2[MFMA ? 1 : NA2];
#pragma unroll
  for (int k = 0; k < (MFMA ? 1 : NA2); ++k) acc2[k] = v2f{0.0f, 0.0f};
  // MFMA: accm[b][c] = the 16 x 16 block (pixels 16 b .. 16 b + 15 of this wave) x (channels 16 c .. 16 c + 15) in the D
  // layout (lane l, register r: pixel 4 (l >> 4) + r, channel l & 15); up to four blending entries wait in w0..w3
  // (w = alpha T per pixel, 0 where the pixel does not blend) with their LDS slots in `myslot` (row k of the wave: entry k)
  f32x4 accm[MFMA ? 4 : 1][MFMA ? NC : 1];
#pragma unroll
  for (int b_ = 0; b_ < (MFMA ? 4 : 1); ++b_)
#pragma unroll
    for (int c_ = 0; c_ < (MFMA ? NC : 1); ++c_) accm[b_][c_] = f32x4{0.f, 0.f, 0.f, 0.f};
  float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
  int gfill = 0;   // wave-uniform: entries waiting
  u32 myslot = 0;
  const int bcol = tid & 15;
  auto mfma_flush = [&]() {
    if constexpr (MFMA) {
      // 4 x 4 transpose of (row of 16 lanes, register): A_b[row k] = w_k[row b] — two half swaps, two row swaps
      const auto s02 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, w0), __builtin_bit_cast(unsigned, w2), false, false);
      const auto s13 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, w1), __builtin_bit_cast(unsigned, w3), false, false);
      const unsigned p0 = s02[0], p2 = s02[1], p1 = s13[0], p3 = s13[1];
      const auto t01 = __builtin_amdgcn_permlane16_swap(p0, p1, false, false);
      const auto t23 = __builtin_amdgcn_permlane16_swap(p2, p3, false, false);
      const unsigned q0 = t01[0], q1 = t01[1], q2 = t23[0], q3 = t23[1];
      const float a_[4] = {__builtin_bit_cast(float, q0), __builtin_bit_cast(float, q1), __builtin_bit_cast(float, q2),
                           __builtin_bit_cast(float, q3)};
#pragma unroll
      for (int c_ = 0; c_ < NC; ++c_) {
        const float bv = s_feat[myslot * FR + 16 * c_ + bcol];  // B[k][j]: channel 16 c + j of the entry in row k
#pragma unroll
        for (int b_ = 0; b_ < 4; ++b_) accm[b_][c_] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[b_], bv, accm[b_][c_], 0, 0, 0);
      }
      w0 = 0.f; w1 = 0.f; w2 = 0.f; w3 = 0.f;
      gfill = 0;
    }
  };

  for (int base = 0; base < n; base += B) {
    // also the barrier that separates the previous batch's flush from this batch's staging
    if (__syncthreads_and(done_m == ~0ull)) break;
    const int cnt = min(B, n - base);
    if constexpr (CUT) last_base = base;
    {
      const int e = tid & (B - 1);
      if (e == cnt && (cnt & 1) && tid < B) {  // odd tail: the partner slot of the last entry can never be reached
        float* q = reinterpret_cast<float*>(&s_pair[(e >> 1) * 4]) + 1;
        q[0] = 0.f; q[2] = 0.f; q[4] = __builtin_inff(); q[6] = 0.f; q[8] = 0.f; q[10] = 0.f; q[12] = 0.f;
      }
      if (e < cnt) {
        const u32 sp = r0 + (u32)base + (u32)e;
        const u32 u = src[sp];        // emission index of the instance
        const u32 gid = inst_gid[u];  // its Gaussian
        if (tid < B) {
          s_id[e] = gid;
          s_src[e] = u;
          s_hit[e] = make_uint2(0u, 0u);
          const float2 m = reinterpret_cast<const float2*>(means2D)[gid];
          const float4 c = reinterpret_cast<const float4*>(conic_opacity)[gid];
          // alpha = o * exp(power) can only reach 1/255 if power >= -ln(255 o).  Keep a margin far
          // above the error of __logf and of the pinned exp, so that the wave-level early-out
          // below never drops a pair the exact test would keep.
          const float L = -__logf(255.0f * c.w);
          float* q = reinterpret_cast<float*>(&s_pair[(e >> 1) * 4]) + (e & 1);
          q[0] = m.x;
          q[2] = m.y;
          q[4] = L - (1e-3f + 1e-4f * fabsf(L));
          q[6] = c.x;
          q[8] = c.y;
          q[10] = c.z;
          q[12] = c.w;
        } else {
          float* fr = &s_feat[e * FR];
          fr[0] = colors[3 * (size_t)gid + 0];
          fr[1] = colors[3 * (size_t)gid + 1];
          fr[2] = colors[3 * (size_t)gid + 2];
          fr[3] = depths[gid];
#pragma unroll
          for (int ch = 0; ch < F; ++ch) fr[4 + ch] = lang[(size_t)gid * F + ch];
#pragma unroll
          for (int ch = 4 + F; ch < FR; ++ch) fr[ch] = 0.f;  // read by the packed accumulation
        }
      }
    }
    __syncthreads();

    if (done_m != ~0ull) {
      // The lane predicates of CR/forward.cu:449-476 live in scalar registers as 64-bit wave masks: every test
      // is one v_cmp whose result is combined with s_and / s_andn2, "any lane" is an s_cmp, and only the
      // accumulation runs under a lane mask (inverse_ballot -> exec).  Two list entries are evaluated per
      // iteration: their geometry, power and exp run on packed fp32 (each component is the IEEE operation of the
      // scalar code, so the bits are unchanged); the compositing itself stays strictly sequential.
      const v2f pixx2 = {pixfx, pixfx}, pixy2 = {pixfy, pixfy};
      for (int j = 0; j < cnt; j += 2) {
        const float4 q0 = s_pair[(j >> 1) * 4 + 0], q1 = s_pair[(j >> 1) * 4 + 1], q2 = s_pair[(j >> 1) * 4 + 2];
        const v2f dx = v2f{q0.x, q0.y} - pixx2, dy = v2f{q0.z, q0.w} - pixy2;
        const v2f ca = {q1.z, q1.w}, cb = {q2.x, q2.y}, cc = {q2.z, q2.w};
        const v2f power = v2f{-0.5f, -0.5f} * (ca * dx * dx + cc * dy * dy) - cb * dx * dy;
        u64 live = ~done_m;
        // wave-level early-out: no live pixel of this slot can reach the alpha floor of either entry
        const u64 reach0 = ballot(!(power.x < q1.x)) & live, reach1 = ballot(!(power.y < q1.y)) & live;
        FWD_STAT(0, (j + 1 < cnt) ? 2 : 1);
        FWD_STAT(1, (reach0 != 0ull) + (reach1 != 0ull));
        if ((reach0 | reach1) == 0ull) continue;
        const float2 op = *reinterpret_cast<const float2*>(&s_pair[(j >> 1) * 4 + 3]);
        const v2f G = pinned_expf2(power);
        const float alpha0 = fminf_ref(0.99f, op.x * G.x), alpha1 = fminf_ref(0.99f, op.y * G.y);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const u64 reach = h ? reach1 : reach0;
          if (reach == 0ull) continue;
          const float pw = h ? power.y : power.x, alpha = h ? alpha1 : alpha0;
          const int jj = j + h;
          live = ~done_m;
          const float test_T = T * (1 - alpha);
          const u64 ok_m = ballot(!(pw > 0.0f)) & ballot(!(alpha < 1.0f / 255.0f)) & live;
          const u64 term_m = ok_m & ballot(test_T < 0.0001f);
          const u64 contrib_m = ok_m & ~term_m;
          done_m |= term_m;
          FWD_STAT(4, 1);
          if (MFMA && contrib_m != 0ull) {
            const bool cl = __builtin_amdgcn_inverse_ballot_w64(contrib_m);
            const float wv = cl ? alpha * T : 0.0f;
            T = cl ? test_T : T;
            last_contributor = cl ? (u32)(base + jj + 1) : last_contributor;
            if (__builtin_amdgcn_inverse_ballot_w64(0xFFFFull << (16 * gfill))) myslot = (u32)jj;
            if (gfill == 0) w0 = wv;
            else if (gfill == 1) w1 = wv;
            else if (gfill == 2) w2 = wv;
            else w3 = wv;
            if (++gfill == 4) mfma_flush();
          }
          if (contrib_m != 0ull) {
            FWD_STAT(2, 1);
            FWD_STAT(3, __popcll(contrib_m));
            if (!MFMA && __builtin_amdgcn_inverse_ballot_w64(contrib_m)) {
              // C += f * alpha * T as fma(f * alpha, T, C): what nvcc's default contraction makes of the
              // reference's expression (CR/forward.cu:479-484), and what the oracle restates
              const v2f* fr2 = reinterpret_cast<const v2f*>(&s_feat[jj * FR]);
              if constexpr (ACC == 2) {
                const float wv = alpha * T;
                const v2f w2 = {wv, wv};
#pragma unroll
                for (int k = 0; k < NA2; ++k) acc2[k] = __builtin_elementwise_fma(fr2[k], w2, acc2[k]);
              } else {
                // (the same arithmetic on scalar fp32 — mul, fma per channel instead of the packed pairs — measured equal in
                //  round 4, 0.1648 against 0.1647 ms: op_sel broadcasts alpha and T, there are no operand moves to save)
                const v2f a2 = {alpha, alpha}, T2 = {T, T};
#pragma unroll
                for (int k = 0; k < NA2; ++k) acc2[k] = __builtin_elementwise_fma(fr2[k] * a2, T2, acc2[k]);
              }
              T = test_T;
              last_contributor = (u32)(base + jj + 1);
            }
            const u32 tc = (u32)__popcll(contrib_m & ballot(test_T > 0.5f));
            u32 rec = 0x8000u | tc;
            if constexpr (TILE == 15) rec |= ((contrib_m & cls_m0) ? 0x100u : 0u) | ((contrib_m & cls_m1) ? 0x200u : 0u);
            if (lane0) reinterpret_cast<uint16_t*>(s_hit)[4 * jj + w] = (uint16_t)rec;
          }
#ifndef OLSR_FWD_INNER_BREAK
          // (no exit between the two entries of a pair: once every pixel of the wave is saturated `live` is empty, the second
          //  entry blends nothing, records nothing and changes no T — and the loop leaves below.  The break that stood here
          //  cost every blending entry ~12 scalar instructions of the structuriser's exit encoding.  The cut-off variant needs
          //  the entry at which the wave saturated.)
          if constexpr (CUT) {
            if (done_m == ~0ull) {
              my_stop = base + jj;
              break;
            }
          }
#else
          if (done_m == ~0ull) {
            if constexpr (CUT) my_stop = base + jj;
            break;
          }
#endif
        }
        if (done_m == ~0ull) break;
      }
      // the entries still waiting read their features from THIS batch's LDS rows: flush before it is restaged
      if (MFMA && gfill != 0) mfma_flush();
    }
    __syncthreads();
    if (tid < B) {  // (waves 0 and 1, whole: the ballots below need every lane)
      u32 fl = 0, cl2 = 0;
      if (tid < cnt) {
        const uint2 hit = s_hit[tid];
        // bits 0-3: forward slots that blended it; bits 4-5: backward waves of the reference-mode survivors
        fl = ((hit.x >> 15) & 1u) | ((hit.x >> 30) & 2u) | ((hit.y >> 13) & 4u) | ((hit.y >> 28) & 8u);
        cl2 = ((hit.x >> 8) | (hit.x >> 24) | (hit.y >> 8) | (hit.y >> 24)) & 3u;
        if (fl) {
          flags[s_src[tid]] = (uint8_t)(fl | (cl2 << 4));
          blended[s_id[tid]] = 1;  // (every writer stores the same byte: the per-Gaussian backward looks rows up only for these)
        }
        const u32 tc = (hit.x & 0x7Fu) + ((hit.x >> 16) & 0x7Fu) + (hit.y & 0x7Fu) + ((hit.y >> 16) & 0x7Fu);
        if (tc) atomicAdd(&n_touched[s_id[tid]], (int)tc);
      }
      // live (instance, slot) pairs and live (instance, packed survivor wave) pairs of this flush: six ballots per wave
      const u32 c1 = (u32)(__popcll(ballot(fl & 1u)) + __popcll(ballot(fl & 2u)) + __popcll(ballot(fl & 4u)) +
                           __popcll(ballot(fl & 8u)));
      const u32 c2 = (u32)(__popcll(ballot(cl2 & 1u)) + __popcll(ballot(cl2 & 2u)));
      if (lane0 && c1) {
        atomicAdd(&s_work, c1);
        atomicAdd(&s_work2, c2);
      }
    }
  }

#ifdef OLSR_FWD_STATS
  if ((tid & 63) == 0) {
    atomicAdd(&g_fwd_stats[0], (unsigned long long)st_0);
    atomicAdd(&g_fwd_stats[1], (unsigned long long)st_1);
    atomicAdd(&g_fwd_stats[2], (unsigned long long)st_2);
    atomicAdd(&g_fwd_stats[3], (unsigned long long)st_3);
    atomicAdd(&g_fwd_stats[4], (unsigned long long)st_4);
    atomicAdd(&g_fwd_stats[5], (unsigned long long)n);
  }
#endif
  // Per-tile depth cut-offs (include/olsr.h).  The frame dropped every Gaussian that lies behind the cut-off of each tile
  // it reaches, so a tile's list is complete up to its cut-off depth and may have holes behind it.  The tile is exact if its
  // last pixel saturated at an entry not deeper than the cut-off; if it read on into the part with holes, or ran out of list
  // without saturating although a cut-off was in force, the frame is flagged.  It leaves its own cut-off for the next frame in
  // the second half of the array — 1.1 x the depth of the entry at which its last pixel saturated (that entry is in the batch
  // staged last, still in LDS) + 0.01, +infinity if a wave never saturated; the tile-order kernel dilates them (k_binning.hip).
  if constexpr (CUT) {
    if (lane0) {
      if (done_m != ~0ull) atomicAdd(&s_undone, 1);
      else if (my_stop >= 0) atomicMax(&s_stop, my_stop);
    }
    __syncthreads();
    if (tid == 0) {
      const float old_cut = depth_cut[tile_id];
      float new_cut = __builtin_inff();
      bool miss = s_undone != 0 && old_cut < __builtin_inff();
      if (s_undone == 0 && s_stop >= 0) {
        const float stop_depth = s_feat[(s_stop - last_base) * FR + 3];
        new_cut = stop_depth * 1.1f + 0.01f;
        miss = stop_depth > old_cut;
      }
      if (miss) atomicOr(cut_miss, 1);
      depth_cut[ntiles + tile_id] = new_cut;
    }
  }
  // backward work estimate of this tile: the number of (instance, slot) pairs it will visit
  __syncthreads();
  if (tid == 0) {
    // (the frame's totals are summed by the tile-order kernel: thousands of same-address atomics from here held the
    //  kernel's tail back by up to 13 us)
    tile_work[tile_id] = s_work;
    tile_work[ntiles + tile_id] = s_work2;
  }

  float acc[2 * NA2];
  if constexpr (MFMA) {
    // D layout -> one pixel per lane, through the (now idle) feature rows: two waves at a time, 64 pixels x FR floats each
    const int lane = tid & 63;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      float* o = s_feat + (w & 1) * 64 * FR;
      if ((w >> 1) == round) {
#pragma unroll
        for (int b_ = 0; b_ < 4; ++b_)
#pragma unroll
          for (int c_ = 0; c_ < NC; ++c_)
#pragma unroll
            for (int r_ = 0; r_ < 4; ++r_) {
              const int ch = 16 * c_ + bcol;
              if (ch < NA) o[(16 * b_ + 4 * (lane >> 4) + r_) * FR + ch] = accm[b_][c_][r_];
            }
      }
      __syncthreads();
      if ((w >> 1) == round) {
#pragma unroll
        for (int ch = 0; ch < NA; ++ch) acc[ch] = o[lane * FR + ch];
      }
      __syncthreads();
    }
  } else {
#pragma unroll
    for (int k = 0; k < NA2; ++k) {
      acc[2 * k] = acc2[k].x;
      acc[2 * k + 1] = acc2[k].y;
    }
  }
  const size_t HW = (size_t)H * W;
  const u32 p = (u32)W * (u32)py + (u32)px;
  if (inside) {
    final_T[p] = T;
    n_contrib[p] = last_contributor;
  }
  if (inside && (LOSS == 0 || fl.write_images)) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) out_color[ch * HW + p] = acc[ch] + T * bg[ch];
    out_depth[p] = acc[3];
    out_opacity[p] = 1 - T;
    if constexpr (F > 0) {
#pragma unroll
      for (int ch = 0; ch < F; ++ch) out_lang[ch * HW + p] = acc[4 + ch];
    }
  }
  if constexpr (LOSS != 0) {
    // ---- the loss of this pixel and its cotangents, from the values the stores above hold (same expressions, same bits) ----
    constexpr bool TRACK = (LOSS == 2);
    float sums[LOSS_SUMS] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (inside) {
      const float ea = fl.use_exposure ? expf(fl.exposure[0]) : 1.f;
      const float eb = fl.use_exposure ? fl.exposure[1] : 0.f;
      const float g0 = fl.gt_image[p], g1 = fl.gt_image[HW + p], g2 = fl.gt_image[2 * HW + p];
      float m = ((g0 + g1) + g2 > fl.thr) ? 1.f : 0.f;
      float op = 1.f;
      if constexpr (TRACK) {
        op = 1 - T;
        if (fl.grad_mask != nullptr) m *= fl.grad_mask[p];
      }
      const float wrgb = fl.alpha / (3.0f * (float)HW);
      fl.d_image[p] = loss_rgb_term<TRACK>(acc[0] + T * bg[0], g0, m, op, fl.use_exposure, ea, eb, wrgb, sums);
      fl.d_image[HW + p] = loss_rgb_term<TRACK>(acc[1] + T * bg[1], g1, m, op, fl.use_exposure, ea, eb, wrgb, sums);
      fl.d_image[2 * HW + p] = loss_rgb_term<TRACK>(acc[2] + T * bg[2], g2, m, op, fl.use_exposure, ea, eb, wrgb, sums);
      fl.d_depth[p] = loss_depth_term<TRACK>(acc[3], fl.gt_depth[p], op, (1.f - fl.alpha) / (float)HW, sums);
    }
    if constexpr (F > 0 && !TRACK) {
      if (fl.gt_lang != nullptr) {  // (uniform)
        // The language target is small (192 x 192 in the reference) and bilinearly enlarged: the pixels of this tile read a
        // window of a few texels per channel.  The workgroup stages that window in LDS once (the idle feature rows) instead of
        // every pixel gathering its 4 F corner values from global memory (60 scattered loads per pixel at F = 15 — the
        // epilogue's cost, measured at 36 us per view with four views in flight); targets whose window does not fit (a target
        // LARGER than the image) take the global path.  Same arithmetic, same bits, either way.
        const float sx = (float)fl.lw / (float)W, sy = (float)fl.lh / (float)H;
        int wx0, wx1, wy0, wy1, t_;
        float tf0, tf1;
        bilinear_index(bx * TILE, sx, fl.lw, wx0, t_, tf0, tf1);
        bilinear_index(min(bx * TILE + TILE - 1, W - 1), sx, fl.lw, t_, wx1, tf0, tf1);
        bilinear_index(by * TILE, sy, fl.lh, wy0, t_, tf0, tf1);
        bilinear_index(min(by * TILE + TILE - 1, H - 1), sy, fl.lh, t_, wy1, tf0, tf1);
        const int ww = wx1 - wx0 + 1, wh = wy1 - wy0 + 1, wn = ww * wh;
        const bool staged = wn * F <= B * FR;
        const size_t plane = (size_t)fl.lh * fl.lw;
        float* s_win = s_feat;  // (idle: the last batch ended with a barrier, and tile_work's barrier is behind us)
        if (staged) {
          // (quotients by reciprocal multiplication, exact for these ranges — i < B * FR, rem < wn: two integer divisions per
          //  element cost ~80 vector instructions)
          const float inv_wn = 1.0f / (float)wn, inv_ww = 1.0f / (float)ww;
          for (int i = tid; i < wn * F; i += 256) {
            const int ch = (int)(((float)i + 0.5f) * inv_wn), rem = i - ch * wn;
            const int yy = (int)(((float)rem + 0.5f) * inv_ww), xx = rem - yy * ww;
            s_win[i] = fl.gt_lang[ch * plane + (size_t)(wy0 + yy) * fl.lw + (wx0 + xx)];
          }
          __syncthreads();
        }
        if (inside) {
          int x0, x1, y0, y1;
          float lx0, lx1, ly0, ly1;
          bilinear_index(px, sx, fl.lw, x0, x1, lx0, lx1);
          bilinear_index(py, sy, fl.lh, y0, y1, ly0, ly1);
          const float wl = fl.lamda / ((float)F * (float)HW);
          if (staged) {
            const float* r0w = s_win + (y0 - wy0) * ww - wx0;
            const float* r1w = s_win + (y1 - wy0) * ww - wx0;
#pragma unroll
            for (int ch = 0; ch < F; ++ch)
              fl.d_lang[ch * HW + p] =
                  loss_lang_term(acc[4 + ch], r0w + ch * wn, r1w + ch * wn, x0, x1, lx0, lx1, ly0, ly1, wl, sums);
          } else {
#pragma unroll
            for (int ch = 0; ch < F; ++ch) {
              const float* r0p = fl.gt_lang + ch * plane + (size_t)y0 * fl.lw;
              const float* r1p = fl.gt_lang + ch * plane + (size_t)y1 * fl.lw;
              fl.d_lang[ch * HW + p] = loss_lang_term(acc[4 + ch], r0p, r1p, x0, x1, lx0, lx1, ly0, ly1, wl, sums);
            }
          }
        }
      }
    }
    // the tile's partial sums, in a fixed order: wave trees, then the four waves in rank order
    float* s_loss = s_feat;  // (the feature rows are idle: the last batch ended with a barrier)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < LOSS_SUMS; ++k) {
      float v = sums[k];
#pragma unroll
      for (int mm = 32; mm >= 1; mm >>= 1) v += __shfl_xor(v, mm);
      if ((tid & 63) == 0) s_loss[w * LOSS_SUMS + k] = v;
    }
    __syncthreads();
    if (tid < LOSS_SUMS)
      fl.partials[(size_t)tile_id * LOSS_SUMS + tid] =
          ((s_loss[tid] + s_loss[LOSS_SUMS + tid]) + s_loss[2 * LOSS_SUMS + tid]) + s_loss[3 * LOSS_SUMS + tid];
  }
}

#ifndef OLSR_FWD_TU_LOSS
#error "compile with -DOLSR_FWD_TU_LOSS=0 (images only) or 1 (the instantiations with the fused loss epilogue)"
#endif

#define OLSR_FWD_ARGS                                                                                                  \
  im.ranges, im.ranges, b.inst_gid, b.src, d.W, d.H, d.gx, d.ntiles, g.means2D, g.conic_opacity, g.depths, colors,     \
      s.language_precomp, s.background, im.final_T, im.n_contrib, out_color, out_language, out_depth, out_opacity,     \
      n_touched, b.flags, im.tile_work, order_inout, g.counters, hint_slot,                                             \
      (s.binning == OLSR_BINNING_ELLIPSE ? s.tile_depth_cut : nullptr), &g.counters[9], g.blended

#if OLSR_FWD_TU_LOSS == 0
template <int TILE, int F>
static void launch_fwd_t(const olsr_scene& s, const FrameDims& d, const GeometryState& g, const BinningState& b,
                         const ImageState& im, float* out_color, float* out_language, float* out_depth,
                         float* out_opacity, int32_t* n_touched, uint32_t* order_inout, const uint32_t* hint_slot,
                         hipStream_t st) {
  const float* colors = s.colors_precomp ? s.colors_precomp : g.rgb;
  const FusedLossArgs none{};
  // (depth cut-offs come with the default accumulation only: olsr_api.hip refuses the other combinations)
  const bool cut = s.binning == OLSR_BINNING_ELLIPSE && s.tile_depth_cut != nullptr;
  if (s.flags & OLSR_FLAG_FWD_ACCUM_MFMA)
    render_fwd_kernel<TILE, F, 1, 0, false><<<d.ntiles, 256, 0, st>>>(OLSR_FWD_ARGS, none);
  else if (s.flags & OLSR_FLAG_FWD_ACCUM_WEIGHT)
    render_fwd_kernel<TILE, F, 2, 0, false><<<d.ntiles, 256, 0, st>>>(OLSR_FWD_ARGS, none);
  else if (cut)
    render_fwd_kernel<TILE, F, 0, 0, true><<<d.ntiles, 256, 0, st>>>(OLSR_FWD_ARGS, none);
  else
    render_fwd_kernel<TILE, F, 0, 0, false><<<d.ntiles, 256, 0, st>>>(OLSR_FWD_ARGS, none);
}
#else
template <int TILE, int F>
static void launch_fwd_t(const olsr_scene& s, const FrameDims& d, const GeometryState& g, const BinningState& b,
                         const ImageState& im, float* out_color, float* out_language, float* out_depth,
                         float* out_opacity, int32_t* n_touched, uint32_t* order_inout, const uint32_t* hint_slot,
                         const olsr_loss_fusion& lf, hipStream_t st) {
  const float* colors = s.colors_precomp ? s.colors_precomp : g.rgb;
  const bool lang_term = !lf.tracking && lf.params.F > 0 && lf.gt_language != nullptr;
  FusedLossArgs fl{};
  fl.gt_image = lf.gt_image;
  fl.gt_depth = lf.gt_depth;
  fl.gt_lang = lang_term ? lf.gt_language : nullptr;
  fl.exposure = lf.exposure;
  fl.grad_mask = lf.grad_mask;
  fl.d_image = lf.dL_dimage;
  fl.d_depth = lf.dL_ddepth;
  fl.d_lang = lf.dL_dlanguage;
  fl.partials = reinterpret_cast<float*>(lf.scratch);
  fl.lw = lf.params.lang_width;
  fl.lh = lf.params.lang_height;
  fl.use_exposure = (lf.exposure != nullptr && !lf.params.initialization) ? 1 : 0;
  fl.write_images = lf.skip_images ? 0 : 1;
  fl.alpha = lf.params.alpha;
  fl.thr = lf.params.rgb_boundary_threshold;
  fl.lamda = lf.params.lamda_lang;
  // (depth cut-offs are an option of the tracking loop: olsr_api.hip refuses them with the mapping loss)
  const bool cut = s.binning == OLSR_BINNING_ELLIPSE && s.tile_depth_cut != nullptr;
  if (lf.tracking && cut)
    render_fwd_kernel<TILE, F, 0, 2, true><<<d.ntiles, 256, 0, st>>>(OLSR_FWD_ARGS, fl);
  else if (lf.tracking)
    render_fwd_kernel<TILE, F, 0, 2, false><<<d.ntiles, 256, 0, st>>>(OLSR_FWD_ARGS, fl);
  else
    render_fwd_kernel<TILE, F, 0, 1, false><<<d.ntiles, 256, 0, st>>>(OLSR_FWD_ARGS, fl);
  // (the tiles' partial sums are reduced by one block of the tile-order kernel that follows: launch_render_forward)
}
#endif
#undef OLSR_FWD_ARGS

#if OLSR_FWD_TU_LOSS == 0
#define OLSR_FWD_EXTRA , hint_slot
#define OLSR_FWD_EXTRA_DECL , const uint32_t* hint_slot
#define OLSR_FWD_NAME launch_render_forward_images
#else
#define OLSR_FWD_EXTRA , hint_slot, lf
#define OLSR_FWD_EXTRA_DECL , const uint32_t* hint_slot, const olsr_loss_fusion& lf
#define OLSR_FWD_NAME launch_render_forward_loss
#endif

template <int TILE>
static void launch_fwd_f(const olsr_scene& s, const FrameDims& d, const GeometryState& g, const BinningState& b,
                         const ImageState& im, float* oc, float* ol, float* od, float* oo, int32_t* nt,
                         uint32_t* ord OLSR_FWD_EXTRA_DECL, hipStream_t st) {
  int F = s.F;
#if OLSR_FWD_TU_LOSS == 1
  // The tracking loss reads colour, depth and opacity only (utils/slam_utils.py:92-121).  When the images are not written
  // either, nothing consumes the language accumulation: the RGB instantiation composites the same colour / depth / T with the
  // same decisions (n_contrib, flags, n_touched are bit-identical) on the language scene's state — 19 instead of 34
  // accumulated lane-values per blend.
  if (lf.tracking && lf.skip_images) F = 0;
#endif
  switch (F) {
    case 0: launch_fwd_t<TILE, 0>(s, d, g, b, im, oc, ol, od, oo, nt, ord OLSR_FWD_EXTRA, st); break;
    case 3: launch_fwd_t<TILE, 3>(s, d, g, b, im, oc, ol, od, oo, nt, ord OLSR_FWD_EXTRA, st); break;
    case 15: launch_fwd_t<TILE, 15>(s, d, g, b, im, oc, ol, od, oo, nt, ord OLSR_FWD_EXTRA, st); break;
    case 16: launch_fwd_t<TILE, 16>(s, d, g, b, im, oc, ol, od, oo, nt, ord OLSR_FWD_EXTRA, st); break;
    case 32: launch_fwd_t<TILE, 32>(s, d, g, b, im, oc, ol, od, oo, nt, ord OLSR_FWD_EXTRA, st); break;
    default: break;
  }
}

// (two translation units so that they compile in parallel: the images-only instantiations, and those with the loss epilogue)
void OLSR_FWD_NAME(const olsr_scene& s, const FrameDims& d, const GeometryState& g, const BinningState& b,
                   const ImageState& im, float* out_color, float* out_language, float* out_depth, float* out_opacity,
                   int32_t* n_touched, uint32_t* tile_order_inout OLSR_FWD_EXTRA_DECL, hipStream_t st) {
  if (d.tile == 15)
    launch_fwd_f<15>(s, d, g, b, im, out_color, out_language, out_depth, out_opacity, n_touched, tile_order_inout
                     OLSR_FWD_EXTRA, st);
  else
    launch_fwd_f<16>(s, d, g, b, im, out_color, out_language, out_depth, out_opacity, n_touched, tile_order_inout
                     OLSR_FWD_EXTRA, st);
}

#if OLSR_FWD_TU_LOSS == 0
void launch_render_forward_loss(const olsr_scene& s, const FrameDims& d, const GeometryState& g, const BinningState& b,
                                const ImageState& im, float* out_color, float* out_language, float* out_depth,
                                float* out_opacity, int32_t* n_touched, uint32_t* tile_order_inout,
                                const uint32_t* hint_slot, const olsr_loss_fusion& lf, hipStream_t st);

void launch_render_forward(const olsr_scene& s, const FrameDims& d, const GeometryState& g, const BinningState& b,
                           const ImageState& im, float* out_color, float* out_language, float* out_depth,
                           float* out_opacity, int32_t* n_touched, uint32_t* tile_order_inout, int32_t* num_rendered_dev,
                           const olsr_loss_fusion* loss, hipStream_t st) {
  const RowsMailbox& rm = rows_mailbox_of_this_call();
  if (loss != nullptr)
    launch_render_forward_loss(s, d, g, b, im, out_color, out_language, out_depth, out_opacity, n_touched, tile_order_inout,
                               rm.hint_slot, *loss, st);
  else
    launch_render_forward_images(s, d, g, b, im, out_color, out_language, out_depth, out_opacity, n_touched,
                                 tile_order_inout, rm.hint_slot, st);
  LossFinalArgs lfa{};
  if (loss != nullptr) {
    const olsr_loss_fusion& lf = *loss;
    olsr_loss_params p = lf.params;
    p.F = s.F;  // (the language term is normalised by the channels it sums over)
    const bool lang_term = !lf.tracking && lf.params.F > 0 && lf.gt_language != nullptr;
    const bool use_exposure = lf.exposure != nullptr && !lf.params.initialization;
    lfa = loss_final_args(reinterpret_cast<const float*>(lf.scratch), d.ntiles, p, lf.tracking != 0, lang_term, use_exposure,
                          lf.loss, lf.dL_dexposure);
  }
  const ForwardTailRows rows{b.flags, rm.compact_rows_n, &g.counters[1], s.bwd_mode == OLSR_BWD_REFERENCE && s.tile == 15,
                             b.rowbase, b.row_status, b.tickets + 8, s.backward_row_capacity, g.counters};
  launch_tile_order(im.tile_work, im.tile_order, tile_order_inout, d.ntiles, im.live_rows, rm.dev, rm.seq, g.counters,
                    num_rendered_dev, rm.sticky, rm.hint_slot,
                    (s.binning == OLSR_BINNING_ELLIPSE ? s.tile_depth_cut : nullptr), d.gx, d.gy, lfa,
                    rm.compact_rows_n >= 0 ? &rows : nullptr, st);
}
#endif

}  // namespace olsr

#if defined(OLSR_FWD_STATS) && OLSR_FWD_TU_LOSS == 0
extern "C" void olsr_debug_fwd_stats(unsigned long long* out8, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(olsr::g_fwd_stats), 8 * sizeof(unsigned long long));
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(olsr::g_fwd_stats), z, sizeof(z));
  }
}
#endif
