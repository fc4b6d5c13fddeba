// olsr_device.h — device-side helpers shared by the gfx950 kernels.
//
// Numerics contract: every expression on a decision path (culling, tile
// rectangle, alpha / transmittance thresholds) is strict IEEE fp32 in the reference's source
// order; the build uses -ffp-contract=off so nothing is fused implicitly, and FMAs appear
// only where written (__builtin_fmaf).  `exp` is the fully specified routine below (the same
// constants and operation sequence the CPU oracle pins), because the reference's CUDA libm
// bits cannot be reproduced by any other libm.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace olsr {

// The backward must not trust this frame's lists / rows: the row scratch was too small, the forward overflowed its instance
// capacity or a depth cut-off of the forward hid contributions (all three: counters[7], set by the row compaction), or a
// look-back ran into its spin bound (counters[8], olsr_state.h) — every gradient is then zero.
// (The cut-off miss, counters[9], is folded into counters[7] by the backward's first kernel instead of being tested here: one
//  more scalar load in this predicate moved render_bwd_kernel<15,32,...> from 0.502 to 0.52 ms, round 4.)
// rows_stamp (0 = no statement): the backward was told that the FORWARD compacted the gradient rows for a scratch of a given
// capacity (olsr_scene.backward_row_capacity).  The compaction leaves rows_stamp_of(that capacity) in counters[11], the
// frame's first kernel clears it: a backward whose expectation does not match — the forward ran without the announcement, or
// with another capacity, so rowbase / counters[6] / counters[7] are stale or sized for another scratch — reads and writes
// nothing through them, hands out zero gradients and reports an overflow.
__host__ __device__ __forceinline__ int32_t rows_stamp_of(long long row_capacity) {
  return row_capacity > 0 ? (int32_t)((row_capacity & 0x3FFFFFFFll) + 1) : 0;
}
__device__ __forceinline__ bool frame_unusable(const int32_t* counters, int32_t rows_stamp = 0) {
  return (counters[7] | counters[8]) != 0 || (rows_stamp != 0 && counters[11] != rows_stamp);
}


typedef unsigned int u32;
typedef unsigned long long u64;

// ---- wave64 helpers ---------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ u64 ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool wave_any(bool p) { return ballot(p) != 0ull; }
__device__ __forceinline__ bool wave_all(bool p) { return ballot(!p) == 0ull; }

__device__ __forceinline__ float bits2f(u32 u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ u32 f2bits(float f) { return __builtin_bit_cast(u32, f); }

// Append-to-list slot for the lanes with `flag` set: ONE global atomic per workgroup (<= 16 waves).
// Same-address atomics serialise at the memory side (~5 ns each on MI355X: one per wave of a P = 500 k
// launch is ~40 us); per-wave counts are combined in LDS first.  Must be called by every thread of the block.
__device__ __forceinline__ u32 block_list_slot(bool flag, int32_t* counter) {
  __shared__ u32 s_cnt[16];
  __shared__ u32 s_base;
  const int lane = (int)(threadIdx.x & 63), w = (int)(threadIdx.x >> 6), nw = (int)((blockDim.x + 63) >> 6);
  const u64 m = ballot(flag);
  if (lane == 0) s_cnt[w] = (u32)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 tot = 0;
    for (int i = 0; i < nw; ++i) {
      const u32 c = s_cnt[i];
      s_cnt[i] = tot;
      tot += c;
    }
    s_base = tot ? (u32)atomicAdd(counter, (int)tot) : 0u;
  }
  __syncthreads();
  const u32 slot = s_base + s_cnt[w] + (u32)__popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
  __syncthreads();  // the scratch may be reused by a second call
  return slot;
}

// ---- gfx950 wave reduction of four values at once -----------------------------------------------
// v_permlane32_swap / v_permlane16_swap (new in gfx950) exchange the upper half of one VGPR with the
// lower half of another (halves of 32 lanes, resp. rows of 16 inside each half).  One swap + one add
// folds the lane dimension in half for TWO values at once, with no select and no LDS permute:
//     (a, b) -> a' = {a.lo, b.lo}, b' = {a.hi, b.hi};  a' + b' = { a[l] + a[l+32] | b[l] + b[l+32] }.
// Two levels leave one register holding four values, one per row of 16 lanes; four DPP row
// rotations finish the sum inside each row.  10 instructions per 4 values (vs 24 for four plain
// 6-step reductions), fixed order => bit-reproducible.
// On return every lane of row 0 holds sum(a), row 1 sum(c), row 2 sum(b), row 3 sum(d).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// (the two results are copied to scalars before the bit cast: casting the vector elements in place
//  made this clang add r[0] to itself — verified with scripts/probe/reduce4_probe.hip)
__device__ __forceinline__ float swap32_add(float a, float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b),
                                                  false, false);
  const unsigned x = r[0], y = r[1];
  return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
}
__device__ __forceinline__ float swap16_add(float a, float b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b),
                                                  false, false);
  const unsigned x = r[0], y = r[1];
  return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
}
// (the folds alone, for row_sums3 below: sixteen partial sums per value are left in each row — rows a, c, b, d, and for the
//  pair a in row 0, b in row 2, zeros in rows 1 and 3)
__device__ __forceinline__ float wave_fold4(float a, float b, float c, float d) {
  return swap16_add(swap32_add(a, b), swap32_add(c, d));
}
__device__ __forceinline__ float wave_fold2(float a, float b) { return swap16_add(swap32_add(a, b), 0.0f); }
__device__ __forceinline__ float wave_reduce4(float a, float b, float c, float d) {
  float t = swap16_add(swap32_add(a, b), swap32_add(c, d));
  t += dpp_mov<0x128>(t);  // row_ror:8
  t += dpp_mov<0x124>(t);  // row_ror:4
  t += dpp_mov<0x122>(t);  // row_ror:2
  t += dpp_mov<0x121>(t);  // row_ror:1
  return t;
}
// Two values: one swap folds the halves (lanes 0-31 carry a, 32-63 carry b), four row rotations and
// one row_bcast:15 into rows 1 and 3 finish.  On return row 1 holds sum(a), row 3 sum(b).
__device__ __forceinline__ float wave_reduce2(float a, float b) {
  float t = swap32_add(a, b);
  t += dpp_mov<0x128>(t);  // row_ror:8
  t += dpp_mov<0x124>(t);  // row_ror:4
  t += dpp_mov<0x122>(t);  // row_ror:2
  t += dpp_mov<0x121>(t);  // row_ror:1
  // row_bcast:15, rows 1 and 3 only: lane 15 of the previous row (a full row sum) is added
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x142, 0xa, 0xf, false));
  return t;
}
// The same two reductions with the lane folding done through LDS instead of the permlane swaps.  The swaps are quarter-rate
// VALU operations (8.2 cycles of the SIMD's vector pipe each, profiles/r3_valu_ubench.json) in kernels that are bound by
// that pipe; an LDS exchange costs the pipe nothing: one 16-byte write per lane, then every lane of row r reads the four
// lanes of its column (one per row of 16: ds_read2st64_b32 twice) for the value its row is responsible for.  `wl` is 1 KB
// of LDS private to the wave (a wave's LDS operations execute in order: no barrier, only the compiler is fenced).
// Same result layout as above (rows a, c, b, d), different summation order.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// (the fold alone: on return lane (row r, column c) holds the sum over the four rows of column c, of the value row r is
//  responsible for — sixteen partial sums per value are left to add up inside the row)
__device__ __forceinline__ float wave_fold4_lds(float a, float b, float c, float d, float* wl) {
  const int l = lane_id();
  reinterpret_cast<float4*>(wl)[l] = make_float4(a, b, c, d);
  wave_lds_fence();
  const int r = l >> 4, col = l & 15;
  const float* p = wl + 4 * col + (((r & 1) << 1) | (r >> 1));  // row 0 -> a, 1 -> c, 2 -> b, 3 -> d
  const float t = (p[0] + p[64]) + (p[128] + p[192]);            // lanes col, col + 16, col + 32, col + 48
  wave_lds_fence();
  return t;
}
__device__ __forceinline__ float wave_fold2_lds(float a, float b, float* wl) {  // rows 0, 1 -> a; rows 2, 3 -> b
  const int l = lane_id();
  reinterpret_cast<float2*>(wl)[l] = make_float2(a, b);
  wave_lds_fence();
  const int r = l >> 4, col = l & 15;
  const float* p = wl + 2 * col + (r >> 1);
  const float t = (p[0] + p[32]) + (p[64] + p[96]);
  wave_lds_fence();
  return t;
}
// In-row sums of THREE folded registers in 7 DPP operations instead of 12: after the first halving (row_ror:8) a register
// only needs eight lanes of each row, after the second four — the free lanes take the next register.  row_ror:n moves data
// towards higher lanes (lane i receives lane i - n of its row, like row_shr), so after `m += row_ror:4(m)` the lanes 4-7
// of a row hold t0's four partial sums and the lanes 12-15 t1's; t2 goes into the lanes 0-3; two quad permutes finish.
// On return the lanes with (lane & 12) == 4 hold the row sum of t0, == 12 that of t1, == 0 that of t2.
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {  // quad_perm, all lanes
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_sums3(float t0, float t1, float t2, bool in_hi8, bool in_0to3) {
  const float u0 = t0 + dpp_mov<0x128>(t0);  // row_ror:8: period 8 inside the row from here on
  const float u1 = t1 + dpp_mov<0x128>(t1);
  const float u2 = t2 + dpp_mov<0x128>(t2);
  float m = in_hi8 ? u1 : u0;                // lanes 0-7: t0, lanes 8-15: t1
  m += dpp_mov<0x124>(m);                    // row_ror:4: complete in lanes 4-7 (t0) and 12-15 (t1)
  const float v2 = u2 + dpp_mov<0x124>(u2);  // period 4: complete everywhere
  m = in_0to3 ? v2 : m;                      // lanes 0-3: t2
  m += dpp_quad<0x4E>(m);                    // quad_perm [2,3,0,1]
  m += dpp_quad<0xB1>(m);                    // quad_perm [1,0,3,2]
  return m;
}
__device__ __forceinline__ float wave_reduce4_lds(float a, float b, float c, float d, float* wl) {
  float t = wave_fold4_lds(a, b, c, d, wl);
  t += dpp_mov<0x128>(t);  // row_ror:8
  t += dpp_mov<0x124>(t);  // row_ror:4
  t += dpp_mov<0x122>(t);  // row_ror:2
  t += dpp_mov<0x121>(t);  // row_ror:1
  return t;
}
// rows 0 and 1 hold sum(a), rows 2 and 3 sum(b) (wave_reduce2's contract — row 1: a, row 3: b — is contained)
__device__ __forceinline__ float wave_reduce2_lds(float a, float b, float* wl) {
  float t = wave_fold2_lds(a, b, wl);
  t += dpp_mov<0x128>(t);
  t += dpp_mov<0x124>(t);
  t += dpp_mov<0x122>(t);
  t += dpp_mov<0x121>(t);
  return t;
}
typedef float v2f __attribute__((ext_vector_type(2)));
// lane that holds value i (0..3) of a wave_reduce4 result: rows are ordered a, c, b, d
__device__ __forceinline__ int reduce4_lane(int i) { return 16 * (((i & 1) << 1) | ((i >> 1) & 1)); }
__device__ __forceinline__ float lane_read(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// ---- pinned exp --------------------------------------------------------------------------
// Cephes-style expf: n = rne(x*log2e); r = x - n*ln2 (two-term); degree-5 polynomial;
// scale by 2^n.  The oracle clamps to [-87, 88] so 2^n is a normal number; here only the lower clamp is
// kept (one v_max_f32): both composites discard the result whenever power > 0 (CR/forward.cu:457-458), so
// arguments above 88 are never consumed, and for every consumed argument the bits equal the oracle's.
// max(x, -87) as ONE v_max_f32.  __builtin_fmaxf costs two: the compiler puts a canonicalising v_max_f32 x, x in front, because
// maxNum must quiet a signalling NaN operand and it cannot see that x is an arithmetic result; the instruction itself, in the
// IEEE mode kernels run in, returns the other operand for any NaN.  Same bits for every input that is not a signalling NaN.
__device__ __forceinline__ float max_m87(float x) {
  float r;
  asm("v_max_f32 %0, 0xc2ae0000, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ float pinned_expf(float x) {
  x = max_m87(x);
  float n = __builtin_rintf(x * 1.44269504088896341f);
  float r = __builtin_fmaf(n, -0.693359375f, x);
  r = __builtin_fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = __builtin_fmaf(p, r, 1.3981999507e-3f);
  p = __builtin_fmaf(p, r, 8.3334519073e-3f);
  p = __builtin_fmaf(p, r, 4.1665795894e-2f);
  p = __builtin_fmaf(p, r, 1.6666665459e-1f);
  p = __builtin_fmaf(p, r, 5.0000001201e-1f);
  float e = __builtin_fmaf(p, r * r, r) + 1.0f;
  int ni = (int)n;
  return e * bits2f((u32)(ni + 127) << 23);
}

// the same routine on two values at once (packed fp32: every component is the IEEE operation of pinned_expf)
typedef float v2f_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f_ pinned_expf2(v2f_ x) {
  x.x = max_m87(x.x);
  x.y = max_m87(x.y);
  const v2f_ t = x * v2f_{1.44269504088896341f, 1.44269504088896341f};
  const v2f_ n = {__builtin_rintf(t.x), __builtin_rintf(t.y)};
  v2f_ r = __builtin_elementwise_fma(n, v2f_{-0.693359375f, -0.693359375f}, x);
  r = __builtin_elementwise_fma(n, v2f_{2.12194440e-4f, 2.12194440e-4f}, r);
  v2f_ p = {1.9875691500e-4f, 1.9875691500e-4f};
  p = __builtin_elementwise_fma(p, r, v2f_{1.3981999507e-3f, 1.3981999507e-3f});
  p = __builtin_elementwise_fma(p, r, v2f_{8.3334519073e-3f, 8.3334519073e-3f});
  p = __builtin_elementwise_fma(p, r, v2f_{4.1665795894e-2f, 4.1665795894e-2f});
  p = __builtin_elementwise_fma(p, r, v2f_{1.6666665459e-1f, 1.6666665459e-1f});
  p = __builtin_elementwise_fma(p, r, v2f_{5.0000001201e-1f, 5.0000001201e-1f});
  const v2f_ e = __builtin_elementwise_fma(p, r * r, r) + v2f_{1.0f, 1.0f};
  const v2f_ sc = {bits2f((u32)((int)n.x + 127) << 23), bits2f((u32)((int)n.y + 127) << 23)};
  return e * sc;
}

// float -> int, truncating and saturating (v_cvt_i32_f32 semantics; NaN -> 0), written out
// so that the conversion is defined for every input.
__device__ __forceinline__ int f2i_sat(float v) {
  if (v != v) return 0;
  if (v >= 2147483648.0f) return 2147483647;
  if (v <= -2147483648.0f) return (int)0x80000000;
  return (int)v;
}

__device__ __forceinline__ float fminf_ref(float a, float b) { return (b < a) ? b : a; }  // std::min(a,b)
__device__ __forceinline__ float fmaxf_ref(float a, float b) { return (a < b) ? b : a; }  // std::max(a,b)

// ---- geometry helpers (CR/auxiliary.h:41-97) ---------------------------------------------
__device__ __forceinline__ float ndc2Pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

struct Rect { int x0, y0, x1, y1; };
template <int TILE>
__device__ __forceinline__ Rect get_rect(float px, float py, int max_radius, int gx, int gy) {
  Rect r;
  r.x0 = min(gx, max(0, f2i_sat((px - (float)max_radius) / (float)TILE)));
  r.y0 = min(gy, max(0, f2i_sat((py - (float)max_radius) / (float)TILE)));
  // `p.x + max_radius + BLOCK_X - 1` is three float operations in the reference's source order (+ radius, + BLOCK, - 1),
  // not `+ (BLOCK - 1)`: the two differ in the last bit when the sum crosses a power of two, and a whole tile column
  // hangs on that bit (found by scripts/oracle_stress.py: mean x = 58.999992, radius 54, 16-pixel tiles)
  r.x1 = min(gx, max(0, f2i_sat((((px + (float)max_radius) + (float)TILE) - 1.0f) / (float)TILE)));
  r.y1 = min(gy, max(0, f2i_sat((((py + (float)max_radius) + (float)TILE) - 1.0f) / (float)TILE)));
  return r;
}

// ---- exact tile binning (OLSR_BINNING_ELLIPSE) ---------------------------------------------
// The reference bins a Gaussian into every tile of the square that bounds a circle of radius
// ceil(3 sqrt(lambda_max)) (get_rect).  A pixel can only blend the Gaussian if
//   alpha = opacity * exp(power) >= 1/255  <=>  q(d) = a dx^2 + 2 b dx dy + c dy^2 <= 2 ln(255 opacity),
// an ellipse that is usually much smaller than that square (anisotropy, low opacity).  The tiles
// the ellipse misses contribute nothing in the forward and are "whole tile skips" in the
// backward (CR/backward.cu:1087-1093), so dropping them changes no output bit.
//
// Everything here is CONSERVATIVE interval arithmetic in fp32: every rounding is covered by an
// explicit outward pad, so a tile is dropped only if no pixel of it can pass the composite's own
// fp32 alpha test.  The parity suite checks exactly that (bit-identical images, n_touched and
// flags against the reference binning).
//
// cull_threshold: 2 * (L + margins), L = ln(255 opacity).  2e-3 + 1e-4|L| covers __logf, the pinned
// exp and the opacity product; 5e-7 (a + c + |b|) D^2 bounds the rounding of the three products
// and two sums of the composite's `power` at distances up to D = radius + TILE + 1.  Negative:
// no pixel can pass.
__device__ __forceinline__ float cull_threshold(float a, float b, float c, float opacity, int radius, int tile) {
  // not-a-number anywhere: the composite's comparisons then all fail and it blends alpha = 0.99 (fminf_ref): keep
  // every tile of the rect (3e38 makes cull_setup fall back to the full spans)
  if (a != a || b != b || c != c || opacity != opacity) return 3e38f;
  const float o255 = 255.0f * opacity;
  if (!(o255 > 0.0f)) return -1.0f;
  const float L = __logf(o255);
  const float D = (float)radius + (float)tile + 1.0f;
  // (|a| + |c|: a conic that came out of a rounded-to-negative determinant is not positive definite; such
  //  Gaussians keep their whole rect below, but the bound must not turn negative here)
  const float thr = L + 2e-3f + 1e-4f * fabsf(L) + 5e-7f * (fabsf(a) + fabsf(c) + fabsf(b)) * D * D;
  if (thr != thr) return 3e38f;
  if (!(thr >= 0.0f)) return -1.0f;
  return fminf(2.0f * thr * (1.0f + 1e-6f), 3e38f);
}

struct CullEllipse {
  float px, py, b, inv_a, A, det_lo, xstar, ystar, ymax;
  bool exact;  // false: degenerate / out-of-range conic, every row keeps its full rect span
};
__device__ __forceinline__ CullEllipse cull_setup(float px, float py, float a, float b, float c, float t2, int radius) {
  CullEllipse e;
  e.px = px;
  e.py = py;
  e.b = b;
  // a c - b^2 with Kahan's fused difference of products (relative error <= 2 ulp), rounded down
  const float w = b * b;
  const float err = __builtin_fmaf(-b, b, w);
  const float det = __builtin_fmaf(a, c, -w) + err;
  e.det_lo = det * (1.0f - 4e-7f);
  e.exact = (e.det_lo > 0.0f) && (a > 1e-30f) && (c > 1e-30f) && (t2 >= 0.0f) && (t2 < 1e6f) && (a < 1e6f) &&
            (c < 1e6f) && (radius < (1 << 20)) && (fabsf(px) < 1e7f) && (fabsf(py) < 1e7f);
  // v_rcp_f32 / v_sqrt_f32 are accurate to 1 ulp; the outward factors cover that
  e.inv_a = e.exact ? __builtin_amdgcn_rcpf(a) : 0.0f;
  e.A = a * t2 * (1.0f + 4e-7f);  // upper bound of a * t2
  const float inv_det = e.exact ? __builtin_amdgcn_rcpf(e.det_lo) * (1.0f + 4e-7f) : 0.0f;
  e.xstar = e.exact ? __builtin_amdgcn_sqrtf(t2 * c * inv_det) * (1.0f + 1e-6f) : 0.0f;  // rightmost point, rounded out
  e.ystar = e.exact ? -b * __builtin_amdgcn_rcpf(c) * e.xstar : 0.0f;                    // its dy
  e.ymax = e.exact ? __builtin_amdgcn_sqrtf(e.A * inv_det) * (1.0f + 1e-6f) : 3e38f;     // topmost point, rounded out
  return e;
}
// Tile rows [ya, yb) of the reference rect rows [ry0, ry1) that the ellipse's y extent can reach; rows
// outside hold no instance.
template <int TILE>
__device__ __forceinline__ void cull_rows(const CullEllipse& e, int ry0, int ry1, int& ya, int& yb) {
  ya = ry0;
  yb = ry1;
  if (!e.exact || !(e.ymax < 1e7f)) return;
  const float pad = 1e-2f + 4e-7f * (fabsf(e.py) + e.ymax);
  const float lo = floorf((e.py - e.ymax - pad) / (float)TILE), hi = floorf((e.py + e.ymax + pad) / (float)TILE);
  // tile row t covers pixel rows [t * TILE, t * TILE + TILE - 1]: a point below row t's last pixel row and
  // above row t+1's first one belongs to neither, so flooring is conservative on both ends
  ya = max(ry0, (int)fmaxf(lo, -1.0f));
  yb = min(ry1, (int)fminf(hi, 1e6f) + 1);
  if (yb < ya) yb = ya;
}

// Tile columns [xa, xb) of tile row ty (inside the reference rect columns [rx0, rx1)) that hold at
// least one pixel column of ellipse ∩ band, the band being the continuous range of the row's pixel
// rows.  ellipse ∩ band is convex: its projection on x is one interval, and every tile column that
// overlaps the interval is hit.
template <int TILE>
__device__ __forceinline__ void cull_row_span(const CullEllipse& e, int rx0, int rx1, int ty, int W, int H, int& xa,
                                              int& xb) {
  xa = rx0;
  xb = rx1;
  if (!e.exact) return;
  const float y_lo = (float)(ty * TILE), y_hi = (float)min(ty * TILE + TILE - 1, H - 1);
  const float pad_y = 1e-3f + 2e-7f * (fabsf(e.py) + y_hi);
  const float dy0 = (y_lo - e.py) - pad_y, dy1 = (y_hi - e.py) + pad_y;  // band, widened
  // discriminants at the band edges, rounded up
  const float disc0 = (e.A - e.det_lo * dy0 * dy0 * (1.0f - 4e-7f)) + 2e-7f * e.A;
  const float disc1 = (e.A - e.det_lo * dy1 * dy1 * (1.0f - 4e-7f)) + 2e-7f * e.A;
  if (!(disc0 == disc0) || !(disc1 == disc1)) return;  // NaN: keep the full span
  float hi = -3e38f, lo = 3e38f;
  bool any = false;
  if (disc0 >= 0.0f) {
    const float s = __builtin_amdgcn_sqrtf(disc0) * (1.0f + 4e-7f), bd = e.b * dy0;
    const float m = 1e-6f * (fabsf(bd) + s) * e.inv_a;
    hi = (s - bd) * e.inv_a + m;
    lo = (-s - bd) * e.inv_a - m;
    any = true;
  }
  if (disc1 >= 0.0f) {
    const float s = __builtin_amdgcn_sqrtf(disc1) * (1.0f + 4e-7f), bd = e.b * dy1;
    const float m = 1e-6f * (fabsf(bd) + s) * e.inv_a;
    hi = fmaxf(hi, (s - bd) * e.inv_a + m);
    lo = fminf(lo, (-s - bd) * e.inv_a - m);
    any = true;
  }
  const float pad_s = 1e-3f + 1e-5f * fabsf(e.ystar);
  if (dy0 <= e.ystar + pad_s && e.ystar - pad_s <= dy1) {
    hi = fmaxf(hi, e.xstar);
    any = true;
  }
  if (dy0 <= -e.ystar + pad_s && -e.ystar - pad_s <= dy1) {
    lo = fminf(lo, -e.xstar);
    any = true;
  }
  if (!(hi == hi) || !(lo == lo)) return;  // NaN: keep the full span
  if (!any || hi < lo) {                    // the band misses the ellipse
    xb = xa;
    return;
  }
  // integer pixel columns inside [px + lo, px + hi], padded, clipped to the image
  const float pad_x = 1e-3f + 2e-7f * (fabsf(e.px) + fmaxf(fabsf(hi), fabsf(lo)));
  const float Xlo = fmaxf(ceilf(e.px + lo - pad_x), 0.0f), Xhi = fminf(floorf(e.px + hi + pad_x), (float)(W - 1));
  if (Xhi < Xlo) {
    xb = xa;
    return;
  }
  xa = max(rx0, (int)Xlo / TILE);
  xb = min(rx1, (int)Xhi / TILE + 1);
  if (xb < xa) xb = xa;
}

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

__device__ __forceinline__ f3 transformPoint4x3(const f3& p, const float* m) {
  return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
          m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
__device__ __forceinline__ f4 transformPoint4x4(const f3& p, const float* m) {
  return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
          m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]};
}
__device__ __forceinline__ f3 transformVec4x3Transpose(const f3& p, const float* m) {
  return {m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
          m[8] * p.x + m[9] * p.y + m[10] * p.z};
}

// column-major 3x3 with glm's accumulation order: (a*b)[c][r] = a[0][r]b[c][0] + a[1][r]b[c][1] + a[2][r]b[c][2]
struct m3 { float c[3][3]; };
__device__ __forceinline__ m3 mul(const m3& a, const m3& b) {
  m3 o;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) o.c[c][r] = a.c[0][r] * b.c[c][0] + a.c[1][r] * b.c[c][1] + a.c[2][r] * b.c[c][2];
  return o;
}
__device__ __forceinline__ m3 transpose(const m3& a) {
  m3 o;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) o.c[c][r] = a.c[r][c];
  return o;
}

// Shared by forward preprocess and the backward recomputation (CR/forward.cu:77-116,
// CR/backward.cu:171-206).
struct Cov2D {
  f3 t;
  float txtz, tytz;
  m3 J, Wm, T, Vrk, cov;
};
__device__ __forceinline__ void cov2d_common(const f3& mean, float focal_x, float focal_y, float tan_fovx,
                                             float tan_fovy, const float* cov3D, const float* view, Cov2D& o) {
  f3 t = transformPoint4x3(mean, view);
  const float limx = 1.3f * tan_fovx;
  const float limy = 1.3f * tan_fovy;
  o.txtz = t.x / t.z;
  o.tytz = t.y / t.z;
  t.x = fminf_ref(limx, fmaxf_ref(-limx, o.txtz)) * t.z;
  t.y = fminf_ref(limy, fmaxf_ref(-limy, o.tytz)) * t.z;
  o.t = t;
  o.J = {{{focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z)},
          {0.0f, focal_y / t.z, -(focal_y * t.y) / (t.z * t.z)},
          {0.0f, 0.0f, 0.0f}}};
  o.Wm = {{{view[0], view[4], view[8]}, {view[1], view[5], view[9]}, {view[2], view[6], view[10]}}};
  o.T = mul(o.Wm, o.J);
  o.Vrk = {{{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]}, {cov3D[2], cov3D[4], cov3D[5]}}};
  o.cov = mul(mul(transpose(o.T), transpose(o.Vrk)), o.T);
}

// SH constants, CR/auxiliary.h:22-39
__device__ constexpr float SH_C0 = 0.28209479177387814f;
__device__ constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                       -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                       0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                       -0.5900435899266435f};

// Per-tile depth cut-offs (include/olsr.h): the test preprocess applies to a (tile, Gaussian) pair.  NaN cut-offs keep
// everything.
__device__ __forceinline__ bool depth_cut_keeps(float depth, float cut) { return !(depth > cut); }

// XCD-aware tile remap: workgroup b is observed to run on XCD b % 8; give each XCD a
// contiguous run of tiles so neighbouring tiles (which share Gaussians) share an L2.
// Bijective for any n (cdna_hip_programming.md §5 "XCD swizzle must be bijective").
__device__ __forceinline__ int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7;
  const int xcd = b & 7, k = b >> 3;
  const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + k;
}

// ---- caller-side activations folded into preprocess (OLSR_ACT_*, include/olsr.h) -----------------------
__device__ __forceinline__ float act_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ void act_normalize4(const float* q, float* o) {  // F.normalize(q, dim=-1), eps 1e-12
  const float n = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = q[k] / n;
}
// d/dq_raw of normalize: (g - qhat <qhat, g>) / |q|
__device__ __forceinline__ void act_normalize4_backward(const float* q, const float* g, float* o) {
  const float n = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
  float h[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) h[k] = q[k] / n;
  const float dot = h[0] * g[0] + h[1] * g[1] + h[2] * g[2] + h[3] * g[3];
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = (g[k] - h[k] * dot) / n;
}

// ---- the ranks that survive the reference's 225-lane reduction tree ---------------------------
// render_cuda_reduce_sum halves g.size() = 225 with integer division (CR/backward.cu:691-702): the steps
// 112, 56, 28, 14, 7, 3, 1 drop rank 224 and every rank whose residue mod 7 is 2, 5 or 6.  Exactly
// 128 ranks reach element 0; everything the other 97 pixels compute in the reference's backward is
// discarded (and its language gradient only ever comes from rank 0, a survivor).  So in reference
// mode the backward composite runs on the survivors alone, packed into two full waves:
//   packed index s in [0, 128)  <->  rank = 7 * (s / 4) + {0, 1, 3, 4}[s % 4].
__device__ __forceinline__ bool ref15_survives(int rank) {
  const int m = rank % 7;
  return rank < 224 && (m == 0 || m == 1 || m == 3 || m == 4);
}
__device__ __forceinline__ int ref15_rank_of_packed(int s) {
  const int k = s & 3;
  return 7 * (s >> 2) + k + (k >> 1);  // 0, 1, 3, 4
}
__device__ __forceinline__ int ref15_packed_of_rank(int rank) {  // survivors only
  const int m = rank % 7;
  return 4 * (rank / 7) + m - (m >= 3 ? 1 : 0);  // 0->0, 1->1, 3->2, 4->3
}

// Which pixel of its tile a thread of a 256-thread composite workgroup owns (the forward composite and the backward composites
// that are not survivor-packed; both must agree, the forward's slot bits in flags[] say which WAVE blended an instance).
// Round 1-4: wave w = the 64 consecutive ranks 64 w .. 64 w + 63, i.e. a strip 15 (16) pixels wide and ~4 rows tall — and, for
// 15x15 tiles, a fourth wave with 33 pixels.  Since late round 4: wave w = one QUADRANT of the tile (8x8, 7x8, 8x7, 7x7 for
// 15x15 tiles; four 8x8 for 16x16).  A splat's footprint edge crosses a compact region less often than a strip, so fewer
// (entry, wave) pairs are visited and more lanes of a visit blend; the waves' pixel counts are 64 / 56 / 56 / 49 instead of
// 64 / 64 / 64 / 33.  Per-pixel results do not depend on the assignment (each pixel composites its list in order); tile rank 0
// stays thread 0.  Returns TILE * TILE for a lane without a pixel.  OLSR_SLOT_QUADRANTS=0 restores the strips.
#ifndef OLSR_SLOT_QUADRANTS
#define OLSR_SLOT_QUADRANTS 1
#endif
template <int TILE>
__device__ __forceinline__ int slot_rank(int tid) {
#if OLSR_SLOT_QUADRANTS
  const int w = tid >> 6, l = tid & 63;
  const int qx = w & 1, qy = w >> 1;
  int lx, ly;
  bool valid = true;
  if constexpr (TILE == 16) {
    lx = l & 7;
    ly = l >> 3;
  } else {
    const int qw = qx ? TILE - 8 : 8, qh = qy ? TILE - 8 : 8;
    ly = l / qw;
    lx = l - ly * qw;
    valid = ly < qh;
  }
  return valid ? (qy * 8 + ly) * TILE + qx * 8 + lx : TILE * TILE;
#else
  return tid;
#endif
}

constexpr int round_up(int v, int m) { return (v + m - 1) / m * m; }
// staged per-Gaussian feature row: [r, g, b, depth, lang[F]] padded to a multiple of 4 floats
constexpr int feat_row(int F) { return round_up(4 + F, 4); }
// partial-gradient row written per (tile, Gaussian) instance by the backward composite:
// [mean2D.x, mean2D.y, conic.x, conic.y, conic.w, opacity, r, g, b, depth, lang[F]] padded to a multiple of 4 floats
// (16-byte loads in the per-Gaussian sums; round 3: 112 instead of 128 bytes at F = 15)
constexpr int grad_row(int F) { return round_up(10 + F, 4); }

}  // namespace olsr
