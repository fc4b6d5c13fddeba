// k_order_carry.hip — the depth order of the previous frame of a view, repaired under this frame's keys
// (olsr_scene.depth_order_carry, include/olsr.h "Carried depth order").
//
// Replaces — when it succeeds — the P-sized half of cub::DeviceRadixSort::SortPairs (CR/rasterizer_impl.cu:478-483; here:
// sort_hist_kernel + four sort_pass_kernel launches, k_sort.hip) for the dependent loops of the reference: the tracking
// iterations of a frame (utils/slam_frontend.py:163-277) and the mapping iterations over a window of keyframes
// (utils/slam_backend.py:499-670).  Between two iterations the pose / the parameters move by one optimiser step, so the order
// (depth bits, Gaussian index) of the previous iteration is nearly this iteration's.
//
//   phase A   block b merge-sorts ranks [b W, (b + 1) W) of the old order under the NEW keys   -> (key, id) pairs
//   phase B   block j merge-sorts ranks [j W - W / 2, j W + W / 2) of phase A's output         -> the new order
//
// If no Gaussian is more than W / 2 ranks away from where it belongs, the two phases are a complete sort (after phase A the
// first half-window of every aligned window holds exactly the elements that belong there or in the half-window before it;
// DESIGN.md section 12 has the argument).  Nothing is assumed, though: phase B PROVES the result — inside its window the
// sorted pairs must ascend strictly, and its last pair must be smaller than the smallest pair of the next window, which it
// reads from phase A's output (two sorted half-windows: two loads).  All P adjacent pairs strictly ascending means sorted AND
// no index twice, i.e. a permutation of [0, P) in the reference's order.  Anything else — a Gaussian that moved further, a
// carry array that holds zeros, garbage or the order of another scene — raises `miss`, and the radix passes that are
// enqueued behind these two launches run instead of returning at once (k_sort.hip: run_if).  The result never depends on the
// carried array.
//
// Windows are independent: no ticket, no published counts, no look-back — the two launches are bound by one gather of the
// keys and a handful of LDS round trips per element.  The merge sort is the rank form (an element's place in the merged run
// = its offset in its own run + the number of elements of the sibling run in front of it), with a GALLOPING search that
// starts where a nearly sorted input has its answer: the left run's elements expect nothing of the right run in front of
// them, the right run's elements expect the whole left run; one LDS read settles that case, and an element that did move
// pays 2 log2(distance) reads.  Stable (left < right on ties), so even an input with repeated indices comes out as a
// rearrangement of itself and the strictness test sees the repeat.
#include <cstdlib>

#include "olsr_device.h"
#include "olsr_kernels.h"

namespace olsr {

constexpr int OC_W = 2048;  // ranks per window: two emission blocks (EMIT_CHUNK), so phase B's half-window offset stays aligned
static_assert(OC_W == 2 * EMIT_CHUNK, "phase B leaves one instance total per emission block");

// # elements of the ascending run B[0, n) that are < x, looked for from the FRONT
__device__ __forceinline__ int oc_count_less_front(const u64* __restrict__ B, int n, u64 x) {
  if (!(B[0] < x)) return 0;
  int lo = 0, step = 1;  // B[lo] < x
  while (lo + step < n && B[lo + step] < x) {
    lo += step;
    step <<= 1;
  }
  int hi = lo + step < n ? lo + step : n;  // B[hi] >= x, or hi == n
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (B[mid] < x) lo = mid;
    else hi = mid;
  }
  return hi;
}
// # elements of the ascending run A[0, n) that are <= x, looked for from the BACK
__device__ __forceinline__ int oc_count_leq_back(const u64* __restrict__ A, int n, u64 x) {
  if (A[n - 1] <= x) return n;
  int hi = n - 1, step = 1;  // A[hi] > x
  while (hi - step >= 0 && A[hi - step] > x) {
    hi -= step;
    step <<= 1;
  }
  int lo = hi - step >= 0 ? hi - step : -1;  // A[lo] <= x, or lo == -1
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (A[mid] <= x) lo = mid;
    else hi = mid;
  }
  return hi;
}

// Stable merge sort of the OC_W pairs in `a` (LDS), ping-pong with `b`; returns the buffer that holds the result.
// Thread t owns elements t, t + T, ... of every level (consecutive lanes -> consecutive 8-byte words: conflict-free).
template <int T>
__device__ __forceinline__ u64* oc_window_sort(u64* a, u64* b) {
  constexpr int E = OC_W / T;
  const int tid = threadIdx.x;
#pragma unroll 1
  for (int run = 1; run < OC_W; run <<= 1) {
    u64 x[E];
    int dest[E];
#pragma unroll
    for (int k = 0; k < E; ++k) x[k] = a[tid + k * T];
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const int i = tid + k * T;
      const int base = i & ~(2 * run - 1);
      if ((i & run) == 0) dest[k] = i + oc_count_less_front(a + base + run, run, x[k]);           // left run: offset i - base
      else dest[k] = i - run + oc_count_leq_back(a + base, run, x[k]);                            // right run: offset i - base - run
    }
#pragma unroll
    for (int k = 0; k < E; ++k) b[dest[k]] = x[k];
    __syncthreads();
    u64* t = a;
    a = b;
    b = t;
  }
  return a;
}

// pairs that stand in for ranks outside [0, P): below every real pair (real keys are >= 1, k_preprocess.hip) / above them,
// distinct, and already in order
__device__ __forceinline__ u64 oc_pad_low(int l) { return (u64)(u32)l; }
__device__ __forceinline__ u64 oc_pad_high(int l) { return 0xFFFFFFFF00000000ull | (u64)(u32)l; }

template <int T>
__global__ __launch_bounds__(T) void order_repair_a_kernel(int P, const u32* __restrict__ carry,
                                                           const u32* __restrict__ keys, u32* __restrict__ out_key,
                                                           u32* __restrict__ out_gid, u32* __restrict__ miss) {
  constexpr int E = OC_W / T;
  __shared__ __attribute__((aligned(16))) u64 s0[OC_W];
  __shared__ __attribute__((aligned(16))) u64 s1[OC_W];
  const int tid = threadIdx.x;
  const int base = (int)blockIdx.x * OC_W;
  bool bad = false;
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const int l = tid + k * T, i = base + l;
    u64 c = oc_pad_high(l);
    if (i < P) {
      const u32 g = carry[i];
      const bool ok = g < (u32)P;  // (an array that never held an order: index nothing with it)
      bad |= !ok;
      c = ((u64)(ok ? keys[g] : 0xFFFFFFFFu) << 32) | (u64)g;
    }
    s0[l] = c;
  }
  __syncthreads();
  const u64* r = oc_window_sort<T>(s0, s1);
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const int l = tid + k * T, i = base + l;
    if (i < P) {
      const u64 c = r[l];
      out_key[i] = (u32)(c >> 32);
      out_gid[i] = (u32)c;
    }
  }
  if (wave_any(bad) && lane_id() == 0) atomicOr(miss, 1u);
}

// totals (may be null): per emission block of EMIT_CHUNK ranks, the instances its Gaussians emit (what the last radix pass
// leaves in emit_status for emit_offsets_kernel)
template <int T>
__global__ __launch_bounds__(T) void order_repair_b_kernel(int P, const u32* __restrict__ in_key,
                                                           const u32* __restrict__ in_gid, u32* __restrict__ carry,
                                                           const u32* __restrict__ inst_count, u64* __restrict__ totals,
                                                           u32* __restrict__ miss) {
  constexpr int E = OC_W / T;
  __shared__ __attribute__((aligned(16))) u64 s0[OC_W];
  __shared__ __attribute__((aligned(16))) u64 s1[OC_W];
  __shared__ u64 s_tot[2][T / 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int start = (int)blockIdx.x * OC_W - OC_W / 2;
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const int l = tid + k * T, i = start + l;
    u64 c;
    if (i < 0) c = oc_pad_low(l);
    else if (i < P) c = ((u64)in_key[i] << 32) | (u64)in_gid[i];
    else c = oc_pad_high(l);
    s0[l] = c;
  }
  __syncthreads();
  const u64* r = oc_window_sort<T>(s0, s1);
  bool bad = false;
  u64 tot[2] = {0ull, 0ull};  // (emit_total_pack, olsr_state.h: instances | emitting Gaussians << 40)
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const int l = tid + k * T, i = start + l;
    if (i >= 0 && i < P) {
      const u64 c = r[l];
      const u32 g = (u32)c;
      carry[i] = g;
      if (l > 0 && i > 0) bad |= !(r[l - 1] < c);  // (rank i - 1 is in this window unless l == 0: the test below covers that pair)
      bad |= !(g < (u32)P);
      if (g < (u32)P) tot[l >= OC_W / 2 ? 1 : 0] += emit_total_pack(inst_count[g]);
    }
  }
  // the pair across the window's upper edge: this window's largest against the next window's smallest, which is the smaller
  // head of the two sorted half-windows of phase A's output it is made of
  if (tid == 0) {
    const int nb = start + OC_W;  // first rank of the next window
    if (nb < P) {
      u64 nmin = ((u64)in_key[nb] << 32) | (u64)in_gid[nb];
      if (nb + OC_W / 2 < P) {
        const u64 o = ((u64)in_key[nb + OC_W / 2] << 32) | (u64)in_gid[nb + OC_W / 2];
        nmin = o < nmin ? o : nmin;
      }
      bad |= !(r[OC_W - 1] < nmin);
    }
  }
  if (wave_any(bad) && lane == 0) atomicOr(miss, 1u);
  if (totals != nullptr) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      u64 t = tot[h];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m);
      if (lane == 0) s_tot[h][w] = t;
    }
    __syncthreads();
    if (tid < 2) {
      u64 t = 0;
      for (int i = 0; i < T / 64; ++i) t += s_tot[tid][i];
      const int first = start + tid * (OC_W / 2);  // first rank of this half-window = an emission block's first rank
      if (first >= 0 && first < P) totals[first / EMIT_CHUNK] = t;
    }
  }
}

void launch_order_repair(int P, uint32_t* carry, const uint32_t* keys, uint32_t* tmp_key, uint32_t* tmp_gid,
                         const uint32_t* inst_count, uint32_t* totals, uint32_t* miss, bool frames_in_flight,
                         hipStream_t st) {
  if (P <= 0) return;
  const int na = (P + OC_W - 1) / OC_W;
  const int nb = (P + OC_W / 2 + OC_W - 1) / OC_W;  // windows [j W - W / 2, j W + W / 2) that reach below P
  // (four-wave workgroups beside another frame's composite, sixteen waves alone: the same choice as the radix passes,
  //  OLSR_FLAG_FRAMES_IN_FLIGHT)
  static const int forced = [] {  // (experiment knob, read once: OLSR_CARRY_THREADS=256 / 512 / 1024)
    const char* e = std::getenv("OLSR_CARRY_THREADS");
    return e ? std::atoi(e) : 0;
  }();
  const int T = forced ? forced : (frames_in_flight ? 256 : 1024);
  if (T == 512) {
    order_repair_a_kernel<512><<<na, 512, 0, st>>>(P, carry, keys, tmp_key, tmp_gid, miss);
    order_repair_b_kernel<512><<<nb, 512, 0, st>>>(P, tmp_key, tmp_gid, carry, inst_count, reinterpret_cast<u64*>(totals), miss);
  } else if (T == 256) {
    order_repair_a_kernel<256><<<na, 256, 0, st>>>(P, carry, keys, tmp_key, tmp_gid, miss);
    order_repair_b_kernel<256><<<nb, 256, 0, st>>>(P, tmp_key, tmp_gid, carry, inst_count, reinterpret_cast<u64*>(totals), miss);
  } else {
    order_repair_a_kernel<1024><<<na, 1024, 0, st>>>(P, carry, keys, tmp_key, tmp_gid, miss);
    order_repair_b_kernel<1024><<<nb, 1024, 0, st>>>(P, tmp_key, tmp_gid, carry, inst_count, reinterpret_cast<u64*>(totals), miss);
  }
}

}  // namespace olsr
