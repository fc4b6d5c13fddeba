// k_binning.hip — instance offsets, emission, (tile, depth) ordering and tile ranges.
//
// Replaces cub::DeviceScan::InclusiveSum, duplicateWithKeys, cub::DeviceRadixSort::SortPairs
// and identifyTileRanges (CR/rasterizer_impl.cu:451-493, 70-138).  The reference sorts R
// 64-bit (tile | depth) keys; the required order is (tile, depth bits, Gaussian index).
// MI355X-first formulation with the same result:
//   1. stable radix sort of the P Gaussians by depth bits (P << R);
//   2. emit instances in that order (so emission order is already (depth, index) sorted);
//   3. stable radix sort of the R instances by tile id only (<= 16 bits -> 2 byte passes).
// A stable sort of a (depth, index)-ordered stream by tile yields exactly (tile, depth, index).
//
// All kernels are wave64 code: per-wave ranking uses 64-bit ballots, block = 4 waves.
#include "olsr_device.h"
#include "olsr_kernels.h"
#include "olsr_loss_device.h"

namespace olsr {

// ------------------------------------------------------------------------------- scans
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = SCAN_CHUNK / SCAN_THREADS;  // 16

__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    u32 o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}

// exclusive prefix of `v` across a 256-thread block; *total receives the block sum
__device__ __forceinline__ u32 block_excl_scan_256(u32 v, u32* total) {
  __shared__ u32 wave_sums[4];
  const int lane = lane_id(), w = threadIdx.x >> 6;
  const u32 incl = wave_incl_scan(v);
  if (lane == 63) wave_sums[w] = incl;
  __syncthreads();
  u32 base = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < w) base += wave_sums[i];
  if (total) *total = wave_sums[0] + wave_sums[1] + wave_sums[2] + wave_sums[3];
  __syncthreads();
  return base + incl - v;
}

// Sum of partials[0 .. b) computed by the whole block (the per-chunk totals are few: one per 4096
// elements), which saves the separate single-block scan launch between reduce and apply.
__device__ __forceinline__ u32 block_prefix_of_partials(const u32* __restrict__ partials, int b) {
  __shared__ u32 red[4];
  u32 s = 0;
  for (int j = threadIdx.x; j < b; j += SCAN_THREADS) s += partials[j];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const u32 tot = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return tot;
}

struct LoadPlain {
  const u32* p;
  __device__ __forceinline__ u32 operator()(int64_t i) const { return p[i]; }
};
// tiles_touched gathered in depth order
struct LoadGather {
  const u32* vals;
  const u32* order;
  __device__ __forceinline__ u32 operator()(int64_t i) const { return vals[order[i]]; }
};

template <class Load>
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(Load ld, int64_t n, u32* partials) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
  u32 s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k)
    if (base + k < n) s += ld(base + k);
  u32 total;
  block_excl_scan_256(s, &total);
  if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

template <class Load, bool INCLUSIVE>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(Load ld, int64_t n, const u32* partials, u32* out) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
  u32 v[SCAN_ITEMS];
  u32 s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    v[k] = (base + k < n) ? ld(base + k) : 0;
    s += v[k];
  }
  u32 run = block_excl_scan_256(s, nullptr) + block_prefix_of_partials(partials, blockIdx.x);
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    if (INCLUSIVE) run += v[k];
    if (base + k < n) out[base + k] = run;
    if (!INCLUSIVE) run += v[k];
  }
}

template <class Load, bool INCLUSIVE>
static void device_scan(Load ld, int64_t n, u32* out, u32* partials, hipStream_t st) {
  if (n <= 0) return;
  const int nb = scan_blocks(n);
  scan_reduce_kernel<Load><<<nb, SCAN_THREADS, 0, st>>>(ld, n, partials);
  scan_apply_kernel<Load, INCLUSIVE><<<nb, SCAN_THREADS, 0, st>>>(ld, n, partials, out);
}

// ------------------------------------------------------------------------------- radix sort
// One LSD pass = histogram kernel + scan of the [digit][block] table + scatter kernel.
// The digit width is chosen per sort (<= 8 bits): a 12-bit tile id is sorted in two 6-bit passes.
constexpr int SORT_THREADS = 256;
constexpr int SORT_ROUNDS = SORT_CHUNK / SORT_THREADS;  // 16 rounds of 64 per wave
// The self-scanning scatter reads the whole [block][digit] table in every block (nblk^2 * NB loads in total): it
// wins while that table is small — measured: 123 blocks x 256 digits -24 us per sort, 489 x 256 +17 us.
constexpr int SELF_SCAN_MAX_TABLE = 40960;              // blocks * digits

__device__ __forceinline__ int64_t bounded_n(int64_t n_host, const int32_t* n_dev) {
  if (n_dev) {
    const int64_t nd = (int64_t)(*n_dev);
    return nd < n_host ? nd : n_host;
  }
  return n_host;
}

// number of keys of block b whose digit is d: table[d * nblk + b] (the layout the device-wide scan walks), or,
// transposed, table[b * (dmask + 1) + d] (the layout the self-scanning scatter reads coalesced)
__global__ __launch_bounds__(SORT_THREADS) void radix_hist_kernel(const u32* __restrict__ keys, int64_t n_host,
                                                                   const int32_t* __restrict__ n_dev, int shift,
                                                                   u32 dmask, int transposed, u32* __restrict__ table) {
  __shared__ u32 hist[256];
  const int64_t n = bounded_n(n_host, n_dev);
  hist[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * SORT_CHUNK;
#pragma unroll 4
  for (int k = 0; k < SORT_ROUNDS; ++k) {
    const int64_t i = base + (int64_t)k * SORT_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&hist[(keys[i] >> shift) & dmask], 1u);
  }
  __syncthreads();
  if (threadIdx.x <= dmask)
    table[transposed ? (size_t)blockIdx.x * (dmask + 1) + threadIdx.x : (size_t)threadIdx.x * gridDim.x + blockIdx.x] =
        hist[threadIdx.x];
}

// Stable scatter.  Wave w of block b owns the contiguous run [b*CHUNK + w*1024, +1024) and walks
// it in 16 rounds of 64 consecutive keys, so the order inside a block is (wave, round, lane) ==
// ascending input index.  Ranks inside a round come from 64-bit ballots (match-any over the digit
// bits).  The block first orders its 4096 pairs by digit in LDS, then writes every digit's run
// with consecutive lanes -> coalesced stores instead of 4-byte scatters.
//
// SELF_SCAN (tables of up to SELF_SCAN_MAX_TABLE entries): the table arrives RAW and transposed ([block][digit]) from
// the histogram kernel and each block derives its own bases — digit totals (column sums), their exclusive
// prefix, and the sum of its column over the earlier blocks — with coalesced, independent loads (all 256
// threads; 256 / NB threads share a digit).  Saves the two launches of the device-wide scan per pass.
template <int DB, bool SELF_SCAN>
__global__ __launch_bounds__(SORT_THREADS) void radix_scatter_kernel(
    const u32* __restrict__ keys_in, const u32* __restrict__ vals_in, int64_t n_host,
    const int32_t* __restrict__ n_dev, int shift, const u32* __restrict__ table, u32* __restrict__ keys_out,
    u32* __restrict__ vals_out) {
  constexpr u32 NB = 1u << DB;
  constexpr u32 DMASK = NB - 1u;
  __shared__ u32 cnt[4][NB];     // per-wave digit counts -> per-wave local starts
  __shared__ u32 gbase[NB];      // global start of this block's run of digit d
  __shared__ u32 dstart[NB + 1]; // local start of digit d inside the block
  __shared__ u32 ex_key[SORT_CHUNK];
  __shared__ u32 ex_val[SORT_CHUNK];
  __shared__ u32 s_part[2][SORT_THREADS];  // SELF_SCAN partial column sums
  const int64_t n = bounded_n(n_host, n_dev);
  const int lane = lane_id(), w = threadIdx.x >> 6;
  const int64_t bbase = (int64_t)blockIdx.x * SORT_CHUNK;
  const int64_t wbase = bbase + (int64_t)w * (SORT_CHUNK / 4);
  u32 key[SORT_ROUNDS], val[SORT_ROUNDS];
  for (u32 i = threadIdx.x; i < 4 * NB; i += SORT_THREADS) (&cnt[0][0])[i] = 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r) {
    const int64_t i = wbase + r * 64 + lane;
    const bool valid = i < n;
    key[r] = valid ? keys_in[i] : 0xFFFFFFFFu;
    val[r] = valid ? (vals_in ? vals_in[i] : (u32)i) : 0u;
    if (valid) atomicAdd(&cnt[w][(key[r] >> shift) & DMASK], 1u);
  }
  __syncthreads();
  {  // thread d owns digit d: block-local exclusive prefix in (digit, wave) order
    const u32 d = threadIdx.x;
    u32 c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    if (d < NB) { c0 = cnt[0][d]; c1 = cnt[1][d]; c2 = cnt[2][d]; c3 = cnt[3][d]; }
    const u32 tot = c0 + c1 + c2 + c3;
    const u32 start = block_excl_scan_256(tot, nullptr);
    u32 gb = 0;
    if constexpr (SELF_SCAN) {
      constexpr u32 PARTS = SORT_THREADS / NB;  // threads per digit
      const u32 dd = threadIdx.x & DMASK, part = threadIdx.x >> DB;
      u32 tot_p = 0, bef_p = 0;
#pragma unroll 8
      for (u32 bb = part; bb < gridDim.x; bb += PARTS) {
        const u32 c = table[(size_t)bb * NB + dd];
        bef_p += (bb < blockIdx.x) ? c : 0u;
        tot_p += c;
      }
      u32 row_total = tot_p, row_before = bef_p;
      if constexpr (PARTS > 1) {
        s_part[0][threadIdx.x] = tot_p;
        s_part[1][threadIdx.x] = bef_p;
        __syncthreads();
        row_total = 0;
        row_before = 0;
        if (d < NB) {
#pragma unroll
          for (u32 q = 0; q < PARTS; ++q) {
            row_total += s_part[0][q * NB + d];
            row_before += s_part[1][q * NB + d];
          }
        }
      }
      gb = block_excl_scan_256(row_total, nullptr) + row_before;
    } else {
      if (d < NB) gb = table[(size_t)d * gridDim.x + blockIdx.x];
    }
    if (d < NB) {
      dstart[d] = start;
      cnt[0][d] = start;
      cnt[1][d] = start + c0;
      cnt[2][d] = start + c0 + c1;
      cnt[3][d] = start + c0 + c1 + c2;
      gbase[d] = gb;
    }
  }
  __syncthreads();
  volatile u32* my = cnt[w];
  const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r) {
    const int64_t i = wbase + r * 64 + lane;
    const bool valid = i < n;
    const u32 d = (key[r] >> shift) & DMASK;
    u64 peers = ballot(valid);
#pragma unroll
    for (int bit = 0; bit < DB; ++bit) {
      const bool one = (d >> bit) & 1u;
      const u64 bm = ballot(one);
      peers &= one ? bm : ~bm;
    }
    if (valid) {
      const u32 rank = (u32)__popcll(peers & lt_mask);
      const u32 start = my[d];
      if (rank == 0) my[d] = start + (u32)__popcll(peers);
      ex_key[start + rank] = key[r];
      ex_val[start + rank] = val[r];
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  const int64_t rem = n - bbase;
  const u32 nvalid = rem >= SORT_CHUNK ? (u32)SORT_CHUNK : (rem > 0 ? (u32)rem : 0u);
#pragma unroll 4
  for (int k = 0; k < SORT_ROUNDS; ++k) {
    const u32 slot = (u32)k * SORT_THREADS + threadIdx.x;
    if (slot < nvalid) {
      const u32 kk = ex_key[slot];
      const u32 d = (kk >> shift) & DMASK;
      const u32 pos = gbase[d] + (slot - dstart[d]);
      keys_out[pos] = kk;
      vals_out[pos] = ex_val[slot];
    }
  }
}

int launch_radix_sort(const SortBuffers& b, int64_t n_host, const int32_t* n_dev, int bits, bool vals_in_identity,
                      hipStream_t st) {
  if (n_host <= 0) return 0;
  const int nblk = sort_blocks(n_host);
  const int passes = (bits + 7) / 8;
  const int db = (bits + passes - 1) / passes;  // digit width, <= 8
  u32 *kin = b.key_a, *kout = b.key_b, *vin = b.val_a, *vout = b.val_b;
  int where = 0;
  for (int p = 0; p < passes; ++p) {
    const int shift = db * p;
    const bool self_scan = (int64_t)nblk * (1 << db) <= SELF_SCAN_MAX_TABLE;
    radix_hist_kernel<<<nblk, SORT_THREADS, 0, st>>>(kin, n_host, n_dev, shift, (1u << db) - 1u, self_scan ? 1 : 0,
                                                     b.table);
    if (!self_scan)
      device_scan<LoadPlain, false>(LoadPlain{b.table}, (int64_t)(1 << db) * nblk, b.table, b.partials, st);
    const u32* vsrc = (p == 0 && vals_in_identity) ? nullptr : vin;
#define OLSR_SCATTER(DBV)                                                                                       \
  case DBV:                                                                                                     \
    if (self_scan)                                                                                              \
      radix_scatter_kernel<DBV, true><<<nblk, SORT_THREADS, 0, st>>>(kin, vsrc, n_host, n_dev, shift, b.table,  \
                                                                     kout, vout);                               \
    else                                                                                                        \
      radix_scatter_kernel<DBV, false><<<nblk, SORT_THREADS, 0, st>>>(kin, vsrc, n_host, n_dev, shift, b.table, \
                                                                      kout, vout);                              \
    break;
    switch (db) {
      OLSR_SCATTER(1) OLSR_SCATTER(2) OLSR_SCATTER(3) OLSR_SCATTER(4)
      OLSR_SCATTER(5) OLSR_SCATTER(6) OLSR_SCATTER(7) OLSR_SCATTER(8)
      default: break;
    }
#undef OLSR_SCATTER
    u32* t = kin; kin = kout; kout = t;
    t = vin; vin = vout; vout = t;
    where ^= 1;
  }
  return where;
}

// ------------------------------------------------------------------------------- emission
// cub::DeviceScan::InclusiveSum + duplicateWithKeys (CR/rasterizer_impl.cu:451, 70-111) in two kernels (below:
// "output-balanced emission").  The depth half of the reference's key is implied by the emission order; only the
// tile id is written.  A first formulation gave every Gaussian to one lane (up to 32 instances) or one wave (more):
// a wave waited for its largest member (5.4 instances on average) and the wave-per-Gaussian kernel paid a latency
// chain per item — 62 us for the 2.7 M instances of the headline frame, 41 us now.
constexpr u32 EMIT_BIG = OLSR_BIG_FOOTPRINT;
constexpr int EMIT_THREADS = EMIT_CHUNK;  // 1024: one list atomic per 1024 Gaussians

// Block totals of a single-pass scan: two 32-bit words per block, {READY | total, READY | inclusive prefix}, zeroed
// before the launch; published and polled with relaxed agent-scope accesses (the data is the flag).
constexpr u32 LB_READY = 0x80000000u;
__device__ __forceinline__ u32 lb_load(const u32* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lb_store(u32* p, u32 v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u32 lb_wait(const u32* p, int spin_limit, int32_t* sync_error) {
  u32 v = lb_load(p);
  // (bounded: a predecessor publishes within microseconds; a corrupted state buffer must not hang the GPU)
  for (int spin = 0; !(v & LB_READY) && spin < spin_limit; ++spin) {
    __builtin_amdgcn_s_sleep(1);
    v = lb_load(p);
  }
  // it never arrived: the prefix is garbage — mark the frame (the backward then writes zero gradients and reports it)
  if (!(v & LB_READY)) atomicOr(sync_error, 1);
  return v & ~LB_READY;
}
// Sum of the totals of blocks [0, b), for ALL threads of block b (which publishes `total` here).  Blocks take their
// index from a ticket, so every predecessor has started and publishes without waiting for anyone: the look-back is
// ONE batch — every thread fetches the total of one predecessor (a dependent trip to the fabric costs ~2 us, a chain
// of 64-wide windows costs one per window).  Only beyond THREADS predecessors does a block wait for an inclusive
// prefix (that of the last block of the previous group of THREADS).
template <int THREADS>
__device__ __forceinline__ u32 lb_block_exclusive(u32* status, u32 b, u32 total, u32* s_red /* [THREADS / 64 + 1] */,
                                                  int spin_limit, int32_t* sync_error) {
  const u32 tid = threadIdx.x;
  const u32 first = b & ~(u32)(THREADS - 1);
  if (tid == 0) lb_store(&status[2 * b], LB_READY | total);
  u32 v = 0;
  if (first + tid < b) v = lb_wait(&status[2 * (first + tid)], spin_limit, sync_error);
  if (tid == THREADS - 1 && first > 0)  // (first + tid >= b for this thread)
    v += lb_wait(&status[2 * (first - 1) + 1], spin_limit, sync_error);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  if ((tid & 63) == 0) s_red[tid >> 6] = v;
  __syncthreads();
  u32 excl = 0;
#pragma unroll
  for (int i = 0; i < THREADS / 64; ++i) excl += s_red[i];
  if (tid == 0) lb_store(&status[2 * b + 1], LB_READY | (excl + total));
  __syncthreads();  // (s_red may be reused)
  return excl;
}

// ---- output-balanced emission ------------------------------------------------------------------------------
//   emit_offsets_kernel   one thread per depth rank: first instance of every rank (rank_off), of every block of 1024
//                         ranks (blk_base), inst_start, the big / mid lists for the backward's row sums, and the
//                         binning buffer's synchronisation words.  The block totals come from the depth sort's last
//                         pass, so no block waits for another one;
//   emit_balanced_kernel  block b writes instances [b * EB_OUT, (b + 1) * EB_OUT): it finds the depth ranks that own
//                         them (two counting searches over the monotone blk_base / rank_off), expands Gaussians to
//                         tile rows (a table in LDS), evaluates one row span per thread, and writes keys / inst_gid
//                         in output order — a thread its own short row, a wave together a long one.  Every block has
//                         the same amount of output whatever the footprint mix (near splats cover hundreds to
//                         thousands of tiles and cluster at the front of the depth order); a Gaussian that straddles
//                         two blocks has its row spans evaluated by both.
#ifndef OLSR_EB_OUT
#define OLSR_EB_OUT 1024
#endif
#ifndef OLSR_EB_T
#define OLSR_EB_T 256
#endif
constexpr int EB_T = OLSR_EB_T;
constexpr int EB_OUT = OLSR_EB_OUT;
constexpr int EB_ROWCAP = 16 * EB_T;   // rows per fill of the row -> Gaussian table
static_assert((EB_OUT & (EB_OUT - 1)) == 0 && EB_OUT >= 4, "the emission window is a power of two");
static_assert(EB_T <= 256 && EB_T % 64 == 0, "the row -> Gaussian table holds thread indices in one byte");
constexpr u32 EB_SHORT_ROW = 8;   // rows up to this many tiles are written by their own thread

// (4-wave workgroups, four consecutive depth ranks per thread: a block still covers the EMIT_CHUNK ranks whose total the
//  depth sort accumulated, but it fits beside the resident blocks of a compositing kernel of another frame in flight —
//  16-wave workgroups only run in the gaps between composites)
constexpr int EO_T = 256;
constexpr int EO_PER = EMIT_CHUNK / EO_T;  // 4
static_assert(EO_PER == 4, "one 16-byte load of the depth order per thread");

// base slot of this thread's `cnt` list entries: one aggregated atomic per block (the list order influences no result)
__device__ __forceinline__ u32 block_list_base(u32 cnt, int32_t* counter, u32* s_cnt /* [EO_T / 64] */, u32* s_base) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  u32 incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u32 o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  __syncthreads();  // (scratch may still be read from the previous use)
  if (lane == 63) s_cnt[w] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 tot = 0;
    for (int i = 0; i < EO_T / 64; ++i) {
      const u32 c = s_cnt[i];
      s_cnt[i] = tot;
      tot += c;
    }
    *s_base = tot ? (u32)atomicAdd(counter, (int)tot) : 0u;
  }
  __syncthreads();
  return *s_base + s_cnt[w] + incl - cnt;
}

__global__ __launch_bounds__(EO_T) void emit_offsets_kernel(
    int P, const u32* __restrict__ order, const u32* __restrict__ inst_count, int32_t* __restrict__ counters,
    const u64* __restrict__ block_totals_sort, uint4* __restrict__ bin_sync, int bin_sync_quads,
    u32* __restrict__ c_off, u32* __restrict__ c_gid, u32* __restrict__ win_start, u32* __restrict__ inst_start,
    uint4* __restrict__ big_list, u32 eb_shift, const u64* __restrict__ alt_totals, const u32* __restrict__ alt_unless,
    const int32_t* __restrict__ n_order_dev) {
  __shared__ u64 s_wsum[EO_T / 64 + 1];
  __shared__ u32 s_lsum[EO_T / 64 + 1];
  __shared__ u32 s_lbase;
  // (the carried depth order, k_order_carry.hip: a repaired order comes with its own block totals; the radix passes' totals
  //  count when the repair missed.  Uniform.)
  const u64* __restrict__ block_totals =
      (alt_totals != nullptr && __hip_atomic_load(alt_unless, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) ? alt_totals
                                                                                                                 : block_totals_sort;
  const u32 eb_out = 1u << eb_shift;  // output window of an emission block (a power of two)
  // the words the tile sort and the row compaction synchronise through live in the binning buffer, which exists only
  // from here on (the drop-in entry allocates it after the instance count is known)
  for (int q = (int)(blockIdx.x * EO_T + threadIdx.x); q < bin_sync_quads; q += (int)(gridDim.x * EO_T))
    bin_sync[q] = make_uint4(0u, 0u, 0u, 0u);
  if (counters[2] != 0) return;  // more instances than the caller's capacity: nothing is emitted (uniform)
  if (counters[8] != 0) return;  // the depth sort lost a predecessor's counts: its order is garbage, index nothing with it
  const u32 b = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r0 = (int)(b * EMIT_CHUNK + threadIdx.x * EO_PER);  // this thread's four consecutive depth ranks
  // (P_true sizes the Gaussian-indexed arrays; the ORDER holds n_order entries — all P, or only the Gaussians that emit when
  //  the depth sort compacted its input, k_sort.hip.  Blocks beyond them have nothing to do: uniform.)
  const int P_true = P;
  if (n_order_dev != nullptr) P = min(P, *n_order_dev);
  if ((int)(b * EMIT_CHUNK) >= P && !(b == 0 && P == 0)) return;
  u32 g[EO_PER], n[EO_PER];
  if (r0 + EO_PER <= P) {
    const uint4 q = *reinterpret_cast<const uint4*>(order + r0);  // (the order array is 256-byte aligned)
    g[0] = q.x; g[1] = q.y; g[2] = q.z; g[3] = q.w;
  } else {
#pragma unroll
    for (int k = 0; k < EO_PER; ++k) g[k] = (r0 + k < P) ? order[r0 + k] : 0u;
  }
  // (instances | emitting Gaussians << 40, like the block totals: ONE scan yields the first instance of every rank and its
  //  place among the ranks that emit anything)
  u64 mine = 0;
#pragma unroll
  for (int k = 0; k < EO_PER; ++k) {
    n[k] = (r0 + k < P) ? inst_count[g[k]] : 0u;  // 0 when culled
    mine += emit_total_pack(n[k]);
  }
  u64 incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u64 o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) s_wsum[w] = incl;
  __syncthreads();
  u64 wbase = 0;
#pragma unroll
  for (int i = 0; i < EO_T / 64; ++i) wbase += (i < w) ? s_wsum[i] : 0ull;
  __syncthreads();
  // first instance (and first emitting rank) of this block = the totals of all earlier blocks (left behind by the depth
  // sort's last pass, or by the repair of a carried order)
  u64 pre = 0;
  for (u32 j = threadIdx.x; j < b; j += EO_T) pre += block_totals[j];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) pre += __shfl_xor(pre, m);
  if (lane == 0) s_wsum[w] = pre;
  __syncthreads();
  u64 base = 0;
#pragma unroll
  for (int i = 0; i < EO_T / 64; ++i) base += s_wsum[i];
  const u64 first = base + wbase + incl - mine;
  constexpr u64 LOW = (1ull << EMIT_TOTAL_SHIFT) - 1ull;
  u32 off[EO_PER];
  u32 run = (u32)(first & LOW);
  u32 ci = (u32)(first >> EMIT_TOTAL_SHIFT);  // this thread's first rank among the emitting ones
  u32 nbig = 0, nmid = 0;
#pragma unroll
  for (int k = 0; k < EO_PER; ++k) {
    off[k] = run;
    run += n[k];
    nbig += (n[k] > EMIT_BIG) ? 1u : 0u;
    nmid += (n[k] > OLSR_MID_FOOTPRINT && n[k] <= EMIT_BIG) ? 1u : 0u;
  }
  // The emission walks the ranks that EMIT: {first instance, Gaussian} of the j-th of them, and per output window the one
  // that owns its first instance.  (Round 6: with a sort key for every Gaussian the silent ones stand between the emitting
  // ones instead of behind them — a view that sees a fifth of the map — and emit_balanced_kernel's batches of 256 ranks
  // were four fifths empty: 22 -> 77 us on the room map.  Compacted, a batch is 256 emitting Gaussians whatever the view sees.)
#pragma unroll
  for (int k = 0; k < EO_PER; ++k) {
    if (n[k] > 0) {
      inst_start[g[k]] = off[k];
      c_off[ci] = off[k];
      c_gid[ci] = g[k];
      // this rank owns the first instance of every output window [j * EB_OUT, ...) that starts inside its run
      for (u32 j = (off[k] + eb_out - 1u) >> eb_shift; (j << eb_shift) < off[k] + n[k]; ++j)
        win_start[j] = ci;
      ++ci;
    }
  }
  // (the thread that holds the last rank has seen every rank: the number of emitting Gaussians)
  if ((r0 <= P - 1 && P - 1 < r0 + EO_PER) || (P == 0 && b == 0 && threadIdx.x == 0)) counters[10] = (int32_t)ci;
  // the two work lists of the backward's row sums: big footprints from the front, medium ones from the back
  u32 slot = block_list_base(nbig, &counters[5], s_lsum, &s_lbase);
#pragma unroll
  for (int k = 0; k < EO_PER; ++k)
    if (n[k] > EMIT_BIG) big_list[slot++] = make_uint4(g[k], off[k], n[k], 0u);
  u32 mslot = block_list_base(nmid, &counters[4], s_lsum, &s_lbase);
#pragma unroll
  for (int k = 0; k < EO_PER; ++k)
    if (n[k] > OLSR_MID_FOOTPRINT && n[k] <= EMIT_BIG) big_list[(u32)P_true - 1u - (mslot++)] = make_uint4(g[k], off[k], n[k], 0u);
}

// exclusive scan of v over the EB_T threads of the block; *total = the sum (two barriers; s_w: [EB_T / 64])
__device__ __forceinline__ u32 eb_scan(u32 v, u32* s_w, u32* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  u32 incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u32 o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  __syncthreads();  // (s_w may still be read from the previous use)
  if (lane == 63) s_w[w] = incl;
  __syncthreads();
  u32 wb = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < EB_T / 64; ++i) {
    const u32 c = s_w[i];
    wb += (i < w) ? c : 0u;
    tot += c;
  }
  *total = tot;
  return wb + incl - v;
}

// what a row's thread needs about its Gaussian: the ellipse of cull_setup, the rect's columns, where the rows start
struct EbGauss {
  float4 e0;  // px py b inv_a
  float4 e1;  // A det_lo xstar ystar
  uint4 m;    // exact, x0 | x1 << 16, first row, first instance
};

template <int TILE>
__global__ __launch_bounds__(EB_T) void emit_balanced_kernel(
    const u32* __restrict__ order, const u32* __restrict__ rank_off, const u32* __restrict__ win_start,
    const float4* __restrict__ emit_rec, int ellipse, int W, int H, int gx, const int32_t* __restrict__ counters,
    u32* __restrict__ keys, u32* __restrict__ inst_gid, u32 eb_out) {
  // (order / rank_off: the Gaussian and the first instance of the j-th EMITTING depth rank, j < P = counters[10] —
  //  emit_offsets_kernel's compacted list; "rank" below means a position in that list)
  __shared__ EbGauss s_gs[EB_T];
  __shared__ u32 s_g[EB_T], s_first[EB_T], s_rowoff[EB_T];
  __shared__ uint8_t s_owner[EB_ROWCAP];
  __shared__ u32 s_w[EB_T / 64];
  __shared__ u32 s_flag;
  if (counters[2] != 0 || counters[8] != 0) return;
  const u32 R = (u32)counters[1];
  const int P = counters[10];
  const u32 o0 = blockIdx.x * eb_out;
  if (o0 >= R) return;
  const u32 o1 = min(o0 + eb_out, R);
  const int tid = threadIdx.x, lane = tid & 63;

  // the rank that owns instance o0 (left behind by emit_offsets_kernel)
  const int r_lo = (int)win_start[blockIdx.x];
  auto load_rank = [&](int r, u32& off, u32& end, u32& g) {
    off = 0; end = 0; g = 0;
    if (r < P) {
      off = rank_off[r];
      end = (r + 1 < P) ? rank_off[r + 1] : R;
      g = order[r];
    }
  };
  u32 off, end, g;
  load_rank(r_lo + tid, off, end, g);

  for (int rb0 = r_lo;; rb0 += EB_T) {
    // ---- one batch of EB_T depth ranks
    const int r = rb0 + tid;
    const bool use = end > off && off < o1 && end > o0;
    u32 nrows = 0;
    if (use) {
      const float4 r0 = emit_rec[2 * (size_t)g], r1 = emit_rec[2 * (size_t)g + 1];
      const u32 yw = __float_as_uint(r1.z), xw = __float_as_uint(r1.w);
      nrows = yw >> 16;
      EbGauss q;
      q.m = make_uint4(0u, xw, yw & 0xFFFFu, off);
      q.e0 = make_float4(0.f, 0.f, 0.f, 0.f);
      q.e1 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ellipse) {
        const CullEllipse e = cull_setup(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, 0);
        q.e0 = make_float4(e.px, e.py, e.b, e.inv_a);
        q.e1 = make_float4(e.A, e.det_lo, e.xstar, e.ystar);
        q.m.x = e.exact ? 1u : 0u;
      }
      s_gs[tid] = q;
    }
    s_g[tid] = g;
    if (tid == EB_T - 1) s_flag = (r < P && end < o1) ? 1u : 0u;  // ranks beyond this batch still own instances
    // the next batch's ranks are fetched while this one is expanded
    u32 off_n, end_n, g_n;
    load_rank(r + EB_T, off_n, end_n, g_n);
    u32 total_rows;
    const u32 rowoff = eb_scan(nrows, s_w, &total_rows);
    s_rowoff[tid] = rowoff;
    __syncthreads();
    const bool more = s_flag != 0u;

    u32 carry = 0;
    for (u32 sbase = 0; sbase < total_rows; sbase += EB_ROWCAP) {
      // row -> Gaussian table for rows [sbase, sbase + EB_ROWCAP) of the batch
      {
        const u32 k0 = max(rowoff, sbase), k1 = min(rowoff + nrows, sbase + (u32)EB_ROWCAP);
        for (u32 k = k0; k < k1; ++k) s_owner[k - sbase] = (uint8_t)tid;
      }
      __syncthreads();
      const u32 send = min(total_rows, sbase + (u32)EB_ROWCAP);
      for (u32 rbase = sbase; rbase < send; rbase += EB_T) {
        // ---- one row per thread
        const u32 idx = rbase + (u32)tid;
        const bool live = idx < send;
        int owner = 0;
        u32 cnt = 0, key0 = 0;
        if (live) {
          owner = (int)s_owner[idx - sbase];
          const EbGauss q = s_gs[owner];
          const int y = (int)q.m.z + (int)(idx - s_rowoff[owner]);
          int xa = (int)(q.m.y & 0xFFFFu), xb = (int)(q.m.y >> 16);
          if (ellipse) {
            // exact binning: the same row spans preprocess counted (same function, same inputs)
            CullEllipse e;
            e.px = q.e0.x; e.py = q.e0.y; e.b = q.e0.z; e.inv_a = q.e0.w;
            e.A = q.e1.x; e.det_lo = q.e1.y; e.xstar = q.e1.z; e.ystar = q.e1.w;
            e.ymax = 0.f;
            e.exact = q.m.x != 0u;
            cull_row_span<TILE>(e, xa, xb, y, W, H, xa, xb);
          }
          cnt = (u32)(xb - xa);
          key0 = (u32)(y * gx + xa);
        }
        u32 chunk_total;
        const u32 S = carry + eb_scan(cnt, s_w, &chunk_total);
        carry += chunk_total;
        // a row's first instance = its Gaussian's first instance + the instances of the Gaussian's earlier rows
        if (live && idx == s_rowoff[owner]) s_first[owner] = S;
        __syncthreads();
        u32 out = 0, gid = 0;
        if (live) {
          out = s_gs[owner].m.w + (S - s_first[owner]);
          gid = s_g[owner];
        }
        // the part of the row inside this block's window
        const u32 jlo = max(out, o0), jhi = min(out + cnt, o1);
        const bool inside = live && jlo < jhi;
        if (inside && cnt <= EB_SHORT_ROW) {
          for (u32 j = jlo; j < jhi; ++j) {
            keys[j] = key0 + (j - out);
            inst_gid[j] = gid;
          }
        }
        u64 longrows = ballot(inside && cnt > EB_SHORT_ROW);
        while (longrows) {  // the wave writes a long row together
          const int src = (int)__builtin_ctzll(longrows);
          longrows &= longrows - 1;
          const u32 lo_s = (u32)__shfl((int)jlo, src), hi_s = (u32)__shfl((int)jhi, src);
          const u32 out_s = (u32)__shfl((int)out, src), key_s = (u32)__shfl((int)key0, src);
          const u32 gid_s = (u32)__shfl((int)gid, src);
          for (u32 j = lo_s + (u32)lane; j < hi_s; j += 64) {
            keys[j] = key_s + (j - out_s);
            inst_gid[j] = gid_s;
          }
        }
      }
      __syncthreads();  // (s_owner is refilled)
    }
    if (!more) break;
    off = off_n;
    end = end_n;
    g = g_n;
    __syncthreads();  // (s_gs / s_g / s_rowoff are rewritten by the next batch)
  }
}

// per-emission-block instance totals for a depth order that did not come from the fused sort (multi-kernel passes)
__global__ __launch_bounds__(EMIT_THREADS) void emit_totals_kernel(int P, const u32* __restrict__ order,
                                                                   const u32* __restrict__ inst_count,
                                                                   u64* __restrict__ totals) {
  __shared__ u64 s_w[EMIT_THREADS / 64];
  const int r = (int)(blockIdx.x * EMIT_THREADS + threadIdx.x);
  u64 n = (r < P) ? emit_total_pack(inst_count[order[r]]) : 0ull;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) n += __shfl_xor(n, m);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 t = 0;
    for (int i = 0; i < EMIT_THREADS / 64; ++i) t += s_w[i];
    totals[blockIdx.x] = t;
  }
}
void launch_emit_totals(const uint32_t* order, int P, const uint32_t* inst_count, uint32_t* emit_totals, hipStream_t st) {
  if (P <= 0) return;
  emit_totals_kernel<<<(P + EMIT_THREADS - 1) / EMIT_THREADS, EMIT_THREADS, 0, st>>>(P, order, inst_count,
                                                                                     reinterpret_cast<u64*>(emit_totals));
}

template <int TILE>
static void launch_emit_t(const olsr_scene& s, const FrameDims& d, const GeometryState& g, const BinningState& b,
                          int64_t bin_sync_words, int64_t n_host, const u32* order, const u32* alt_totals,
                          const u32* alt_unless, const int32_t* n_order_dev, hipStream_t st) {
  const int nb = (s.P + EMIT_THREADS - 1) / EMIT_THREADS;
  const int ellipse = (s.binning == OLSR_BINNING_ELLIPSE);
  const u32* totals = g.emit_status;  // per-block instance totals, accumulated by the depth sort's last pass
  uint4* bsync = reinterpret_cast<uint4*>(b.sync_words);
  const int quads = (int)((bin_sync_words + 3) / 4);
  u32* rank_off = g.key_b;   // the depth keys are dead once the order exists: first instance of the j-th emitting rank ...
  u32* rank_gid = g.val_b;   // ... and its Gaussian (the sort's / the repair's second value buffer is dead as well)
  u32* win_start = b.key_b;  // the tile sort's second key buffer is not in use yet
  // Output window of an emission block.  With per-tile depth cut-offs most depth ranks emit nothing: a window's instances
  // come from four times as many ranks, walked in serial batches of 256 — a quarter of the window keeps the per-block
  // chain where it was (emit_balanced 38.9 -> see profiles/r4_experiments.json).
  const bool cut_mode = ellipse && s.tile_depth_cut != nullptr;
  const int eb_out = cut_mode ? EB_OUT / 4 : EB_OUT;
  emit_offsets_kernel<<<nb, EO_T, 0, st>>>(s.P, order, g.tiles_touched, g.counters, reinterpret_cast<const u64*>(totals), bsync,
                                           quads, rank_off, rank_gid, win_start, g.inst_start, g.big_list,
                                           (u32)__builtin_ctz((unsigned)eb_out), reinterpret_cast<const u64*>(alt_totals),
                                           alt_unless, n_order_dev);
  const int64_t nblk = (n_host + eb_out - 1) / eb_out;
  if (nblk > 0)
    emit_balanced_kernel<TILE><<<(int)nblk, EB_T, 0, st>>>(rank_gid, rank_off, win_start, g.emit_rec,
                                                           ellipse, d.W, d.H, d.gx, g.counters, b.key_a, b.inst_gid,
                                                           (u32)eb_out);
}

void launch_emit(const olsr_scene& s, const FrameDims& d, const GeometryState& g, const BinningState& b,
                 int64_t bin_sync_words, int64_t n_host, const uint32_t* order, const uint32_t* alt_totals,
                 const uint32_t* alt_unless, const int32_t* n_order_dev, hipStream_t st) {
  if (s.P <= 0) return;
  if (d.tile == 15) launch_emit_t<15>(s, d, g, b, bin_sync_words, n_host, order, alt_totals, alt_unless, n_order_dev, st);
  else launch_emit_t<16>(s, d, g, b, bin_sync_words, n_host, order, alt_totals, alt_unless, n_order_dev, st);
}

// ------------------------------------------------------------------------------- row compaction
// Exclusive scan of popcount(flags[u]) — how many slot rows each instance owns — in ONE kernel: a 1024-thread block
// covers 16384 instances (a thread fetches its 16 flag bytes with one 16-byte load: the flag array is 256-byte aligned
// and padded to a multiple of 16), publishes its total and gets the sum of the earlier blocks by decoupled look-back
// (lb_exclusive_prefix above).  The status words and the ticket are zeroed by the forward's emission; because a
// backward may be repeated on the same forward, the last block to finish zeroes them again for the next launch.
// rows per instance = popcount((flag >> shift) & mask): shift 0 / mask 15 = one row per forward slot,
// shift 4 / mask 3 = one row per packed survivor wave (reference mode, 15x15 tiles)
__device__ __forceinline__ void load_popc16(const uint8_t* __restrict__ flags, int64_t base, int64_t n, int shift,
                                            u32 mask, u32 (&v)[16]) {
  uint4 q = make_uint4(0u, 0u, 0u, 0u);
  if (base + 16 <= n) {
    q = *reinterpret_cast<const uint4*>(flags + base);
  } else {
    u32 w[4] = {0u, 0u, 0u, 0u};
    for (int k = 0; k < 16; ++k)
      if (base + k < n) w[k >> 2] |= (u32)flags[base + k] << (8 * (k & 3));
    q = make_uint4(w[0], w[1], w[2], w[3]);
  }
  const u32 w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = (u32)__popc((w4[k >> 2] >> (8 * (k & 3) + shift)) & mask);
}

constexpr int ROWS_THREADS = OLSR_ROWS_THREADS;
static_assert(ROWS_CHUNK == ROWS_THREADS * 16, "one 16-byte load of flags per thread");

struct RowCompactionArgs {
  const uint8_t* flags;
  int64_t n_host;
  const int32_t* n_dev;
  int shift;
  u32 mask;
  u32* rowbase;
  u32* status;
  u32* sync;  // [0] ticket, [1] finished blocks
  long long row_capacity;
  int32_t* counters;
  int32_t* status_dev;
  int spin_limit;
};

// one block's share; nblocks = the blocks of the launch that run this body (the whole grid of row_compaction_kernel, the
// row part of forward_tail_kernel)
__device__ __forceinline__ void row_compaction_block(const RowCompactionArgs& ra, const u32 nblocks) {
  const uint8_t* __restrict__ flags = ra.flags;
  const int64_t n_host = ra.n_host;
  const int32_t* __restrict__ n_dev = ra.n_dev;
  const int shift = ra.shift;
  const u32 mask = ra.mask;
  u32* __restrict__ rowbase = ra.rowbase;
  u32* status = ra.status;
  u32* sync = ra.sync;
  const long long row_capacity = ra.row_capacity;
  int32_t* counters = ra.counters;
  int32_t* __restrict__ status_dev = ra.status_dev;
  const int spin_limit = ra.spin_limit;
  __shared__ u32 s_bid;
  __shared__ u32 s_wsum[ROWS_THREADS / 64 + 1];
  __shared__ u32 s_last;
  // block index = a ticket, always: a block waits only for blocks that have started (deadlock-free under any dispatch
  // order and any co-tenancy with other frames' kernels; see sort_pass_kernel)
  if (threadIdx.x == 0) s_bid = atomicAdd(&sync[0], 1u);
  __syncthreads();
  const u32 b = s_bid;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t n = bounded_n(n_host, n_dev);  // flags beyond the instances of this frame are stale
  const int64_t bbase = (int64_t)b * ROWS_CHUNK;
  // (a ticket beyond the grid: the ticket word was not the zero the emission left — corrupted from outside)
  if (b >= nblocks && threadIdx.x == 0) atomicOr(&counters[8], 1);
  if (b < nblocks && bbase <= n) {  // (the block that holds index n writes the total; later blocks have nothing to do)
    const int64_t base = bbase + (int64_t)threadIdx.x * 16;
    u32 v[16];
    load_popc16(flags, base, n, shift, mask, v);
    u32 sum = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += v[k];
    u32 incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const u32 o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    if (lane == 63) s_wsum[w] = incl;
    __syncthreads();
    u32 wbase = 0, total = 0;
#pragma unroll
    for (int i = 0; i < ROWS_THREADS / 64; ++i) {
      const u32 c = s_wsum[i];
      wbase += (i < w) ? c : 0u;
      total += c;
    }
    __syncthreads();
    const u32 pre = lb_block_exclusive<ROWS_THREADS>(status, b, total, s_wsum, spin_limit, &counters[8]);
    u32 run = pre + wbase + incl - sum;
    // (static indices only: a dynamically indexed private array would be promoted to 64 KB of LDS)
    u32 L = 0;
    const int at_n = (base <= n && n < base + 16) ? (int)(n - base) : -1;
    uint4 q4[4];
    u32* o = reinterpret_cast<u32*>(q4);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      o[k] = run;
      L = (k == at_n) ? run : L;
      run += v[k];
    }
    if (base + 16 <= n) {
      uint4* dst = reinterpret_cast<uint4*>(rowbase + base);
      dst[0] = q4[0];
      dst[1] = q4[1];
      dst[2] = q4[2];
      dst[3] = q4[3];
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (base + k < n) rowbase[base + k] = o[k];
    }
    if (at_n >= 0) {  // the thread whose span holds index n: flags at and beyond n count as zero, so L is the total
      // a forward that overflowed its instance capacity (counters[2]) emitted nothing: inst_start / rowbase / rows
      // of this frame do not exist, so the backward must not read them — report it like a row-capacity overflow
      // (every later kernel then writes zero gradients)
      const int32_t ov = ((long long)L > row_capacity || counters[2] != 0) ? 1 : 0;
      rowbase[n] = L;
      counters[6] = (int32_t)L;
      // (a frame whose depth cut-offs hid contributions — counters[9], OLSR_STATUS_CUT_MISS — hands out no gradients either;
      //  the backward's last kernel reports it as status 3)
      counters[7] = (ov != 0 || counters[9] != 0) ? 1 : 0;
      counters[11] = rows_stamp_of(row_capacity);  // "compacted for a scratch of this many rows" (olsr_device.h: frame_unusable)
      if (status_dev) {
        status_dev[0] = (int32_t)L;
        status_dev[1] = ov;
      }
    }
  }
  // self-reset for a repeated backward on the same forward: whoever finishes last has seen every block done
  __syncthreads();
  if (threadIdx.x == 0) s_last = (atomicAdd(&sync[1], 1u) == nblocks - 1u) ? 1u : 0u;
  __syncthreads();
  if (s_last) {
    for (u32 i = threadIdx.x; i < 2 * nblocks; i += ROWS_THREADS) lb_store(&status[i], 0u);
    if (threadIdx.x == 0) {
      __hip_atomic_store(&sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ __launch_bounds__(ROWS_THREADS) void row_compaction_kernel(const RowCompactionArgs ra) {
  row_compaction_block(ra, gridDim.x);
}

void launch_row_compaction(const uint8_t* flags, int64_t n_host, const int32_t* n_dev, bool packed_ref15,
                           uint32_t* rowbase, uint32_t* row_status, uint32_t* sync, int64_t row_capacity,
                           int32_t* counters, int32_t* status_dev, hipStream_t st) {
  const int shift = packed_ref15 ? 4 : 0;
  const u32 mask = packed_ref15 ? 3u : 15u;
  if (n_host < 0) n_host = 0;
  const int nb = (int)((n_host + 1 + ROWS_CHUNK - 1) / ROWS_CHUNK);  // index n itself belongs to a block
  const RowCompactionArgs ra{flags, n_host, n_dev, shift, mask, rowbase, row_status, sync, (long long)row_capacity, counters,
                             status_dev, sort_knobs().spin_limit.load(std::memory_order_relaxed)};
  row_compaction_kernel<<<nb, ROWS_THREADS, 0, st>>>(ra);
}

// ------------------------------------------------------------------------------- ranges
// identifyTileRanges (CR/rasterizer_impl.cu:116-138); ranges were zeroed by finalize_counts_kernel.
// Also clears the liveness flag of every instance of this frame (the forward composite only writes
// the flags of instances that blend something).
__global__ __launch_bounds__(256) void tile_ranges_kernel(const u32* __restrict__ keys, int64_t n_host,
                                                          const int32_t* __restrict__ n_dev, u32* __restrict__ ranges,
                                                          uint8_t* __restrict__ flags) {
  const int64_t n = bounded_n(n_host, n_dev);
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  flags[idx] = 0;
  const u32 cur = keys[idx];
  if (idx == 0)
    ranges[2 * cur] = 0;
  else {
    const u32 prev = keys[idx - 1];
    if (cur != prev) {
      ranges[2 * prev + 1] = (u32)idx;
      ranges[2 * cur] = (u32)idx;
    }
  }
  if (idx == n - 1) ranges[2 * cur + 1] = (u32)n;
}

void launch_tile_ranges(const uint32_t* sorted_keys, int64_t n_host, const int32_t* n_dev, uint32_t* ranges,
                        uint8_t* flags, hipStream_t st) {
  if (n_host <= 0) return;
  const int nb = (int)((n_host + 255) / 256);
  tile_ranges_kernel<<<nb, 256, 0, st>>>(sorted_keys, n_host, n_dev, ranges, flags);
}

// ------------------------------------------------------------------------------- tile order
// Longest-processing-time-first inside each XCD chunk.  xcd_remap gives XCD x the contiguous
// tiles [start_x, start_x + len_x); workgroup b = 8k + x of the backward composite takes the k-th
// heaviest tile of chunk x, so stragglers start early while neighbouring tiles still share an L2.
// Rank sort (len <= a few thousand): rank = #tiles heavier, ties by index -> a permutation.
// The frame's gradient-row counts = the sums of the two tile_work planes.  Every participating block adds its share
// (a, b: this thread's partial sums) to live_rows[0..1]; the block that finishes last sends the totals to the host like
// the instance count travels: plain system-scope stores into mapped host memory, the sequence number last.
__device__ __forceinline__ void sum_and_post_live_rows(u32 a, u32 b, u32 nblocks, u32* live_rows, int32_t* mailbox,
                                                       int32_t seq) {
  __shared__ u32 s_sum[2];
  if (threadIdx.x == 0) s_sum[0] = s_sum[1] = 0;
  __syncthreads();
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    a += __shfl_xor(a, m);
    b += __shfl_xor(b, m);
  }
  if ((threadIdx.x & 63) == 0) {
    if (a) atomicAdd(&s_sum[0], a);
    if (b) atomicAdd(&s_sum[1], b);
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  if (s_sum[0]) atomicAdd(&live_rows[0], s_sum[0]);
  if (s_sum[1]) atomicAdd(&live_rows[1], s_sum[1]);
  __threadfence();
  if (atomicAdd(&live_rows[3], 1u) != nblocks - 1) return;
  __threadfence();
  const u32 t0 = __hip_atomic_load(&live_rows[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const u32 t1 = __hip_atomic_load(&live_rows[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&mailbox[0], (int32_t)t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&mailbox[1], (int32_t)t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&mailbox[2], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Per-tile depth cut-offs (include/olsr.h): the forward composite left each tile's own cut-off in the second half of the
// caller's array; the cut-off the NEXT frame applies to a tile is the largest of its 3 x 3 neighbourhood — what a tile needs
// one frame later is, after a small camera step, what it or a neighbour needed now (a depth edge that moves into the tile).
struct CutDilate {
  float* cut;  // [2 * ntiles]: applied (written here) | raw (read here), or null
  int gx, gy;
};
__device__ __forceinline__ void dilate_depth_cuts(const CutDilate& cd, int thread, int nthreads) {
  if (cd.cut == nullptr) return;
  const int ntiles = cd.gx * cd.gy;
  const float* raw = cd.cut + ntiles;
  for (int t = thread; t < ntiles; t += nthreads) {
    const int tx = t % cd.gx, ty = t / cd.gx;
    float m = raw[t];
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int x = tx + dx, y = ty + dy;
        if (x >= 0 && x < cd.gx && y >= 0 && y < cd.gy) m = fmaxf(m, raw[y * cd.gx + x]);
      }
    cd.cut[t] = m;
  }
}

struct TileOrderArgs {
  const u32* work;
  u32* order;
  u32* order_copy;
  int ntiles;
  u32* live_rows;
  int32_t* mailbox;
  int32_t seq;
  const int32_t* counters;
  int32_t* num_rendered_dev;
  int32_t* sticky;
  const u32* hint_slot;
  CutDilate cd;
  LossFinalArgs lfa;
};

// block (bx, by) of a logical (8, ny) grid of 256-thread blocks (tile_order_kernel's own grid, or the first 8 ny blocks of
// forward_tail_kernel, whose threads beyond 256 have left)
__device__ __forceinline__ void tile_order_block(const TileOrderArgs& ta, const int bx, const int by, const int ny) {
  const u32* __restrict__ work = ta.work;
  u32* __restrict__ order = ta.order;
  u32* __restrict__ order_copy = ta.order_copy;
  const int ntiles = ta.ntiles;
  u32* __restrict__ live_rows = ta.live_rows;
  int32_t* mailbox = ta.mailbox;
  const int32_t seq = ta.seq;
  const int32_t* __restrict__ counters = ta.counters;
  int32_t* __restrict__ num_rendered_dev = ta.num_rendered_dev;
  int32_t* sticky = ta.sticky;
  const u32* __restrict__ hint_slot = ta.hint_slot;
  const CutDilate& cd = ta.cd;
  const LossFinalArgs& lfa = ta.lfa;
  extern __shared__ __attribute__((aligned(16))) u32 s_work[];  // the chunk's weights, padded to a multiple of 64
  // the loss of a forward with the fused epilogue: its per-tile partial sums are reduced here, by the grid's last block
  // (uniform branch; the block then does its share of the tile order like every other)
  if (lfa.partials != nullptr && bx == 7 && by == ny - 1) {
    __shared__ double s_red[4][LOSS_SUMS];
    loss_final_block(lfa, s_red);
  }
  if (order_copy != nullptr && hint_slot != nullptr) order_copy += (size_t)hint_slot[0] * (size_t)ntiles;
  dilate_depth_cuts(cd, (int)((by * 8 + bx) * 256 + threadIdx.x),
                    (int)(8 * ny * 256));
  // the forward's last kernel: a synchronisation error of this frame (olsr_state.h, counters[8]) reaches the caller here.
  // (When this launch also compacts the backward's rows — forward_tail_kernel —, an error raised by THOSE blocks, a look-back
  //  of the compaction that gives up, may come after this read: it is then reported by the backward of the frame, through
  //  tau_final_kernel, like any error of a backward-side compaction; a forward that is never followed by a backward uses no
  //  rows, so nothing of what it returned depends on them.  ADVICE round 5.)
  if (bx == 0 && by == 0 && threadIdx.x == 0 && counters[8] != 0) {
    if (num_rendered_dev != nullptr) num_rendered_dev[1] = 2;
    if (sticky != nullptr) __hip_atomic_store(sticky, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else if (bx == 0 && by == 0 && threadIdx.x == 0 && counters[9] != 0) {
    // a tile's depth cut-off may have hidden contributions (OLSR_STATUS_CUT_MISS); an overflow (1) stays
    if (num_rendered_dev != nullptr && num_rendered_dev[1] == 0) num_rendered_dev[1] = 3;
  }
  const int x = bx;  // XCD
  const int q = ntiles >> 3, r = ntiles & 7;
  const int start = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  const int len = q + (x < r ? 1 : 0);
  const int len64 = (len + 63) & ~63;
  u32 sum0 = 0, sum1 = 0;
  const bool summing = mailbox != nullptr && by == 0;  // the first block of each chunk also sums its two planes
  for (int j = threadIdx.x; j < len64; j += 256) {
    const u32 w = (j < len) ? work[start + j] : 0u;
    s_work[j] = w;
    sum0 += w;
    if (summing && j < len) sum1 += work[ntiles + start + j];
  }
  if (summing) sum_and_post_live_rows(sum0, sum1, 8u, live_rows, mailbox, seq);  // (block-uniform; syncs inside)
  if (len == 0) return;  // (fewer than 8 tiles: this XCD's chunk is empty)
  __syncthreads();
  // 16 tiles per block, 16 lanes per tile: lane s of a tile's row compares against the entries j = 4 s + 64 k ..
  // (one 16-byte LDS read each), the 16 partial ranks are summed inside the row with DPP rotations.  The zero padding
  // never outranks anything (a padded slot j >= len has weight 0 <= wi and j > i).
  const int i = by * 16 + (threadIdx.x >> 4);  // tile (clamped: every lane takes part in the row sum)
  const int sub = threadIdx.x & 15;
  const int ii = i < len ? i : len - 1;
  const u32 wi = s_work[ii];
  int rank = 0;
  for (int j = 4 * sub; j < len64; j += 64) {
    const uint4 w4 = *reinterpret_cast<const uint4*>(&s_work[j]);
    rank += (w4.x > wi) || (w4.x == wi && j < ii);
    rank += (w4.y > wi) || (w4.y == wi && j + 1 < ii);
    rank += (w4.z > wi) || (w4.z == wi && j + 2 < ii);
    rank += (w4.w > wi) || (w4.w == wi && j + 3 < ii);
  }
  float rf = (float)rank;  // (exact: ranks are far below 2^24)
  rf += dpp_mov<0x128>(rf);  // row_ror:8
  rf += dpp_mov<0x124>(rf);  // row_ror:4
  rf += dpp_mov<0x122>(rf);  // row_ror:2
  rf += dpp_mov<0x121>(rf);  // row_ror:1
  if (sub == 0 && i < len) {
    const int rk = (int)rf;
    order[start + rk] = (u32)(start + i);
    if (order_copy != nullptr) order_copy[start + rk] = (u32)(start + i);  // the caller's hint for its next frame
  }
}

__global__ __launch_bounds__(256) void tile_order_kernel(const TileOrderArgs ta) {
  tile_order_block(ta, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}

// The forward's last launch when its caller has announced the backward's row capacity (olsr_scene.backward_row_capacity): the
// tile order and the row compaction depend on the forward composite only, not on each other — one launch, the 8 ny light
// tile-order blocks first (the first four waves of a block; the others leave at once), then the row compaction's blocks.
__global__ __launch_bounds__(ROWS_THREADS) void forward_tail_kernel(const TileOrderArgs ta, const int ny,
                                                                    const RowCompactionArgs ra, const u32 nb_rows) {
  const int nb_order = 8 * ny;
  if ((int)blockIdx.x < nb_order) {
    if (threadIdx.x >= 256) return;  // (wave-uniform; a finished wave no longer counts at the block's barriers)
    tile_order_block(ta, (int)blockIdx.x & 7, (int)blockIdx.x >> 3, ny);
    return;
  }
  row_compaction_block(ra, nb_rows);
}

// images beyond ~120 k tiles (8K x 8K): a chunk no longer fits the LDS rank sort; keep the natural order
__global__ __launch_bounds__(256) void tile_order_identity_kernel(const u32* __restrict__ work, u32* __restrict__ order,
                                                                  u32* __restrict__ order_copy, int ntiles,
                                                                  u32* __restrict__ live_rows, int32_t* mailbox,
                                                                  int32_t seq, const int32_t* __restrict__ counters,
                                                                  int32_t* __restrict__ num_rendered_dev, int32_t* sticky,
                                                                  const u32* __restrict__ hint_slot, const CutDilate cd,
                                                                  const LossFinalArgs lfa) {
  if (lfa.partials != nullptr && blockIdx.x == gridDim.x - 1) {
    __shared__ double s_red[4][LOSS_SUMS];
    loss_final_block(lfa, s_red);
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (order_copy != nullptr && hint_slot != nullptr) order_copy += (size_t)hint_slot[0] * (size_t)ntiles;
  dilate_depth_cuts(cd, i, (int)(gridDim.x * 256));
  if (i == 0 && counters[8] != 0) {
    if (num_rendered_dev != nullptr) num_rendered_dev[1] = 2;
    if (sticky != nullptr) __hip_atomic_store(sticky, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else if (i == 0 && counters[9] != 0) {
    if (num_rendered_dev != nullptr && num_rendered_dev[1] == 0) num_rendered_dev[1] = 3;
  }
  if (i < ntiles) {
    order[i] = (u32)i;
    if (order_copy != nullptr) order_copy[i] = (u32)i;
  }
  if (mailbox != nullptr)
    sum_and_post_live_rows(i < ntiles ? work[i] : 0u, i < ntiles ? work[ntiles + i] : 0u, gridDim.x, live_rows, mailbox, seq);
}

void launch_tile_order(const uint32_t* tile_work, uint32_t* tile_order, uint32_t* order_copy, int ntiles,
                       uint32_t* live_rows, int32_t* rows_mailbox, int32_t rows_seq, const int32_t* counters,
                       int32_t* num_rendered_dev, int32_t* sticky_error, const uint32_t* hint_slot, float* depth_cut,
                       int gx, int gy, const LossFinalArgs& loss_final, const ForwardTailRows* rows, hipStream_t st) {
  if (ntiles <= 0) return;
  const CutDilate cd{depth_cut, gx, gy};
  const int len = (ntiles >> 3) + 1;
  if (sizeof(u32) * (size_t)(len + 64) > 60 * 1024) {
    if (rows != nullptr)  // (no merged form for the identity order: the compaction is its own launch, in front)
      launch_row_compaction(rows->flags, rows->n_host, rows->n_dev, rows->packed_ref15, rows->rowbase, rows->row_status,
                            rows->sync, rows->row_capacity, rows->counters, nullptr, st);
    tile_order_identity_kernel<<<(ntiles + 255) / 256, 256, 0, st>>>(tile_work, tile_order, order_copy, ntiles, live_rows,
                                                                     rows_mailbox, rows_seq, counters, num_rendered_dev,
                                                                     sticky_error, hint_slot, cd, loss_final);
    return;
  }
  const TileOrderArgs ta{tile_work, tile_order, order_copy, ntiles, live_rows, rows_mailbox, rows_seq, counters,
                         num_rendered_dev, sticky_error, hint_slot, cd, loss_final};
  const int ny = (len + 15) / 16;
  if (rows != nullptr) {
    const int shift = rows->packed_ref15 ? 4 : 0;
    const u32 mask = rows->packed_ref15 ? 3u : 15u;
    const int64_t n_host = rows->n_host < 0 ? 0 : rows->n_host;
    const int nb = (int)((n_host + 1 + ROWS_CHUNK - 1) / ROWS_CHUNK);
    const RowCompactionArgs ra{rows->flags, n_host, rows->n_dev, shift, mask, rows->rowbase, rows->row_status, rows->sync,
                               (long long)rows->row_capacity, rows->counters, nullptr,
                               sort_knobs().spin_limit.load(std::memory_order_relaxed)};
    forward_tail_kernel<<<8 * ny + nb, ROWS_THREADS, sizeof(u32) * (size_t)(len + 64), st>>>(ta, ny, ra, (u32)nb);
    return;
  }
  tile_order_kernel<<<dim3(8, ny), 256, sizeof(u32) * (size_t)(len + 64), st>>>(ta);
}

}  // namespace olsr
