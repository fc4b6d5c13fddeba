// k_sort.hip — stable LSD radix sort with ONE kernel per pass.
//
// Replaces cub::DeviceRadixSort::SortPairs (CR/rasterizer_impl.cu:478-483) together with k_binning.hip's split
// into a P-sized depth sort and an R-sized tile sort.  The multi-kernel passes of launch_radix_sort (histogram ->
// device-wide scan -> scatter, k_binning.hip) spend most of their time in launch latency when the whole sort is a
// few MB: 123-block grids of 4 waves on a 256-CU part, 8 launches to sort 4 MB.  Here a pass is one launch:
//
//   * the digit totals of ALL passes of a sort are counted up front in one read of the keys (sort_hist_kernel) —
//     keys do not change between LSD passes, only their order does;
//   * a block takes its chunk in TICKET order (dynamic block id from one atomic), counts its digits, PUBLISHES the
//     counts (one 16-bit word per digit: bit 15 = ready, the data is the flag — no fence), and sums the counts of all
//     its predecessors — ONE batch of independent 16-byte loads per thread (eight digits each), spinning only on the
//     few words that are not published yet: every dependent trip to the fabric costs ~2 us, so there is exactly one.  A predecessor never waits for anything
//     before publishing and has already started (it holds an earlier ticket), so the wait is deadlock-free under any
//     dispatch order.  This is the chained-scan idea of Onesweep without its serial look-back: on MI355X every
//     block of a 500 k-key sort is resident at once, a look-back chain would be walked in lockstep, while summing
//     published counts is a batch of independent, coalesced L2 reads (<= blocks x digits words);
//   * 1024-thread blocks (16 waves) with 2-8 keys per thread keep a CU's LDS and issue slots busy where the old
//     4-wave blocks ran 16 dependent ranking rounds per wave;
//   * ranking: per-wave digit counters in LDS, 64-bit ballots for the rank inside a round of 64 keys (match-any
//     over the digit bits), then the block's keys are ordered by digit in LDS so that every digit's run leaves with
//     consecutive lanes -> coalesced stores.
// The last tile-sort pass also derives the per-tile ranges (identifyTileRanges, CR/rasterizer_impl.cu:116-138) and
// skips the sorted keys nobody reads; the first one clears the liveness flags of the frame's instances.
#include <cstdio>
#include <cstdlib>

#include "olsr_device.h"
#include "olsr_kernels.h"

namespace olsr {

constexpr u32 FS_READY = 0x80000000u;
constexpr int FS_T = FUSED_SORT_THREADS;  // 1024
constexpr int FS_W = FS_T / 64;           // 16 waves
// (the look-back's spin bound is SortKnobs::spin_limit, default 1 << 22 polls: seconds; a predecessor publishes within microseconds)

__device__ __forceinline__ int64_t fs_bounded_n(int64_t n_host, const int32_t* n_dev) {
  if (n_dev) {
    const int64_t nd = (int64_t)(*n_dev);
    return nd < n_host ? nd : n_host;
  }
  return n_host;
}

__device__ __forceinline__ u32 fs_wave_incl_scan(u32 v) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u32 o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}
// exclusive prefix across the threads of a block of NW waves; s_w: NW words of LDS scratch (reusable after return)
template <int NW = FS_W>
__device__ __forceinline__ u32 fs_block_excl_scan(u32 v, u32* s_w) {
  const int lane = lane_id(), w = threadIdx.x >> 6;
  const u32 incl = fs_wave_incl_scan(v);
  if (lane == 63) s_w[w] = incl;
  __syncthreads();
  u32 base = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) base += (i < w) ? s_w[i] : 0u;
  __syncthreads();
  return base + incl - v;
}

// Published digit counts: one 16-bit word per digit (bit 15 = ready, count <= 8192 below), eight of them per 16-byte
// granule.  Granules are written and read whole with agent-scope (sc1: write-through / L1-bypassing) accesses; every
// 16-bit word carries its own ready bit, so a reader needs no ordering between them.
typedef unsigned short u16;
constexpr u32 FS_READY16 = 0x8000u;
constexpr u32 FS_READY_PAIR = 0x80008000u;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool fs_granule_ready(const u32x4& v) {
  return ((v.x & v.y & v.z & v.w) & FS_READY_PAIR) == FS_READY_PAIR;
}

// ---- digit totals of every pass of a sort, one read of the keys ---------------------------------------------
// hist[p][d] += #keys whose digit p is d (hist zeroed by an earlier kernel of the frame); digits are `db` bits wide.
// `house` (block 0 only, may be null): the frame's bookkeeping that used to be finalize_counts_kernel.
struct FrameHousekeeping {
  const u32* part_rect;   // preprocess' per-block partial sums
  const u32* part_count;
  int nparts;
  long long capacity;
  int32_t* counters;
  int32_t* num_rendered_dev;
  u32* ranges;
  int nranges;  // 2 * tiles
  int32_t* host_mailbox;
  int32_t host_seq;
  u32* live_rows;  // [4] zeroed here, summed by the tile-order kernel
  u32* hint_base;  // (may be null) per-view tile orders of the synchronising entry: wave 1 of block 0 picks this frame's slot
  const float* view;
};

// One wave: compare the frame's view matrix with the HINT_SLOTS stored ones and name the nearest slot (within HINT_VIEW_TOL per
// entry), else recycle the least recently used one (olsr_api.hip describes the buffer).  Round 4 ran this as a launch of its
// own in front of the forward composite.
__device__ __forceinline__ void hint_pick_wave(u32* base, const float* __restrict__ view, int lane) {
  float d = __builtin_inff();
  u32 age = 0xFFFFFFFFu;
  if (lane < HINT_SLOTS) {
    d = 0.f;
    for (int i = 0; i < 16; ++i) {
      const float sv = __uint_as_float(base[4 + HINT_SLOTS + 16 * lane + i]);
      const float e = fabsf(view[i] - sv);
      d = (e == e && d >= e) ? d : ((e == e) ? e : __builtin_inff());  // max; a NaN (empty slot) matches nothing
    }
    age = base[4 + lane];
  }
  // nearest slot, else the least recently used one (ties: the lower slot)
  float dbest = d;
  int ibest = lane;
  u32 abest = age;
  int iold = lane;
  for (int m = 32; m >= 1; m >>= 1) {
    const float od = __shfl_xor(dbest, m);
    const int oi = __shfl_xor(ibest, m);
    if (od < dbest || (od == dbest && oi < ibest)) { dbest = od; ibest = oi; }
    const u32 oa = (u32)__shfl_xor((int)abest, m);
    const int oo = __shfl_xor(iold, m);
    if (oa < abest || (oa == abest && oo < iold)) { abest = oa; iold = oo; }
  }
  const int pick = (dbest <= HINT_VIEW_TOL) ? ibest : iold;
  const float mine = (lane < 16) ? view[lane] : 0.f;
  if (lane < 16) base[4 + HINT_SLOTS + 16 * pick + lane] = __float_as_uint(mine);
  if (lane == 0) {
    const u32 c = base[1] + 1u;
    base[0] = (u32)pick;
    base[1] = c;
    base[4 + pick] = c;
  }
}

// The frame's bookkeeping, done by ONE 1024-thread block between two kernels of the frame (block 0 of the depth sort's
// histogram kernel, or the one-launch small depth sort): R = sum of the per-Gaussian instance counts (preprocess' block
// partials), the reference's num_rendered (rect binning), overflow against the caller's capacity; work-list and row counters
// reset; tile ranges set to "empty" (start = UINT_MAX, end = 0: the last tile-sort pass lowers / raises them with atomics).
// s_red: 2 x (T / 64) words of LDS; T = threads of the block.
template <int T>
__device__ __forceinline__ void frame_housekeeping(const FrameHousekeeping& house, int tid, u32 (*s_red)[T / 64]) {
  u32 rect = 0, cnt = 0;
  for (int i = tid; i < house.nparts; i += T) {
    rect += house.part_rect[i];
    cnt += house.part_count[i];
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    rect += __shfl_xor(rect, m);
    cnt += __shfl_xor(cnt, m);
  }
  if ((tid & 63) == 0) {
    s_red[0][tid >> 6] = rect;
    s_red[1][tid >> 6] = cnt;
  }
  for (int i = tid; i < house.nranges; i += T) house.ranges[i] = (i & 1) ? 0u : 0xFFFFFFFFu;
  __syncthreads();
  if (tid == 0) {
    rect = 0;
    cnt = 0;
    for (int i = 0; i < T / 64; ++i) {
      rect += s_red[0][i];
      cnt += s_red[1][i];
    }
    const bool ok = (long long)cnt <= house.capacity && cnt <= 0x7FFFFFFFu;
    house.counters[0] = (int32_t)cnt;
    house.counters[1] = ok ? (int32_t)cnt : 0;
    house.counters[2] = ok ? 0 : 1;
    house.counters[3] = (int32_t)rect;
    house.counters[4] = 0;
    house.counters[5] = 0;
    house.counters[6] = 0;
    house.counters[7] = 0;
    house.counters[8] = 0;  // synchronisation error of this frame (olsr_state.h)
    house.counters[9] = 0;  // a tile with a depth cut-off did not saturate (include/olsr.h, OLSR_STATUS_CUT_MISS)
    house.counters[11] = 0;  // "the rows of this frame were compacted for a scratch of N rows" (set by the row compaction)
    if (house.live_rows) {
      house.live_rows[0] = 0;
      house.live_rows[1] = 0;
      house.live_rows[3] = 0;
    }
    if (house.num_rendered_dev) {
      house.num_rendered_dev[0] = (int32_t)cnt;
      house.num_rendered_dev[1] = ok ? 0 : 1;
    }
    if (house.host_mailbox) {  // the drop-in entry's host is polling for the count (olsr_api.hip: PinnedCount)
      __hip_atomic_store(&house.host_mailbox[0], (int32_t)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&house.host_mailbox[1], house.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Visible-set compaction (round 6): the depth sort orders only the Gaussians that EMIT instances.
// The histogram kernel reads every key anyway; it now also reads the per-Gaussian instance counts, writes the (key, index) of
// the emitting Gaussians densely, IN INDEX ORDER — a wave's 256 consecutive keys are exactly one block of preprocess, whose
// count of emitting Gaussians (part_vis) gives the wave its base after one scan of <= 8 192 partials in LDS — and counts
// digits of those only.  The passes then sort n_out keys (known on the device; their grids stay sized for P, blocks beyond
// return at once).  A view that sees a fifth of the map — the room map — sorts 99 k keys instead of 500 k.  No extra launch,
// no look-back: the prefix comes from a finished kernel.
constexpr int COMPACT_MAX_PARTS = 8192;  // 2 M Gaussians (32 KB of LDS for the prefix); beyond: no compaction
struct CompactArgs {
  const u32* inst_count = nullptr;  // [P] instances per Gaussian; null: no compaction
  const u32* part_vis = nullptr;    // [nparts] emitting Gaussians per 256
  int nparts = 0;
  u32* out_keys = nullptr;          // [P]
  u32* out_gid = nullptr;           // [P]
  int32_t* n_out = nullptr;         // the number of emitting Gaussians
};

template <int T>
__global__ __launch_bounds__(T) void sort_hist_kernel(const u32* __restrict__ keys, int64_t n_host,
                                                         const int32_t* __restrict__ n_dev, int passes, int db,
                                                         u32* __restrict__ hist, FrameHousekeeping house,
                                                         int do_house, const u32* __restrict__ run_if, CompactArgs ca) {
  __shared__ u32 h[4][256];
  __shared__ u32 s_red[2][T / 64];
  extern __shared__ u32 s_pref[];  // [nparts] (compaction only): emitting Gaussians in front of every block of 256
  const int tid = threadIdx.x;
  // run_if (may be null): the sort is only needed when *run_if != 0 — the carried depth order could not be repaired
  // (k_order_carry.hip).  The frame's bookkeeping of block 0 happens either way.
  if (run_if != nullptr && __hip_atomic_load(run_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
    if (do_house && blockIdx.x == 0) {
      if (house.hint_base != nullptr && (tid >> 6) == 1) hint_pick_wave(house.hint_base, house.view, tid & 63);
      frame_housekeeping<T>(house, tid, s_red);
    }
    return;
  }
  for (int i = tid; i < 4 * 256; i += T) (&h[0][0])[i] = 0;
  const bool compact = ca.inst_count != nullptr;
  if (compact) {
    // exclusive prefix of part_vis over the blocks of 256 Gaussians: every thread a contiguous share, one block scan
    const int per = (ca.nparts + T - 1) / T;
    const int p0 = tid * per, p1 = min(p0 + per, ca.nparts);
    u32 mine = 0;
    for (int i = p0; i < p1; ++i) mine += ca.part_vis[i];
    __shared__ u32 s_scan[2 * (T / 64)];
    u32 run = fs_block_excl_scan<T / 64>(mine, s_scan);
    for (int i = p0; i < p1; ++i) {
      s_pref[i] = run;
      run += ca.part_vis[i];
    }
    if (blockIdx.x == 0 && p0 < ca.nparts && p1 == ca.nparts) *ca.n_out = (int32_t)run;  // (the thread that holds the last share)
  }
  __syncthreads();
  const int64_t n = fs_bounded_n(n_host, n_dev);
  const u32 mask = (1u << db) - 1u;
  const int64_t stride = (int64_t)gridDim.x * T * 4;
  constexpr int HU = 4;  // independent 16-byte loads in flight per thread (the loop is latency-bound otherwise)
  for (int64_t i00 = ((int64_t)blockIdx.x * T + tid) * 4; i00 < n; i00 += stride * HU) {
    u32 k[HU][4];
    u32 emits[HU];  // bit j: key j of slice u belongs to a Gaussian that emits instances (all ones without compaction)
#pragma unroll
    for (int u = 0; u < HU; ++u) {
      const int64_t i0 = i00 + (int64_t)u * stride;
      emits[u] = 0xFu;
      if (i0 + 4 <= n) {
        const uint4 q = *reinterpret_cast<const uint4*>(keys + i0);
        k[u][0] = q.x; k[u][1] = q.y; k[u][2] = q.z; k[u][3] = q.w;
        if (compact) {
          const uint4 c = *reinterpret_cast<const uint4*>(ca.inst_count + i0);
          emits[u] = (c.x ? 1u : 0u) | (c.y ? 2u : 0u) | (c.z ? 4u : 0u) | (c.w ? 8u : 0u);
        }
      } else {
        if (compact) emits[u] = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          k[u][j] = (i0 + j < n) ? keys[i0 + j] : 0u;
          if (compact && i0 + j < n && ca.inst_count[i0 + j] != 0u) emits[u] |= 1u << j;
        }
      }
    }
    if (compact) {
      // the wave's 64 x 4 consecutive keys are ONE block of 256 Gaussians of preprocess: its emitting ones go, in index order,
      // behind those of all earlier blocks
#pragma unroll
      for (int u = 0; u < HU; ++u) {
        const int64_t i0 = i00 + (int64_t)u * stride;
        const int64_t wave0 = i0 - 4 * (int64_t)(tid & 63);  // (wave-uniform; a multiple of 256)
        if (wave0 >= n) continue;
        const u32 c = (u32)__builtin_popcount(emits[u]);
        u32 pos = s_pref[(int)(wave0 >> 8)] + fs_wave_incl_scan(c) - c;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (emits[u] & (1u << j)) {
            ca.out_keys[pos] = k[u][j];
            ca.out_gid[pos] = (u32)(i0 + j);
            ++pos;
          }
      }
    }
#pragma unroll
    for (int u = 0; u < HU; ++u) {
      const int64_t i0 = i00 + (int64_t)u * stride;
      // (block-uniform: nothing of this block's u-th slice lies below n)
      if ((int64_t)blockIdx.x * T * 4 + (i00 - ((int64_t)blockIdx.x * T + tid) * 4) + (int64_t)u * stride >= n) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool valid = i0 + j < n && ((emits[u] >> j) & 1u) != 0u;
        for (int p = 0; p < passes; ++p) {
          const u32 dg = (k[u][j] >> (db * p)) & mask;
          // the high digits of depth keys (sign, exponent) and of tile ids are nearly constant across a wave: 64
          // lanes on one LDS counter serialise, so a wave whose valid lanes all agree adds its population with one lane
          const u64 vm = ballot(valid);
          const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)(valid ? dg : 0xFFFFu));
          const bool uniform = vm != 0ull && ballot(valid && dg == d0) == vm;  // (d0 of an invalid lane matches nothing)
          if (uniform) {
            if (lane_id() == (int)__builtin_ctzll(vm)) atomicAdd(&h[p][d0], (u32)__popcll(vm));
          } else if (valid) {
            atomicAdd(&h[p][dg], 1u);
          }
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < passes * 256; i += T) {
    const u32 c = (&h[0][0])[i];
    if (c != 0) atomicAdd(&hist[i], c);
  }
  if (do_house && blockIdx.x == 0 && house.hint_base != nullptr && (tid >> 6) == 1)
    hint_pick_wave(house.hint_base, house.view, tid & 63);  // (wave 1: wave 0's lane 0 finishes the counters below)
  if (do_house && blockIdx.x == 0) frame_housekeeping<T>(house, tid, s_red);
}

// ---- optional phase timing (olsr_debug_sort_timing): block b of every pass launched while it is set records the
// shader clock (s_memtime) at its phase boundaries into timing[(launch * max_blocks + b) * 8 + phase]
struct SortTiming {
  unsigned long long* buf = nullptr;
  int max_blocks = 0, max_launches = 0, launch = 0;
};
static thread_local SortTiming g_timing;
void debug_set_sort_timing(unsigned long long* buf, int max_blocks, int max_launches) {
  g_timing.buf = buf;
  g_timing.max_blocks = max_blocks;
  g_timing.max_launches = max_launches;
  g_timing.launch = 0;
}
#define FS_STAMP(k)                                                                  \
  do {                                                                               \
    if (timing != nullptr && tid == 0) timing[(size_t)b * 8 + (k)] = __builtin_readcyclecounter(); \
  } while (0)

// ---- the rank of a key among the equal digits of its round of 64 (round 5: written out by hand) ------------------------------
// peers = the lanes of the wave that hold the same DB-bit digit.  Per bit: one ballot of the bit, and the lane keeps the lanes
// that agree with it — peers &= (bit set ? ballot : ~ballot) = peers & ~(ballot ^ m) with m = 0 - bit (all ones where the bit is
// set): an extract, a compare, a subtract and an xnor + and per half — 7 vector instructions per bit.  The compiler's rendering
// of `peers &= one ? bm : ~bm` on 64-bit values took 13 (two compares, a select, a 64-bit add to build the mask, two xors, two
// ands); the sorts are a tenth of the frame's vector instructions and every one of them is in this loop.  The rank below the
// lane and the group's size come from v_mbcnt / v_bcnt on the halves.  Scalar masks stay out of it: the halves live in VGPRs.
typedef __attribute__((address_space(3))) u32 lds_u32;
template <int DB>
__device__ __forceinline__ void wave_match_rank(u32 dg, bool valid, u32& rank, u32& pop) {
  const u64 vm = ballot(valid);
  u32 lo = (u32)vm, hi = (u32)(vm >> 32);
#pragma unroll
  for (int bit = 0; bit < DB; ++bit) {
    const u32 b = (dg >> bit) & 1u;
    const u64 bm = ballot(b != 0u);
    const u32 m = 0u - b;
    lo &= ~((u32)bm ^ m);
    hi &= ~((u32)(bm >> 32) ^ m);
  }
  rank = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));  // peers in lanes below this one
  pop = (u32)__builtin_popcount(lo) + (u32)__builtin_popcount(hi);
}
// the per-wave running start of a digit, in LDS: plain ds_read / ds_write that the compiler neither caches nor reorders (the
// `volatile u32*` this replaces was lowered to FLAT loads and stores with a vmcnt(0) wait after each)
__device__ __forceinline__ u32 lds_load(const u32* p) {
  return *reinterpret_cast<const volatile lds_u32*>((const lds_u32*)p);
}
__device__ __forceinline__ void lds_store(u32* p, u32 v) { *reinterpret_cast<volatile lds_u32*>((lds_u32*)p) = v; }

// ---- one pass ---------------------------------------------------------------------------------------------
// FLAGS (runtime, uniform): bit 0 = values are the identity (first tile-sort pass; also clears flags_clear[i]),
// bit 1 = do not write the sorted keys (last pass of a sort), bit 2 = derive tile ranges (last tile-sort pass).
// bit 3 = last depth pass: add every Gaussian's instance count (inst_count[g] = tiles_touched) to the total of the emission block
// its final depth rank falls in (emit_totals[rank / EMIT_CHUNK]), so the emission needs no scan of its own.
constexpr int FSF_IDENTITY = 1, FSF_NO_KEYS = 2, FSF_RANGES = 4, FSF_EMIT_TOTALS = 8;

template <int DB, int KPT, int T>
// (waves per SIMD the register allocation aims for: a sixteen-wave block of four or more keys per thread holds 52 - 150 KB of LDS,
//  so at most one or two of them share a CU — 4 waves per SIMD and 128 VGPRs; round 5 asked for 8 there and paid with 6 - 28
//  spilled VGPRs per lane on the shapes sort_plan picks for 1 M - 4 M keys: config 5's depth sort)
__global__ __launch_bounds__(T, (T == 1024 ? (KPT <= 2 ? 8 : 4) : (KPT <= 8 ? 8 : 6))) __attribute__((amdgpu_num_sgpr(80))) void sort_pass_kernel(const u32* __restrict__ keys_in,
                                                         const u32* __restrict__ vals_in, int64_t n_host,
                                                         const int32_t* __restrict__ n_dev, int shift,
                                                         const u32* __restrict__ ghist, u16* status, u32* ticket,
                                                         u32* __restrict__ keys_out, u32* __restrict__ vals_out,
                                                         int fsf, uint8_t* __restrict__ flags_clear, u32* ranges,
                                                         const u32* __restrict__ inst_count, u64* emit_totals,
                                                         unsigned long long* timing, int32_t* sync_error, int spin_limit,
                                                         int fault, const u32* __restrict__ run_if) {
  // (run_if: see sort_hist_kernel — a pass of a sort nobody needs returns before it draws a ticket)
  if (run_if != nullptr && __hip_atomic_load(run_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
  constexpr int TW = T / 64;  // waves of the block
  static_assert(T == 1024 || T == 256, "block sizes of the radix passes");
  constexpr u32 NB = 1u << DB;
  static_assert((int)NB <= T, "a thread per digit");
  constexpr u32 DMASK = NB - 1u;
  constexpr int CHUNK = T * KPT;
  constexpr int C = (int)NB / 8;   // 16-byte status granules per block row
  constexpr int RPB = T / C;    // predecessor rows read per batch (one granule per thread)
  constexpr int U = (DB <= 6) ? 4 : 8;  // batches in flight per thread: one round covers U * RPB >= 256 predecessors
  extern __shared__ __attribute__((aligned(16))) u32 fs_smem[];
  u32* cnt = fs_smem;              // [16][NB] per-wave digit counts -> per-wave local starts; later the look-back partials
  u32* dstart = cnt + TW * NB;   // [NB + 1] local start of digit d inside the block (+ sentinel)
  u32* gbase = dstart + NB + 4;    // [NB] global start of this block's run of digit d
  u16* pub = reinterpret_cast<u16*>(gbase + NB);  // [NB] this block's published row (16-byte aligned)
  u32* ex_key = gbase + NB + NB / 2;  // [CHUNK]
  u32* ex_val = ex_key + CHUNK;       // [CHUNK]
  __shared__ u32 s_bid;
  __shared__ u32 s_w[2 * TW];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // Chunk index = a TICKET, always: a block only ever waits for blocks that hold a smaller ticket, i.e. that have
  // started — deadlock-free under any dispatch order and any co-tenancy.  (Round 2 took the launch order instead while
  // the grid had at most one block per CU; with several frames in flight two such grids can capture each other's XCDs
  // and wait for predecessors that can no longer be placed — ADVICE round 2.)  The ticket's round trip to the fabric
  // (~2 us) is covered by what needs no chunk: clearing the counters and the prefix of the global digit totals.
  if (tid == 0) s_bid = atomicAdd(ticket, 1u);
  for (u32 i = tid; i < TW * NB; i += T) cnt[i] = 0;
  const u32 gh = ((u32)tid < NB) ? ghist[tid] : 0u;
  const u32 gdig = fs_block_excl_scan<TW>(gh, s_w);  // global start of every digit (two barriers: s_bid is visible after them)
  const u32 b = s_bid;
  FS_STAMP(0);
  if (timing != nullptr && tid == 0) {
    timing[(size_t)b * 8 + 6] = (unsigned long long)(__builtin_amdgcn_s_getreg(63508) & 15);  // HW_REG_XCC_ID
    timing[(size_t)b * 8 + 7] = (unsigned long long)blockIdx.x;
  }
  const int64_t n = fs_bounded_n(n_host, n_dev);
  const int64_t bbase = (int64_t)b * CHUNK;
  // The frame is already broken (an earlier kernel of it lost a predecessor's counts): keys and values may be garbage — the
  // last tile pass would index the tile ranges with them — so do nothing.  Uniform for an error raised by an earlier kernel;
  // an error raised inside THIS pass is seen by late blocks only, whose successors then give up in turn (the frame is lost
  // either way, and this pass's own stores are clamped).
  if (__hip_atomic_load(sync_error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  if (b >= gridDim.x) {  // a ticket beyond the grid: the ticket word was not the zero the frame's head left (corrupted)
    if (tid == 0) atomicOr(sync_error, 1);
    return;
  }
  if (bbase >= n) return;  // (an empty block has only empty successors: nobody waits for it)
  const int64_t wbase = bbase + (int64_t)w * (64 * KPT);

  u32 key[KPT], val[KPT];
#pragma unroll
  for (int r = 0; r < KPT; ++r) {
    const int64_t i = wbase + r * 64 + lane;
    const bool valid = i < n;
    key[r] = valid ? keys_in[i] : 0xFFFFFFFFu;
    val[r] = (fsf & FSF_IDENTITY) ? (u32)i : (valid ? vals_in[i] : 0u);
    if (valid) atomicAdd(&cnt[w * NB + ((key[r] >> shift) & DMASK)], 1u);
    if ((fsf & FSF_IDENTITY) && valid && flags_clear != nullptr) flags_clear[i] = 0;
  }
  __syncthreads();
  FS_STAMP(1);

  // thread d owns digit d: counts of the 16 waves -> block total (published) and per-wave starts
  const u32 d = (u32)tid;
  u32 c[TW];
  u32 tot = 0;
  if (d < NB) {
#pragma unroll
    for (int i = 0; i < TW; ++i) {
      c[i] = cnt[i * NB + d];
      tot += c[i];
    }
    pub[d] = (u16)(FS_READY16 | tot);
  }
  // local start of every digit inside the block
  const u32 start = fs_block_excl_scan<TW>(d < NB ? tot : 0u, s_w);
  // publish this block's row: NB / 8 granules of eight 16-bit counts, write-through (sc1)
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(status, 0, (int)(gridDim.x * NB * 2u), 0x00020000);
  if (tid < C && !(fault != 0 && b == 0)) {  // (fault: test hook — this block's counts never arrive)
    const u32x4 g4 = *reinterpret_cast<const u32x4*>(pub + 8 * tid);
    __builtin_amdgcn_raw_buffer_store_b128(g4, rsrc, (int)((b * NB + 8u * (u32)tid) * 2u), 0, /*sc1*/ 16);
  }
  if (d < NB) {
    dstart[d] = start;
    u32 run = start;
#pragma unroll
    for (int i = 0; i < TW; ++i) {
      cnt[i * NB + d] = run;
      run += c[i];
    }
    if (d == NB - 1) dstart[NB] = run;
  }
  __syncthreads();
  FS_STAMP(2);

  // rank: wave w walks its 64 * KPT consecutive keys in KPT rounds of 64; inside a round the rank among equal
  // digits comes from ballots, the running per-wave start from LDS -> order inside the block = ascending input index.
  // (Done BEFORE the look-back: it needs nothing from other blocks, and meanwhile their counts become visible — a
  // look-back issued right after the publish finds nothing ready and thousands of polling waves slow every publish.)
  {
    u32* my = cnt + w * NB;
#pragma unroll
    for (int r = 0; r < KPT; ++r) {
      const int64_t i = wbase + r * 64 + lane;
      const bool valid = i < n;
      const u32 dg = (key[r] >> shift) & DMASK;
      u32 rank, pop;
      wave_match_rank<DB>(dg, valid, rank, pop);
      if (valid) {
        const u32 st0 = lds_load(&my[dg]);
        if (rank == 0) lds_store(&my[dg], st0 + pop);
        ex_key[st0 + rank] = key[r];
        ex_val[st0 + rank] = val[r];
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  FS_STAMP(3);

  // counts of all predecessors: every thread owns one granule column (8 digits) of RPB-strided rows; U independent
  // 16-byte loads in flight, spinning only on granules that are not fully published yet
  {
    const int gc = tid % C, rsub = tid / C;
    u32 acc[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    bool gave_up = false;
    for (u32 bp0 = 0; bp0 < b; bp0 += RPB * U) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const u32 row = bp0 + (u32)rsub + (u32)u * RPB;
        v[u] = u32x4{FS_READY_PAIR, FS_READY_PAIR, FS_READY_PAIR, FS_READY_PAIR};  // (beyond b: ready, zero counts)
        if (row < b) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((row * NB + 8u * (u32)gc) * 2u), 0, 16);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const u32 row = bp0 + (u32)rsub + (u32)u * RPB;
        // (bounded: a predecessor publishes within microseconds; a corrupted state buffer must not hang the GPU)
        for (int spin = 0; !fs_granule_ready(v[u]) && spin < spin_limit; ++spin) {
          __builtin_amdgcn_s_sleep(4);
          v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((row * NB + 8u * (u32)gc) * 2u), 0, 16);
        }
        gave_up |= !fs_granule_ready(v[u]);
        acc[0] += v[u].x & 0x7FFFu;
        acc[1] += (v[u].x >> 16) & 0x7FFFu;
        acc[2] += v[u].y & 0x7FFFu;
        acc[3] += (v[u].y >> 16) & 0x7FFFu;
        acc[4] += v[u].z & 0x7FFFu;
        acc[5] += (v[u].z >> 16) & 0x7FFFu;
        acc[6] += v[u].w & 0x7FFFu;
        acc[7] += (v[u].w >> 16) & 0x7FFFu;
      }
    }
    // A predecessor that never published: the counts below are garbage.  Nothing is written out of bounds (the stores are
    // clamped), and the frame is marked: the forward reports it, the backward writes zero gradients.
    if (gave_up) atomicOr(sync_error, 1);
    // lanes of a wave that own the same granule column differ in the lane bits >= log2(C): fold them, then the lanes
    // < C hold the wave's partial sums of their eight digits (cnt is dead after the ranking: reused as [16][NB])
#pragma unroll
    for (int m = C; m < 64; m <<= 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += __shfl_xor(acc[i], m);
    }
    if (lane < C) {
#pragma unroll
      for (int i = 0; i < 8; ++i) cnt[w * NB + lane * 8 + i] = acc[i];
    }
  }
  __syncthreads();
  if (d < NB) {
    u32 pred = 0;
#pragma unroll
    for (int i = 0; i < TW; ++i) pred += cnt[i * NB + d];
    gbase[d] = gdig + pred;
  }
  __syncthreads();
  FS_STAMP(4);

  const int64_t rem = n - bbase;
  const u32 nvalid = rem >= CHUNK ? (u32)CHUNK : (u32)rem;
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    const u32 slot = (u32)k * T + (u32)tid;
    if (slot < nvalid) {
      const u32 kk = ex_key[slot];
      const u32 dg = (kk >> shift) & DMASK;
      const u32 ds = dstart[dg];
      const u32 pos = gbase[dg] + (slot - ds);
      // (always true for a healthy state buffer; if the look-back above ran into its spin bound — a state buffer
      //  corrupted from outside mid-frame — the counts are garbage and nothing may be written out of bounds)
      if ((int64_t)pos >= n) continue;
      if (!(fsf & FSF_NO_KEYS)) keys_out[pos] = kk;
      const u32 vv = ex_val[slot];
      vals_out[pos] = vv;
      if (fsf & FSF_EMIT_TOTALS) {
        // pos is the Gaussian's final depth rank.  Lanes are consecutive slots: runs of consecutive ranks, so the
        // emission block (rank / 1024) is piecewise constant across the wave — a segmented scan, and only the last
        // lane of every piece adds its piece's sum: one to three atomics per wave instead of 64.
        // (packed: instances in the low 40 bits, the number of Gaussians that emit any above — emit_total_pack, olsr_state.h)
        const u64 cntg = emit_total_pack(inst_count[vv]);  // (P x 4 bytes: stays in L2, unlike the 32-byte emission records)
        const u32 bucket = pos / (u32)EMIT_CHUNK;
        const u64 act = ballot(true);  // (evaluated by every active lane)
        const u32 b0 = (u32)__builtin_amdgcn_readfirstlane((int)bucket);
        if (act == ~0ull && ballot(bucket != b0) == 0ull) {
          // the common case: a full wave inside one emission block — a plain wave sum
          u64 t = cntg;
#pragma unroll
          for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m);
          if (lane == 0 && t) atomicAdd(&emit_totals[b0], t);
        } else {
          u64 run = cntg;
#pragma unroll
          for (int sft = 1; sft < 64; sft <<= 1) {
            const u64 o = __shfl_up(run, sft);
            const u32 ob = __shfl_up(bucket, sft);
            if (lane >= sft && ob == bucket) run += o;
          }
          // (a piece longer than the shift distance is covered because pieces are contiguous: equality with the lane
          //  sft below implies equality with every lane in between)
          const u32 nb_ = __shfl_down(bucket, 1);
          const bool next_active = (act >> ((lane + 1) & 63)) & 1ull;
          const bool last = (lane == 63) || !next_active || nb_ != bucket;
          if (last && run) atomicAdd(&emit_totals[bucket], run);
        }
      }
      if (fsf & FSF_RANGES) {
        // After the last pass equal keys (tile ids) are contiguous, and inside one digit's run of this block the
        // keys ascend (the input was sorted by the lower digits).  A tile's range starts where the key changes;
        // the head / tail of a block's digit run may continue a tile of the neighbouring block, so those use
        // atomicMin / atomicMax (ranges were initialised to {UINT_MAX, 0}); the true boundary always wins.
        const bool head = (slot == ds);
        const bool tail = (slot + 1 == dstart[dg + 1]) || (slot + 1 == nvalid);
        if (head || ex_key[slot - 1] != kk) atomicMin(&ranges[2 * kk], pos);
        if (tail || ex_key[slot + 1] != kk) atomicMax(&ranges[2 * kk + 1], pos + 1u);
      }
    }
  }
  FS_STAMP(5);
}

// ---- the whole depth sort in ONE launch, for at most 8 192 Gaussians (round 5) ----------------
// A radix pass has a floor of ~10 us whatever it sorts (launch, ticket, publish, look-back: dependent trips through the
// fabric), and the depth sort is a histogram launch and four passes: BASELINE config 1 spent 48.7 us ordering 10 k depth
// keys.  Up to SMALL_SORT_MAX keys fit one workgroup's registers and LDS: the four 8-bit LSD passes run inside the block —
// per-wave digit counters, ballots for the rank inside a round of 64 keys, an exchange through LDS between passes: the ranking
// of sort_pass_kernel without its publish / look-back — and nothing leaves the CU until the final order is written.  The
// block also does the frame's bookkeeping and picks the drop-in entry's hint slot (what block 0 of sort_hist_kernel does),
// so the stage is one launch instead of five.  Same result bit for bit: a stable LSD sort of (depth bits, index).
// Measured (scripts/probe/small_sort.py, depth_sort stage, one launch against histogram + four passes): 1 000 Gaussians 14.3
// against 34.6 us, 2 000: 15.7 / 40.1, 4 000: 23.2 / 46.2, 8 000: 36.0 / 47.0 — about 10 + 3.3 us per key and thread, one
// CU doing what 256 idle ones cannot help with — and 63 / 48 at 10 000 (sixteen keys per thread), so the limit is 8 192:
// BASELINE config 1 (10 k Gaussians) keeps the pass kernels and its 48 us.
constexpr int SMALL_SORT_MAX = 8192;

template <int KPT>
__global__ __launch_bounds__(FS_T) void sort_small_kernel(const u32* __restrict__ keys_in, int n,
                                                          u32* __restrict__ vals_out, const u32* __restrict__ inst_count,
                                                          u64* __restrict__ emit_totals, FrameHousekeeping house,
                                                          const u32* __restrict__ run_if) {
  constexpr u32 NB = 256, DMASK = 255;
  constexpr int CHUNK = FS_T * KPT;
  extern __shared__ __attribute__((aligned(16))) u32 fs_smem[];
  u32* cnt = fs_smem;             // [16][256] per-wave digit counts -> per-wave starts
  u32* ex_key = cnt + FS_W * NB;  // [CHUNK]
  u32* ex_val = ex_key + CHUNK;   // [CHUNK]
  __shared__ u32 s_w[2 * FS_W];
  __shared__ u32 s_red[2][FS_W];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (house.hint_base != nullptr && w == 1) hint_pick_wave(house.hint_base, house.view, lane);
  frame_housekeeping<FS_T>(house, tid, s_red);
  // (run_if: see sort_hist_kernel — the carried order was repaired, only the bookkeeping above was needed; uniform)
  if (run_if != nullptr && __hip_atomic_load(run_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
  const int wbase = w * (64 * KPT);
  u32 key[KPT], val[KPT];
#pragma unroll
  for (int r = 0; r < KPT; ++r) {
    const int i = wbase + r * 64 + lane;
    key[r] = (i < n) ? keys_in[i] : 0xFFFFFFFFu;
    val[r] = (u32)i;  // (values = Gaussian indices: the position in the first pass)
  }
#pragma unroll 1
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 8 * pass;
    __syncthreads();  // (the previous pass's exchange buffers have been read back; s_w / cnt are free)
    for (u32 i = tid; i < FS_W * NB; i += FS_T) cnt[i] = 0;
    __syncthreads();
    // per-wave digit counts and, kept in registers for the scatter below, every key's rank among the equal digits of its
    // round and that group's size: ONE ballot match per key and pass (the high digits of depth keys are nearly constant —
    // 64 lanes adding to one LDS counter would serialise; here the group's first lane adds its population)
    u32 rk[KPT];
#pragma unroll
    for (int r = 0; r < KPT; ++r) {
      const bool valid = wbase + r * 64 + lane < n;
      const u32 dg = (key[r] >> shift) & DMASK;
      u32 rank, pop;
      wave_match_rank<8>(dg, valid, rank, pop);
      rk[r] = rank | (pop << 8);
      if (valid && rank == 0) atomicAdd(&cnt[w * NB + dg], pop);  // (one lane per distinct digit: no same-address conflict)
    }
    __syncthreads();
    // thread d owns digit d: counts of the 16 waves -> the digit's start in the block, then per-wave starts
    const u32 d = (u32)tid;
    u32 c[FS_W];
    u32 tot = 0;
    if (d < NB) {
#pragma unroll
      for (int i = 0; i < FS_W; ++i) {
        c[i] = cnt[i * NB + d];
        tot += c[i];
      }
    }
    const u32 start = fs_block_excl_scan(d < NB ? tot : 0u, s_w);
    if (d < NB) {
      u32 run = start;
#pragma unroll
      for (int i = 0; i < FS_W; ++i) {
        cnt[i * NB + d] = run;
        run += c[i];
      }
    }
    __syncthreads();
    {
      u32* my = cnt + w * NB;
#pragma unroll
      for (int r = 0; r < KPT; ++r) {
        const bool valid = wbase + r * 64 + lane < n;
        const u32 dg = (key[r] >> shift) & DMASK;
        if (valid) {
          const u32 rank = rk[r] & 0xFFu;
          const u32 st0 = lds_load(&my[dg]);
          if (rank == 0) lds_store(&my[dg], st0 + (rk[r] >> 8));
          ex_key[st0 + rank] = key[r];
          ex_val[st0 + rank] = val[r];
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    if (pass < 3) {
#pragma unroll
      for (int r = 0; r < KPT; ++r) {
        const int i = wbase + r * 64 + lane;
        if (i < n) {
          key[r] = ex_key[i];
          val[r] = ex_val[i];
        }
      }
    }
  }
  // the final order, and every Gaussian's instance count added to the total of the emission block (1024 ranks) its rank falls
  // in: slot = k * 1024 + tid lies in block k for every thread, so a wave sum per k is all there is
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    const int slot = k * FS_T + tid;
    u64 t = 0;
    if (slot < n) {
      const u32 vv = ex_val[slot];
      vals_out[slot] = vv;
      t = emit_total_pack(inst_count[vv]);
    }
    static_assert(EMIT_CHUNK == FS_T, "one emission block per 1024 ranks");
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m);
    if (lane == 0 && t) atomicAdd(&emit_totals[k], t);
  }
}

template <int KPT>
static void launch_sort_small_t(const u32* keys, int n, u32* vals_out, const u32* inst_count, u64* emit_totals,
                                const FrameHousekeeping& h, const u32* run_if, hipStream_t st) {
  constexpr size_t smem = sizeof(u32) * ((size_t)FS_W * 256 + 2 * (size_t)FS_T * KPT);
  static bool attr_set = false;  // (per instantiation) blocks above 64 KB of LDS need the opt-in
  if (!attr_set && smem > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sort_small_kernel<KPT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  sort_small_kernel<KPT><<<1, FS_T, smem, st>>>(keys, n, vals_out, inst_count, emit_totals, h, run_if);
}

struct PassArgs {
  const u32 *kin, *vin;
  int64_t n_host;
  const int32_t* n_dev;
  int shift;
  const u32* ghist;
  u16* status;
  u32* ticket;
  u32 *kout, *vout;
  int fsf;
  uint8_t* flags;
  u32* ranges;
  const u32* inst_count;
  u64* emit_totals;
  int nblk;
  int32_t* sync_error;
  int fault;
  const u32* run_if;
};

template <int DB, int KPT, int T>
static void launch_pass_t(const PassArgs& a, hipStream_t st) {
  constexpr size_t NB = 1u << DB;
  constexpr size_t smem = sizeof(u32) * ((T / 64) * NB + NB + 4 + NB + NB / 2 + 2 * (size_t)T * KPT);
  static bool attr_set = false;  // (per instantiation) blocks above 64 KB of LDS need the opt-in
  if (!attr_set && smem > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sort_pass_kernel<DB, KPT, T>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  static bool printed = std::getenv("OLSR_SORT_DEBUG") == nullptr;  // (read once per instantiation)
  if (!printed) {
    int nb = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sort_pass_kernel<DB, KPT, T>, T, smem);
    std::fprintf(stderr, "[olsr] sort_pass_kernel<%d,%d,%d>: %zu B dynamic LDS, occupancy API: %d blocks/CU, grid %d\n", DB,
                 KPT, T, smem, nb, a.nblk);
    printed = true;
  }
  unsigned long long* timing = nullptr;
  if (g_timing.buf && g_timing.launch < g_timing.max_launches && a.nblk <= g_timing.max_blocks)
    timing = g_timing.buf + (size_t)(g_timing.launch++) * g_timing.max_blocks * 8;
  sort_pass_kernel<DB, KPT, T><<<a.nblk, T, smem, st>>>(a.kin, a.vin, a.n_host, a.n_dev, a.shift, a.ghist, a.status,
                                                        a.ticket, a.kout, a.vout, a.fsf, a.flags, a.ranges,
                                                        a.inst_count, a.emit_totals, timing, a.sync_error,
                                                        sort_knobs().spin_limit.load(std::memory_order_relaxed), a.fault,
                                                        a.run_if);
}

template <int DB, int T>
static void launch_pass_kt(int kpt, const PassArgs& a, hipStream_t st) {
  switch (kpt) {
    case 2: launch_pass_t<DB, 2, T>(a, st); break;
    case 4: launch_pass_t<DB, 4, T>(a, st); break;
    case 8: launch_pass_t<DB, 8, T>(a, st); break;
    case 12: launch_pass_t<DB, 12, T>(a, st); break;
    default: launch_pass_t<DB, 16, T>(a, st); break;
  }
}
template <int DB>
static void launch_pass_k(const SortPlan& plan, const PassArgs& a, hipStream_t st) {
  if (plan.threads == 256) launch_pass_kt<DB, 256>(plan.kpt, a, st);
  else launch_pass_kt<DB, 1024>(plan.kpt, a, st);
}

int fused_sort_digit_bits(int bits, int* passes_out) {
  const int passes = (bits + 7) / 8;
  int db = (bits + passes - 1) / passes;
  if (db < 4) db = 4;  // (digits wider than the key are harmless: the upper bits are zero)
  if (passes_out) *passes_out = passes;
  return db;
}

bool fused_sort_applicable(int64_t n_host, int bits) {
  return n_host > 0 && bits <= 32 && fused_sort_fits(n_host);
}

static FrameHousekeeping housekeeping_of(const FusedHouse* house) {
  FrameHousekeeping h{};
  if (house) {
    h.part_rect = house->part_rect;
    h.part_count = house->part_count;
    h.nparts = house->nparts;
    h.capacity = house->capacity;
    h.counters = house->counters;
    h.num_rendered_dev = house->num_rendered_dev;
    h.ranges = house->ranges;
    h.nranges = house->nranges;
    h.host_mailbox = house->host_mailbox;
    h.host_seq = house->host_seq;
    h.live_rows = house->live_rows;
    h.hint_base = house->hint_base;
    h.view = house->view;
  }
  return h;
}

// the depth sort of at most SMALL_SORT_MAX (8 192) Gaussians in one launch (sort_small_kernel); false: not applicable
// as_fallback: behind the repair of a carried depth order (k_order_carry.hip) the sort runs only when the repair missed, and
// what counts on every other frame is how many launches return at once: ONE for the one-launch sort against five for the
// histogram and the passes.  There the one-launch sort takes up to 16 384 Gaussians (sixteen keys per thread: ~63 us when it
// does run, against ~48 us for the passes — on the rare frame; BASELINE config 1, 10 k Gaussians: 28 -> 20 us on the others).
constexpr int SMALL_SORT_MAX_FALLBACK = 16384;
bool small_depth_sort_applicable(int64_t n, bool as_fallback) {
  // (a pinned keys-per-thread or the depth sort's fault hook ask for the pass kernels)
  return n > 0 && n <= (as_fallback ? SMALL_SORT_MAX_FALLBACK : SMALL_SORT_MAX) && sort_plan_forced_kpt() == 0 &&
         sort_knobs().small_sort.load(std::memory_order_relaxed) != 0 &&
         (sort_knobs().fault.load(std::memory_order_relaxed) & 1) == 0;
}
void launch_small_depth_sort(const uint32_t* keys, int n, uint32_t* order_out, const uint32_t* inst_count,
                             uint32_t* emit_totals, const FusedHouse* house, const uint32_t* run_if, hipStream_t st) {
  const FrameHousekeeping h = housekeeping_of(house);
  u64* et = reinterpret_cast<u64*>(emit_totals);
  if (n <= 2 * FS_T) launch_sort_small_t<2>(keys, n, order_out, inst_count, et, h, run_if, st);
  else if (n <= 4 * FS_T) launch_sort_small_t<4>(keys, n, order_out, inst_count, et, h, run_if, st);
  else if (n <= 8 * FS_T) launch_sort_small_t<8>(keys, n, order_out, inst_count, et, h, run_if, st);
  else launch_sort_small_t<16>(keys, n, order_out, inst_count, et, h, run_if, st);
}

bool depth_sort_compaction_applicable(int64_t P) {
  return P > 0 && (P + 255) / 256 <= COMPACT_MAX_PARTS && sort_knobs().compact.load(std::memory_order_relaxed) != 0;
}

void launch_sort_hist(const uint32_t* keys, int64_t n_host, const int32_t* n_dev, int bits, uint32_t* hist,
                      const FusedHouse* house, int threads, hipStream_t st, const uint32_t* run_if,
                      const SortCompaction* compaction) {
  int passes;
  const int db = fused_sort_digit_bits(bits, &passes);
  const FrameHousekeeping h = housekeeping_of(house);
  const int T = threads;
  int64_t nb = (n_host + (int64_t)T * 4 - 1) / ((int64_t)T * 4);  // 4 keys (one 16-byte load) per thread ...
  if (nb < 1) nb = 1;
#ifndef OLSR_HIST_BLOCKS
#define OLSR_HIST_BLOCKS 256  // (measured on 2.7 M keys: 64 / 128 / 256 / 384 / 512 blocks: 21 / 13 / 10 / 12 / 12 us)
#endif
  if (nb > OLSR_HIST_BLOCKS) nb = OLSR_HIST_BLOCKS;  // ... then a grid-stride loop: every block ends with one global atomic per non-empty bin,
                           // and same-address atomics serialise (~10-20 ns each), so few, fat blocks
  CompactArgs ca{};
  size_t smem = 0;
  if (compaction != nullptr) {
    ca.inst_count = compaction->inst_count;
    ca.part_vis = compaction->part_vis;
    ca.nparts = (int)((n_host + 255) / 256);
    ca.out_keys = compaction->out_keys;
    ca.out_gid = compaction->out_gid;
    ca.n_out = compaction->n_out;
    smem = sizeof(u32) * (size_t)ca.nparts;
  }
  if (T == 256) sort_hist_kernel<256><<<(int)nb, 256, smem, st>>>(keys, n_host, n_dev, passes, db, hist, h, house ? 1 : 0, run_if, ca);
  else sort_hist_kernel<1024><<<(int)nb, 1024, smem, st>>>(keys, n_host, n_dev, passes, db, hist, h, house ? 1 : 0, run_if, ca);
}

// Sorts (key, val) pairs on the low `bits` bits of key, ceil(bits / 8) passes.  hist / status / tickets must have been
// zeroed and hist filled by launch_sort_hist.  Returns 0 if the result ends in (key_a, val_a), 1 if in (key_b, val_b).
int launch_sort_fused(const SortBuffers& b, const SortPlan& plan, int64_t n_host, const int32_t* n_dev, int bits,
                      bool vals_in_identity, const uint32_t* hist, uint32_t* status, uint32_t* tickets, uint8_t* flags_clear, uint32_t* ranges,
                      const uint32_t* inst_count, uint32_t* emit_totals, int32_t* sync_error, int fault, hipStream_t st,
                      uint32_t* final_vals_out, const uint32_t* run_if) {
  if (n_host <= 0) return 0;
  int passes;
  const int db = fused_sort_digit_bits(bits, &passes);
  const size_t NB = (size_t)1 << db;
  u32 *kin = b.key_a, *kout = b.key_b, *vin = b.val_a, *vout = b.val_b;
  int where = 0;
  for (int p = 0; p < passes; ++p) {
    int fsf = 0;
    if (p == 0 && vals_in_identity) fsf |= FSF_IDENTITY;
    if (p == passes - 1) fsf |= FSF_NO_KEYS | (ranges ? FSF_RANGES : 0) | (emit_totals ? FSF_EMIT_TOTALS : 0);
    // (final_vals_out: the last pass leaves the sorted values there instead of in the ping-pong buffer — the carried depth order)
    PassArgs a{kin, vin, n_host, n_dev, db * p, hist + 256 * p,
               reinterpret_cast<u16*>(status) + (size_t)p * plan.nblk * NB, tickets + p, kout,
               (p == passes - 1 && final_vals_out != nullptr) ? final_vals_out : vout, fsf, flags_clear,
               ranges, inst_count, reinterpret_cast<u64*>(emit_totals), plan.nblk, sync_error, p == 0 ? fault : 0, run_if};
    switch (db) {
      case 4: launch_pass_k<4>(plan, a, st); break;
      case 5: launch_pass_k<5>(plan, a, st); break;
      case 6: launch_pass_k<6>(plan, a, st); break;
      case 7: launch_pass_k<7>(plan, a, st); break;
      case 8: launch_pass_k<8>(plan, a, st); break;
      default: break;
    }
    u32* t = kin; kin = kout; kout = t;
    t = vin; vin = vout; vout = t;
    where ^= 1;
  }
  return where;
}

}  // namespace olsr
