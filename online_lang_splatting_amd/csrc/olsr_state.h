// olsr_state.h — layout of the three opaque state buffers in HBM.
//
// Counterpart of GeometryState / ImageState / BinningState (CR/rasterizer_impl.h:31-81,
// CR/rasterizer_impl.cu:155-212).  The buffers only need to be self-consistent between a
// forward and its backward; the layout is MI355X-first (struct-of-arrays, every array on a
// 256-byte boundary so wave-wide accesses start on a full cache line):
//
//   geometry  — per-Gaussian results of preprocess, the depth order, instance offsets and the
//               radix-sort scratch for the P-sized depth sort.
//   image     — final transmittance, last-contributor index, per-tile ranges.
//   binning   — per-instance arrays: tile keys (ping/pong), sorted Gaussian list, the
//               sorted->emission map `src`, per-instance slot flags (written by the forward
//               composite) and `rowbase`, the exclusive scan of popcount(flags) that compacts
//               the backward's partial-gradient rows.
//   scratch   — (backward only) the compacted rows [live (instance, slot) pairs][grad_row(F)].
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <atomic>
#include <cstdlib>

namespace olsr {

constexpr size_t ALIGN = 256;
inline size_t align_up(size_t v) { return (v + ALIGN - 1) / ALIGN * ALIGN; }

// elements handled by one block of the radix passes / scans
constexpr int SORT_CHUNK = 4096;
constexpr int SCAN_CHUNK = 4096;
inline int sort_blocks(long long n) { return (int)((n + SORT_CHUNK - 1) / SORT_CHUNK); }
inline int scan_blocks(long long n) { return (int)((n + SCAN_CHUNK - 1) / SCAN_CHUNK); }

// Fused sort passes (k_sort.hip): one 1024-thread block per chunk of 1024 * kpt keys.  The runtime keeps ONE such block
// per CU, so a pass runs in ceil(nblk / 256) rounds and a round costs about (kpt + 1.5) x 2 us (measured on
// the 2.7 M-instance frame: 24 us at kpt = 12, 29 us at kpt = 16, i.e. about kpt + 11 us); kpt is the candidate that minimises rounds x (kpt + 11) — for the 2.7 M instances of the
// headline frame one round of 220 blocks at kpt = 12 instead of 1.3 rounds at kpt = 8.
struct SortPlan {
  int kpt;
  int nblk;
  int threads;  // 256 or 1024 threads per block
};
constexpr int FUSED_SORT_THREADS = 1024;     // the one-launch small sort, and the radix passes' larger block shape
constexpr int FUSED_SORT_MAX_BLOCKS = 16384;  // beyond: the multi-kernel passes (histogram, device-wide scan, scatter)
// (tuning / test knobs, process-wide: seeded from OLSR_SORT_KPT / OLSR_SORT_RESIDENT / OLSR_SORT_LEGACY once at load and
//  set by olsr_debug_sort_knobs — the tests that walk every kernel instantiation and the multi-round ticket order on small
//  inputs; no getenv on a call path)
struct SortKnobs {
  std::atomic<int> kpt{0}, resident{0}, legacy{0};
  // threads per block of the histogram and pass kernels: 0 (default) = the call decides — 1024, or 256 when the scene carries
  // OLSR_FLAG_FRAMES_IN_FLIGHT; 256 / 1024 = forced (tests, experiments).  With several frames in flight a 1024-thread block
  // cannot be placed while another frame's composite kernel fills the CUs — it needs sixteen free wave slots on ONE CU, which
  // only the composite's tail offers — so the radix passes of the other frames stalled until that composite ended (a kernel
  // trace of four frames in flight: 15 us passes stretched to 147 us).  Four-wave blocks with sixteen keys per thread get on
  // sooner: + 2 % frames/s with four frames in flight — and - 11 % with one (the depth sort 69 -> 108 us), hence per call.
  std::atomic<int> threads{0};
  std::atomic<int> compact{1};     // the depth sort orders only the Gaussians that emit instances (k_sort.hip); OLSR_SORT_COMPACT=0 /
                                   // olsr_debug_sort_compact(0): every Gaussian, as rounds 1-5 did
  std::atomic<int> small_sort{1};  // the one-launch depth sort of <= 8 192 Gaussians (k_sort.hip); OLSR_SORT_SMALL=0 / olsr_debug_sort_small(0): off
  // test hooks (olsr_debug_sync_fault): fault bit 0 / 1 = the block holding ticket 0 of the first depth / tile pass never
  // publishes its digit counts (what a status word corrupted mid-frame looks like to its successors); spin_limit = polls a
  // look-back makes before it gives up and raises the frame's synchronisation error
  std::atomic<int> fault{0}, spin_limit{1 << 22};
};
SortKnobs& sort_knobs();  // olsr_api.hip
inline int sort_plan_resident_blocks() {
  const int x = sort_knobs().resident.load(std::memory_order_relaxed);
  return x > 0 ? x : 256;
}
inline int sort_plan_threads(bool frames_in_flight = false) {
  const int forced = sort_knobs().threads.load(std::memory_order_relaxed);
  if (forced == 1024 || forced == 256) return forced;
  return frames_in_flight ? 256 : 1024;
}
inline int sort_plan_forced_kpt() {
  const int x = sort_knobs().kpt.load(std::memory_order_relaxed);
  return (x == 2 || x == 4 || x == 8 || x == 12 || x == 16) ? x : 0;
}
// n_is_capacity: n bounds a count that is only known on the device (olsr_forward_async); the rounds are then estimated
// for 85 % of it — callers size a capacity with headroom, blocks past the real count exit at once, and the choice only
// moves time, never the result.
// fill_pct: the share of a capacity the caller expects to be used (per-tile depth cut-offs leave a fraction of the instances).
inline SortPlan sort_plan(long long n, bool n_is_capacity = false, int fill_pct = 85, bool frames_in_flight = false) {
  static const int cand[5] = {2, 4, 8, 12, 16};
  const long long T = sort_plan_threads(frames_in_flight);
  // blocks of a round: 256 resident 1024-thread blocks (one per CU), or four times as many four-wave blocks
  const long long res = sort_plan_resident_blocks() * (1024 / T);
  const long long n_est = n_is_capacity ? (n * fill_pct + 99) / 100 : n;
  int best = sort_plan_forced_kpt();
  if (best == 2 && T == 256) best = 4;  // (a four-wave block takes at least 1024 keys: the status rows are reserved for that)
  // (a forced kpt whose block count would overrun the status rows reserved for FUSED_SORT_MAX_BLOCKS is ignored)
  if (best != 0 && (n + T * best - 1) / (T * best) > FUSED_SORT_MAX_BLOCKS) best = 0;
  // four-wave blocks: sixteen keys per thread, measured (config 3, four frames in flight, K = 60: kpt 4 / 8 / 12 / 16 ->
  // 2 233 / 2 177 / 2 327 / 2 350 frames/s against 2 304 with 1024-thread blocks): few blocks keep the look-back short
  if (best == 0 && T == 256) best = 16;
  if (best == 0) {
    double best_cost = 0.0;
    for (int i = (T == 256 ? 1 : 0); i < 5; ++i) {
      const long long chunk = T * cand[i];
      if ((n + chunk - 1) / chunk > FUSED_SORT_MAX_BLOCKS && i < 4) continue;
      const long long nb = (n_est + chunk - 1) / chunk;
      // a round costs kpt + 11 us (measured with 1024-thread blocks); with four-wave blocks a block's look-back reads the
      // published counts of up to four times as many predecessors: ~1 us per 128 of them
      const double cost = (double)((nb + res - 1) / res) * (cand[i] + 11.0) + (T == 256 ? (double)nb / 128.0 : 0.0);
      if (best == 0 || cost < best_cost) {
        best = cand[i];
        best_cost = cost;
      }
    }
  }
  return SortPlan{best, (int)((n + T * best - 1) / (T * best)), (int)T};
}
inline bool fused_sort_fits(long long n) { return (n + 4095) / 4096 <= FUSED_SORT_MAX_BLOCKS; }  // (256 threads x 16 keys)
// status words reserved for a sort of n keys: whatever kpt the plan picks (it may differ between a sized and a
// capacity-bounded launch of the same n), the rows of the smallest chunk cover it
inline size_t fused_status_words(long long n, int passes) {
  if (!fused_sort_fits(n)) return 0;
  long long nb = (n + 1023) / 1024;  // the smallest chunk: 256 threads x 4 keys (1024 threads x 2 keys is twice that)
  if (nb > FUSED_SORT_MAX_BLOCKS) nb = FUSED_SORT_MAX_BLOCKS;
  return (size_t)passes * (size_t)nb * 128;  // 256 16-bit counts per row
}
// single-pass scans (emission offsets, row compaction): elements per 1024-thread block
constexpr int EMIT_CHUNK = 1024;
// Per block of EMIT_CHUNK depth ranks the sort (or the repair of a carried order) leaves ONE 64-bit total for the emission:
// the instances the block's Gaussians emit (low 40 bits) and how many of them emit any (above).  The emission compacts the
// emitting ranks (emit_offsets_kernel): with sort keys for every Gaussian (k_preprocess.hip) a view that sees a fifth of the
// map has four silent ranks between two emitting ones.
constexpr int EMIT_TOTAL_SHIFT = 40;
__host__ __device__ inline unsigned long long emit_total_pack(uint32_t instances) {
  return (unsigned long long)instances | ((unsigned long long)(instances != 0u) << EMIT_TOTAL_SHIFT);
}
#ifndef OLSR_ROWS_THREADS
#define OLSR_ROWS_THREADS 1024  // threads of a row-compaction block (16 instances each)
#endif
constexpr int ROWS_CHUNK = 16 * OLSR_ROWS_THREADS;

struct Carver {
  char* base;
  size_t off = 0;
  // a non-null base is rounded up to ALIGN (total() reserves the slack)
  explicit Carver(void* b) : base(b ? (char*)(((uintptr_t)b + ALIGN - 1) / ALIGN * ALIGN) : nullptr) {}
  template <typename T>
  T* take(size_t count) {
    off = align_up(off);
    T* p = base ? (T*)(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
  size_t total() const { return align_up(off) + ALIGN; }
};

struct GeometryState {
  float* depths;          // [P]
  float* means2D;         // [P][2]
  float* conic_opacity;   // [P][4]
  float* cov3D;           // [P][6]
  float* rgb;             // [P][3]
  uint8_t* clamped;       // [P][3]
  uint8_t* blended;       // [P] 1 = some pixel blended some instance of the Gaussian in this frame's forward (zeroed by
                          //     preprocess, set by the forward composite's flush): only those can have gradient rows
  uint32_t* tiles_touched;  // [P] instances the Gaussian emits (rect area, or the exact tile count)
  float4* emit_rec;       // [P][2] everything the emission needs about a Gaussian in ONE 32-byte gather (it runs in
                          //     depth order): {mean x, mean y, conic a, conic b}, {conic c, cull t2, radius, #instances}
  uint32_t* key_a;        // [P] depth keys (ping)
  uint32_t* key_b;        // [P] (pong)
  uint32_t* val_a;        // [P] Gaussian ids (ping)
  uint32_t* val_b;        // [P] (pong)
  uint32_t* depth_order;  // alias of the buffer holding the final order (val_a after 4 passes)
  uint32_t* inst_start;   // [P] Gaussian id -> emission index of its first instance
  uint32_t* radix_table;  // [256 * sort_blocks(P)]
  uint32_t* scan_partials;  // [scan_blocks(max(P, table))]
  int32_t* counters;      // [16]: 0 = R (total instances), 1 = R_eff (0 on overflow), 2 = overflow flag,
                          //      3 = instances of the reference's rect binning (== R unless OLSR_BINNING_ELLIPSE),
                          //      4 = #large-footprint Gaussians (backward), 5 = same for the emission (forward),
                          //      6 = live rows L, 7 = row-capacity overflow,
                          //      8 = synchronisation error: a look-back of a radix pass or of the row compaction ran into its
                          //          spin bound (a status word corrupted mid-frame).  Reset by the frame's first kernel; the
                          //          forward reports it as num_rendered_dev[1] = 2, the backward writes zero gradients and
                          //          reports status_dev[1] = 2 / OLSR_ERR_DEVICE (include/olsr.h),
                          //      9 = a tile with a depth cut-off did not saturate (OLSR_STATUS_CUT_MISS),
                          //      10 = Gaussians that emit instances (the length of the emission's compacted rank list),
                          //      11 = the rows were compacted for a scratch of this capacity (olsr_device.h: rows_stamp_of),
                          //      12 = Gaussians the depth sort ordered (the emitting ones) when it compacted its input.  [13..15] reserved
  float* tau_partials;    // [6 * ceil(P/128)] scratch of the backward's deterministic dL_dtau reduction
  float* gacc;            // [P][grad_row(F)] backward scratch: per-Gaussian sum of its instance rows
  uint4* big_list;        // [P] work lists {id, first instance, #instances} built by the emission: large footprints from
                          //     the front (count: counters[5]), medium ones from the back (count: counters[4])
  // ---- words the fused kernels synchronise through; zeroed by preprocess at the start of every forward
  uint32_t* sync_words;   // start of the zeroed region
  size_t sync_count;      // its length in 32-bit words
  uint32_t* tickets;      // [16] dynamic block ids: 0-3 depth passes, 4 scan+emit; [12] = carry_miss (below)
  uint32_t* sort_hist;    // [4][256] digit totals of the four depth passes
  uint32_t* sort_status;  // [4][blocks][256] per-block digit counts of the depth passes (bit 31 = published)
  uint32_t* emit_status;  // [ceil(P / EMIT_CHUNK)] x 64 bit: per-block totals for the emission (emit_total_pack above)
  uint32_t* part_rect;    // [ceil(P / 256)] preprocess' per-block sums: instances of the reference's rect binning
  uint32_t* part_count;   // [ceil(P / 256)] ... and instances this frame emits
  uint32_t* part_vis;     // [ceil(P / 256)] ... and Gaussians that emit any (the depth sort orders only those, k_sort.hip)
  uint32_t* carry_totals; // [ceil(P / EMIT_CHUNK) + 1] x 64 bit: the same totals for a REPAIRED carried depth order (k_order_carry.hip)
  uint32_t* carry_miss;   // one of the zeroed words (tickets[12]): != 0 = the carried order could not be repaired, the radix passes run
  static GeometryState carve(void* buf, size_t P, int grad_row_floats, size_t& bytes) {
    Carver c(buf);
    GeometryState g;
    g.depths = c.take<float>(P);
    g.means2D = c.take<float>(2 * P);
    g.conic_opacity = c.take<float>(4 * P);
    g.cov3D = c.take<float>(6 * P);
    g.rgb = c.take<float>(3 * P);
    g.clamped = c.take<uint8_t>(3 * P);
    g.blended = c.take<uint8_t>(P);
    g.tiles_touched = c.take<uint32_t>(P);
    g.emit_rec = c.take<float4>(2 * P);
    g.key_a = c.take<uint32_t>(P);
    g.key_b = c.take<uint32_t>(P);
    g.val_a = c.take<uint32_t>(P);
    g.val_b = c.take<uint32_t>(P);
    g.depth_order = g.val_a;
    g.inst_start = c.take<uint32_t>(P);
    const size_t table = 256 * (size_t)sort_blocks((long long)P);
    g.radix_table = c.take<uint32_t>(table);
    g.scan_partials = c.take<uint32_t>((size_t)scan_blocks((long long)(table > P ? table : P)) + 1);
    g.counters = c.take<int32_t>(16);
    g.tau_partials = c.take<float>(6 * ((P + 127) / 128) + 6);
    g.gacc = c.take<float>(P * (size_t)grad_row_floats);
    g.big_list = c.take<uint4>(P);
    {
      const size_t st_words = fused_status_words((long long)P, 4);
      const size_t emit_blocks = 2 * ((P + EMIT_CHUNK - 1) / EMIT_CHUNK + 1);  // 64-bit look-back words
      g.sync_count = 16 + 4 * 256 + st_words + emit_blocks;
      g.sync_count = (g.sync_count + 3) / 4 * 4;  // zeroed with 16-byte stores
      g.sync_words = c.take<uint32_t>(g.sync_count);
      g.tickets = g.sync_words;
      g.sort_hist = g.tickets + 16;
      g.emit_status = g.sort_hist + 4 * 256;
      g.sort_status = g.emit_status + emit_blocks;
    }
    g.part_rect = c.take<uint32_t>((P + 255) / 256 + 1);
    g.part_count = c.take<uint32_t>((P + 255) / 256 + 1);
    g.part_vis = c.take<uint32_t>((P + 255) / 256 + 1);
    g.carry_totals = c.take<uint32_t>(2 * ((P + EMIT_CHUNK - 1) / EMIT_CHUNK + 1));
    g.carry_miss = g.tickets + 12;
    bytes = c.total();
    return g;
  }
};

struct ImageState {
  float* final_T;       // [H*W]
  uint32_t* n_contrib;  // [H*W]
  uint32_t* ranges;     // [tiles][2]
  uint32_t* tile_work;  // [2][tiles] live (instance, slot) pairs, then live (instance, packed survivor wave) pairs of the
                        //     tile (written by the forward composite)
  uint32_t* tile_order; // [tiles] backward launch order: per XCD chunk, heaviest tile first
  uint32_t* live_rows;  // [4] {live (instance, slot) pairs, live (instance, packed survivor wave) pairs, -, blocks done} of
                        //     the frame: the backward's gradient-row count in its two row layouts, summed from tile_work
                        //     by the tile-order kernel when a host mailbox wants them
  static ImageState carve(void* buf, size_t N, size_t tiles, size_t& bytes) {
    Carver c(buf);
    ImageState s;
    s.final_T = c.take<float>(N);
    s.n_contrib = c.take<uint32_t>(N);
    s.ranges = c.take<uint32_t>(2 * tiles);
    s.tile_work = c.take<uint32_t>(2 * tiles);
    s.tile_order = c.take<uint32_t>(tiles);
    s.live_rows = c.take<uint32_t>(4);
    bytes = c.total();
    return s;
  }
};

struct BinningState {
  uint32_t* key_a;       // [R] tile id per instance in emission order (ping)
  uint32_t* key_b;       // [R] (pong)
  uint32_t* val_b;       // [R] emission index after pass 1
  uint32_t* src;         // [R] sorted position -> emission index (final result of the tile sort)
  uint32_t* inst_gid;    // [R] emission index -> Gaussian id
  uint8_t* flags;        // [R] emission index -> bit w: 64-pixel slot w of the tile blended this instance
  uint32_t* radix_table;   // [256 * sort_blocks(R)]
  uint32_t* scan_partials; // [scan_blocks(max(table, R)) + 1]
  uint32_t* rowbase;     // [R + 1] emission index -> first compact row of the instance (backward)
  // ---- words the fused kernels synchronise through; zeroed by the emission at the start of every forward
  uint32_t* sync_words;
  size_t sync_count;
  uint32_t* tickets;     // [16]: 0-3 tile-sort passes, 8 row compaction, 9 its done counter
  uint32_t* tile_hist;   // [4][256] digit totals of the tile-sort passes
  uint32_t* tile_status; // [4][blocks][256]
  uint32_t* row_status;  // [ceil((R + 1) / ROWS_CHUNK)] x 64 bit: per-block live-row totals of the row compaction
  static BinningState carve(void* buf, size_t R, size_t& bytes) {
    Carver c(buf);
    BinningState b;
    b.key_a = c.take<uint32_t>(R);
    b.key_b = c.take<uint32_t>(R);
    b.val_b = c.take<uint32_t>(R);
    b.src = c.take<uint32_t>(R);
    b.inst_gid = c.take<uint32_t>(R);
    b.flags = c.take<uint8_t>((R + 15) / 16 * 16);
    const size_t table = 256 * (size_t)sort_blocks((long long)R);
    b.radix_table = c.take<uint32_t>(table);
    b.scan_partials = c.take<uint32_t>((size_t)scan_blocks((long long)(table > R ? table : R)) + 1);
    b.rowbase = c.take<uint32_t>(R + 1);
    {
      const size_t st_words = fused_status_words((long long)R, 4);
      const size_t row_blocks = 2 * ((R + 1 + ROWS_CHUNK - 1) / ROWS_CHUNK + 1);  // 64-bit look-back words
      b.sync_count = 16 + 4 * 256 + st_words + row_blocks;
      b.sync_count = (b.sync_count + 3) / 4 * 4;
      b.sync_words = c.take<uint32_t>(b.sync_count);
      b.tickets = b.sync_words;
      b.tile_hist = b.tickets + 16;
      b.row_status = b.tile_hist + 4 * 256;
      b.tile_status = b.row_status + row_blocks;  // last: only the part a frame's passes use is zeroed
    }
    bytes = c.total();
    return b;
  }
};

}  // namespace olsr
