// olsr_api.hip — the C-ABI of include/olsr.h: argument checking, state-buffer carving and
// the launch sequence of the forward and backward passes.
//
// Counterpart of CR/rasterizer_impl.cu:216-756 (orchestration) and of the torch glue in
// DGR/rasterize_points.cu (allocation), minus everything torch: the caller owns memory.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/olsr.h"
#include "olsr_device.h"
#include "olsr_kernels.h"

#include <atomic>
#include <chrono>
#include <mutex>
#include "olsr_state.h"

using namespace olsr;

namespace {

thread_local std::string g_err;
thread_local bool g_profiling = false;
struct StageMark {
  const char* name;
  hipEvent_t ev;
};
thread_local std::vector<StageMark> g_marks;
thread_local hipStream_t g_mark_stream = nullptr;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return fail(OLSR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));         \
  } while (0)

void marks_reset() {
  for (auto& m : g_marks) (void)hipEventDestroy(m.ev);
  g_marks.clear();
}
void mark(const char* name, hipStream_t st) {
  if (!g_profiling || g_marks.size() >= (1u << 16)) return;
  g_mark_stream = st;
  StageMark m{name, nullptr};
  if (hipEventCreate(&m.ev) != hipSuccess) return;
  (void)hipEventRecord(m.ev, st);
  g_marks.push_back(m);
}

// CHECK_CUDA(A, debug), CR/auxiliary.h:166-173
int stage_check(const olsr_scene& s, const char* stage, hipStream_t st) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(OLSR_ERR_DEVICE, std::string(stage) + " launch: " + hipGetErrorString(e));
  if (s.debug) {
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(OLSR_ERR_DEVICE, std::string(stage) + ": " + hipGetErrorString(e));
  }
  return OLSR_OK;
}
#define STAGE(name)                                       \
  do {                                                    \
    int _rc = stage_check(s, name, st);                   \
    if (_rc != OLSR_OK) return _rc;                       \
    mark(name, st);                                       \
  } while (0)

bool supported_F(int F) { return F == 0 || F == 3 || F == 15 || F == 16 || F == 32; }

int check_scene(const olsr_scene* s, bool backward) {
  if (!s) return fail(OLSR_ERR_ARG, "scene is NULL");
  if (s->P < 0) return fail(OLSR_ERR_ARG, "P must be >= 0");
  if (s->width <= 0 || s->height <= 0) return fail(OLSR_ERR_ARG, "image size must be positive");
  // (tile coordinates travel as 16-bit fields of the emission record)
  if (s->width > 65535 * 15 || s->height > 65535 * 15) return fail(OLSR_ERR_ARG, "image size beyond 65535 tiles");
  if (s->tile != 15 && s->tile != 16) return fail(OLSR_ERR_ARG, "tile must be 15 or 16");
  if (!backward && s->binning != OLSR_BINNING_RECT && s->binning != OLSR_BINNING_ELLIPSE)
    return fail(OLSR_ERR_ARG, "binning must be OLSR_BINNING_RECT or OLSR_BINNING_ELLIPSE");
  if (!supported_F(s->F)) return fail(OLSR_ERR_ARG, "F (language channels) must be one of 0, 3, 15, 16, 32");
  if (s->D < 0 || s->D > 3) return fail(OLSR_ERR_ARG, "SH degree must be 0..3");
  if (s->P == 0) return OLSR_OK;
  if (!s->means3D || !s->background || !s->viewmatrix || !s->projmatrix || !s->cam_pos)
    return fail(OLSR_ERR_ARG, "means3D, background, viewmatrix, projmatrix and cam_pos are required");
  if (!backward && !s->opacities) return fail(OLSR_ERR_ARG, "opacities are required");
  if (s->flags & ~(OLSR_FLAG_SIGNED_EMPTY_RADII | OLSR_FLAG_FWD_ACCUM_MFMA | OLSR_FLAG_FWD_ACCUM_WEIGHT | OLSR_FLAG_FRAMES_IN_FLIGHT)) return fail(OLSR_ERR_ARG, "flags holds unknown OLSR_FLAG_* bits");
  if (s->activations & ~(OLSR_ACT_OPACITY_SIGMOID | OLSR_ACT_SCALE_EXP | OLSR_ACT_ROTATION_NORMALIZE))
    return fail(OLSR_ERR_ARG, "activations holds unknown OLSR_ACT_* bits");
  if (backward && (s->activations & OLSR_ACT_OPACITY_SIGMOID) && !s->opacities)
    return fail(OLSR_ERR_ARG, "a raw (pre-sigmoid) opacity is needed by backward as well");
  if ((s->activations & (OLSR_ACT_SCALE_EXP | OLSR_ACT_ROTATION_NORMALIZE)) && s->cov3D_precomp)
    return fail(OLSR_ERR_ARG, "scale / rotation activations make no sense with a precomputed 3D covariance");
  if ((s->shs == nullptr) == (s->colors_precomp == nullptr))
    return fail(OLSR_ERR_ARG, "Please provide excatly one of either SHs or precomputed colors!");
  const bool has_sr = s->scales != nullptr && s->rotations != nullptr;
  const bool any_sr = s->scales != nullptr || s->rotations != nullptr;
  if ((!has_sr && !s->cov3D_precomp) || (any_sr && s->cov3D_precomp))
    return fail(OLSR_ERR_ARG, "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
  if (s->shs && (s->M < (s->D + 1) * (s->D + 1))) return fail(OLSR_ERR_ARG, "shs holds fewer coefficients than the degree needs");
  if (s->F > 0 && !s->language_precomp) return fail(OLSR_ERR_ARG, "language_precomp is required when F > 0");
  if (backward && !s->projmatrix_raw) return fail(OLSR_ERR_ARG, "projmatrix_raw is required by backward");
  if (s->backward_row_capacity < 0) return fail(OLSR_ERR_ARG, "backward_row_capacity must be >= 0");
  if (s->backward_row_capacity > 0 && s->bwd_mode != OLSR_BWD_REFERENCE && s->bwd_mode != OLSR_BWD_EXACT)
    return fail(OLSR_ERR_ARG, "backward_row_capacity needs a valid bwd_mode in the forward as well");
  return OLSR_OK;
}

FrameDims frame_dims(const olsr_scene& s) {
  FrameDims d;
  d.W = s.width;
  d.H = s.height;
  d.tile = s.tile;
  d.gx = (s.width + s.tile - 1) / s.tile;
  d.gy = (s.height + s.tile - 1) / s.tile;
  d.ntiles = d.gx * d.gy;
  d.focal_y = s.height / (2.0f * s.tan_fovy);  // CR/rasterizer_impl.cu:394-395
  d.focal_x = s.width / (2.0f * s.tan_fovx);
  return d;
}

int tile_bits(int ntiles) {
  int bits = 1;
  while ((1LL << bits) < (long long)ntiles) ++bits;
  return bits;
}
// where the tile sort leaves its result: 0 = (key_a, src), 1 = (key_b, val_b)
int tile_sort_where(int ntiles) { return ((tile_bits(ntiles) + 7) / 8) & 1; }

struct BinningProvider {
  olsr_alloc_fn fn = nullptr;
  void* user = nullptr;
  void* fixed = nullptr;
  int64_t capacity = -1;  // async mode when >= 0
};

// forces the multi-kernel radix passes (the fallback of sorts too large for the fused ones), so that both paths can be
// tested at any size
bool force_legacy_sort() { return sort_knobs().legacy.load(std::memory_order_relaxed) != 0; }

// The drop-in (synchronising) entry needs the instance count on the host (the reference's blocking D2H,
// CR/rasterizer_impl.cu:454-455).  No copy, no event: block 0 of the histogram kernel stores the count and then a
// sequence number into two words of mapped, coherent host memory with system-scope stores ("the data is the flag"),
// and the host polls the second word — while the GPU goes straight on to the depth sort.  (A hipMemcpyAsync in the
// stream cost a 4 us copy kernel and a 6 us bubble in front of the first radix pass.)
// (the two words live until the process ends: a thread_local destructor could run after the HIP runtime's teardown)
struct PinnedCount {
  int32_t* p = nullptr;   // host view: {count, sequence}
  int32_t* dp = nullptr;  // device view of the same two words
  int32_t seq = 0;
};
thread_local PinnedCount g_pinned;

// The drop-in backward sizes its row scratch from the frame's exact gradient-row count without a synchronisation: the
// forward's last kernel posts {rows (4 slots), rows (packed survivor waves), sequence} into one slot of a ring of mapped
// host words; the matching backward — which runs after the caller's loss, long after the forward finished — looks its
// token up (olsr_live_rows) and falls back to the bound when the slot is not there (yet, or any more).
constexpr int ROWS_RING = 256;
// (process-wide: PyTorch runs a backward on its autograd thread, not on the thread that called the forward)
struct RowsRing {
  std::atomic<int32_t*> p{nullptr};  // host view: [ROWS_RING][4]
  int32_t* dp = nullptr;             // device view
  std::atomic<int32_t> seq{0};
  std::mutex init;
};
RowsRing g_rows;
thread_local int32_t g_last_token = 0;  // token of the last olsr_forward of this thread (0: none)
thread_local RowsMailbox g_rows_call;

// (per-view tile orders of the synchronising entry: described at order_hint_of below)
using olsr::HINT_HDR;
using olsr::HINT_SLOTS;
__global__ void hint_init_kernel(uint32_t* base, int ntiles) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)HINT_HDR + (size_t)HINT_SLOTS * (size_t)ntiles;
  if (i >= total) return;
  if (i < 4 + HINT_SLOTS) base[i] = 0u;
  else if (i < HINT_HDR) base[i] = 0x7FC00000u;  // NaN: matches no view
  else base[i] = (uint32_t)((i - HINT_HDR) % (size_t)ntiles);
}
// (the slot is picked by block 0 of the depth sort's histogram kernel — hint_pick_wave, k_sort.hip: a launch of its own, one
//  wave, cost the synchronising entry 5 us per frame on its dependent chain)
// A synchronisation error (olsr_state.h, counters[8]) is detected on the device after the call that caused it has returned.
// The sync-free entries report it through their status words; for the reference-shaped, synchronising API the last kernel of
// a forward / backward also raises a flag in mapped host memory, and the FIRST library call after the GPU got there fails
// with OLSR_ERR_DEVICE — the way an asynchronous HIP error surfaces.  Returns true (and clears the flag) if one is pending.
// Round 6: the flag is keyed by (device, stream) — one of STICKY_SLOTS mapped host
// words behind the rows ring, handed out in first-come order under a mutex; a call looks at the word of ITS device and stream
// only, so the error of a frame issued on one stream no longer fails an unrelated call on another.  When more (device,
// stream) pairs than slots have been seen, the last slot is shared by the rest (attribution degrades to "one of those").
constexpr int STICKY_SLOTS = 64;
struct StickyKeys {
  std::mutex m;
  int n = 0;
  int dev[STICKY_SLOTS];
  hipStream_t st[STICKY_SLOTS];
} g_sticky_keys;
int sticky_slot_of(hipStream_t st) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_sticky_keys.m);
  for (int i = 0; i < g_sticky_keys.n; ++i)
    if (g_sticky_keys.dev[i] == dev && g_sticky_keys.st[i] == st) return i;
  if (g_sticky_keys.n < STICKY_SLOTS - 1) {
    g_sticky_keys.dev[g_sticky_keys.n] = dev;
    g_sticky_keys.st[g_sticky_keys.n] = st;
    return g_sticky_keys.n++;
  }
  return STICKY_SLOTS - 1;
}
bool take_sticky_sync_error(hipStream_t st) {
  int32_t* ring = g_rows.p.load(std::memory_order_acquire);
  if (!ring) return false;
  return __atomic_exchange_n(&ring[4 * ROWS_RING + sticky_slot_of(st)], 0, __ATOMIC_ACQ_REL) != 0;
}
int32_t* sticky_sync_error_dev(hipStream_t st) {
  return g_rows.p.load(std::memory_order_acquire) ? g_rows.dp + 4 * ROWS_RING + sticky_slot_of(st) : nullptr;
}
const char* const STICKY_MSG =
    "device-side synchronisation error in an earlier frame issued on this device and stream: a look-back of its radix sort / "
    "row compaction never received a predecessor's counts (state buffer corrupted mid-frame?); that frame's images are "
    "invalid and its gradients are zeros.  (Reported like an asynchronous HIP error, by the first reference-shaped call on "
    "the same device and stream after the GPU got there; callers that need the error attributed to a frame use the sync-free "
    "entries, whose status words are per frame.)";

// Diagnostic (olsr_debug_composite_stamps): one-thread kernels in front of and behind every composite launch write the
// device's wall clock (100 MHz, common to all XCDs) into a caller's buffer — when did each composite become eligible, when had
// it finished, on which stream — so that the overlap of several frames in flight can be read without a profiler in the way.
struct StampState {
  unsigned long long* buf = nullptr;
  int capacity = 0;
  std::atomic<int> next{0};
} g_stamps;
__global__ void stamp_kernel(unsigned long long* slot, unsigned long long tag) {
  slot[0] = wall_clock64();
  slot[1] = tag;
}
void stamp(hipStream_t st, int kind) {
  if (!g_stamps.buf) return;
  const int i = g_stamps.next.fetch_add(1);
  if (i >= g_stamps.capacity) return;
  stamp_kernel<<<1, 1, 0, st>>>(g_stamps.buf + 2 * (size_t)i, ((unsigned long long)(uintptr_t)st << 8) | (unsigned)kind);
}

int forward_impl(const olsr_scene& s, void* geom_buf, void* img_buf, const BinningProvider& bp, float* out_color,
                 float* out_language, float* out_depth, float* out_opacity, int32_t* radii, int32_t* n_touched,
                 int32_t* num_rendered_host, int32_t* num_rendered_dev, uint32_t* tile_order_inout, hipStream_t st,
                 const olsr_loss_fusion* loss = nullptr, uint32_t* view_hints = nullptr) {
  const FrameDims d = frame_dims(s);
  const size_t N = (size_t)d.W * d.H;
  size_t gb, ib, bb;
  const GeometryState g = GeometryState::carve(geom_buf, (size_t)s.P, grad_row(s.F), gb);
  const ImageState im = ImageState::carve(img_buf, N, (size_t)d.ntiles, ib);
  mark("begin", st);

  if (!(loss && loss->skip_images) && (!out_color || !out_depth || !out_opacity || (s.F > 0 && !out_language)))
    return fail(OLSR_ERR_ARG, "output image pointers must not be NULL");
  if (s.P > 0 && (!radii || !n_touched)) return fail(OLSR_ERR_ARG, "radii and n_touched must not be NULL");

  if (s.P <= 0) {  // nothing below runs: leave a consistent empty state behind
    HIP_TRY(hipMemsetAsync(g.counters, 0, sizeof(int32_t) * 16, st));
    HIP_TRY(hipMemsetAsync(im.ranges, 0, sizeof(uint32_t) * 2 * (size_t)d.ntiles, st));
    HIP_TRY(hipMemsetAsync(im.live_rows, 0, sizeof(uint32_t) * 4, st));
  }

  const bool legacy = force_legacy_sort();
  const bool sync_mode = bp.capacity < 0;
  int64_t n_host = 0;
  BinningState b{};
  uint32_t* carry = nullptr;  // the caller's carried depth order, when this frame uses it
  uint32_t* order = nullptr;  // where the depth order lies when neither g.depth_order nor the carried array holds it
  const int32_t* n_order_dev = nullptr;  // its length on the device when the sort compacted its input (else: P)
  if (s.P > 0) {
    launch_preprocess(s, d, g, radii, n_touched, st);  // also zeroes the geometry buffer's synchronisation words
    STAGE("preprocess");
    // digit totals of the four depth passes in one read of the keys + the frame's counters (instances emitted,
    // the reference's num_rendered, overflow against the capacity), tile ranges reset to "empty"
    FusedHouse house{g.part_rect, g.part_count, (s.P + 255) / 256, sync_mode ? 0x7FFFFFFFLL : (long long)bp.capacity,
                     g.counters, num_rendered_dev, im.ranges, 2 * d.ntiles, nullptr, 0, im.live_rows,
                     view_hints, s.viewmatrix};  // (the synchronising entry: this frame's slot among the stream's per-view orders)
    if (sync_mode) {
      if (!g_pinned.p) {
        HIP_TRY(hipHostMalloc((void**)&g_pinned.p, 2 * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(hipHostGetDevicePointer((void**)&g_pinned.dp, g_pinned.p, 0));
        g_pinned.p[0] = 0;
        g_pinned.p[1] = 0;
      }
      g_pinned.seq = (g_pinned.seq == 0x7FFFFFFF) ? 1 : g_pinned.seq + 1;
      house.host_mailbox = g_pinned.dp;
      house.host_seq = g_pinned.seq;
    }
    SortBuffers sb{g.key_a, g.key_b, g.val_a, g.val_b, g.radix_table, g.scan_partials};
    const bool in_flight = (s.flags & OLSR_FLAG_FRAMES_IN_FLIGHT) != 0;
    // Carried depth order (include/olsr.h): two launches of independent workgroups repair the order the previous frame of this
    // view left in the caller's array; the sort below is enqueued behind them in any case and does its work only when the
    // repair says it could not prove the result (carry_miss != 0) — the lists never depend on what the array held.
    carry = (!sync_mode && !legacy && fused_sort_applicable(s.P, 32)) ? s.depth_order_carry : nullptr;
    const uint32_t* run_if = carry ? g.carry_miss : nullptr;
    if (carry)
      launch_order_repair(s.P, carry, g.key_a, g.key_b, g.val_b, g.tiles_touched, g.carry_totals, g.carry_miss, in_flight, st);
    if (!legacy && small_depth_sort_applicable(s.P, carry != nullptr)) {
      // at most 8 192 Gaussians: histogram, bookkeeping and all four passes in ONE launch of one workgroup (k_sort.hip)
      launch_small_depth_sort(g.key_a, s.P, carry ? carry : g.depth_order, g.tiles_touched, g.emit_status, &house, run_if, st);
      STAGE("depth_sort");
    } else {
    const SortPlan depth_plan = sort_plan(s.P, false, 85, in_flight);
    const bool fused_depth = !legacy && fused_sort_applicable(s.P, 32);
    // Visible-set compaction (k_sort.hip): without a carried order — whose domain is every Gaussian — the histogram kernel
    // hands the passes only the Gaussians that emit instances, (key, index) densely in key_b / val_b; the order they leave,
    // n_order = counters[12] entries long, goes to the backward's (idle) per-Gaussian scratch: the emission uses the sort's
    // own buffers for its rank lists.
    const bool compacted = fused_depth && carry == nullptr && depth_sort_compaction_applicable(s.P);
    SortCompaction cmp{g.tiles_touched, g.part_vis, g.key_b, g.val_b, &g.counters[12]};
    launch_sort_hist(g.key_a, s.P, nullptr, 32, g.sort_hist, &house, depth_plan.threads, st, run_if, compacted ? &cmp : nullptr);
    // (the last pass also leaves the instance total of every block of 1024 depth ranks behind, for the emission)
    if (compacted) {
      SortBuffers sbc{g.key_b, g.key_a, g.val_b, g.val_a, g.radix_table, g.scan_partials};
      order = reinterpret_cast<uint32_t*>(g.gacc);
      n_order_dev = &g.counters[12];
      launch_sort_fused(sbc, depth_plan, s.P, n_order_dev, 32, false, g.sort_hist, g.sort_status, g.tickets, nullptr, nullptr,
                        g.tiles_touched, g.emit_status, &g.counters[8],
                        sort_knobs().fault.load(std::memory_order_relaxed) & 1, st, order, nullptr);
    } else if (fused_depth) {
      // (values = Gaussian indices: the first pass takes them from the position, preprocess writes no index array)
      launch_sort_fused(sb, depth_plan, s.P, nullptr, 32, true, g.sort_hist, g.sort_status, g.tickets, nullptr, nullptr,
                        g.tiles_touched, g.emit_status, &g.counters[8],
                        sort_knobs().fault.load(std::memory_order_relaxed) & 1, st, carry, run_if);
    } else {
      launch_radix_sort(sb, s.P, nullptr, 32, true, st);
      launch_emit_totals(g.depth_order, s.P, g.tiles_touched, g.emit_status, st);
    }
    STAGE("depth_sort");
    }
  }

  void* bin_buf = nullptr;
  if (!sync_mode) {
    n_host = bp.capacity;
    bin_buf = bp.fixed;
  } else {
    int32_t R = 0;
    if (s.P > 0) {
      // the reference's blocking D2H (CR/rasterizer_impl.cu:454-455) — but the count was ready before the depth sort,
      // which keeps running while the host waits here and allocates
      volatile int32_t* box = g_pinned.p;
      const auto t0 = std::chrono::steady_clock::now();
      for (uint32_t spin = 0; __atomic_load_n(&box[1], __ATOMIC_ACQUIRE) != g_pinned.seq; ++spin) {
        __builtin_ia32_pause();
        if ((spin & 0xFFFFu) == 0xFFFFu) {  // every ~65 k polls: did the stream die, or is this taking seconds?
          const hipError_t q = hipStreamQuery(st);
          if (q != hipSuccess && q != hipErrorNotReady) return fail(OLSR_ERR_DEVICE, hipGetErrorString(q));
          if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
            // the frame's kernels are still enqueued and write into buffers the caller frees on error: drain first
            (void)hipStreamSynchronize(st);
            if (__atomic_load_n(&box[1], __ATOMIC_ACQUIRE) == g_pinned.seq) break;  // (it was only slow)
            return fail(OLSR_ERR_DEVICE, "the instance count never arrived from the device");
          }
        }
      }
      R = box[0];
      if (R < 0) return fail(OLSR_ERR_CAPACITY, "instance count exceeds 2^31-1");
    }
    n_host = R;
    if (num_rendered_host) *num_rendered_host = R;
    const size_t need = olsr_binning_bytes(R, s.F);
    bin_buf = bp.fn ? bp.fn(bp.user, need) : nullptr;
    if (!bin_buf) return fail(OLSR_ERR_ALLOC, "binning allocation callback returned NULL");
  }
  b = BinningState::carve(bin_buf, (size_t)n_host, bb);
  const int32_t* n_dev = &g.counters[1];

  if (s.P > 0) {
    const int tbits = tile_bits(d.ntiles);
    int tpasses = 0;
    const int tdb = fused_sort_digit_bits(tbits, &tpasses);
    const bool fused_tiles = !legacy && fused_sort_applicable(n_host, tbits);
    // (with depth cut-offs a steady-state frame keeps a fraction of the instances its capacity was sized for — config 3: a
    //  quarter —, and a pass is bound by the latency of its fattest block: plan the chunks for that, the uncut first frame of a
    //  sequence pays a few rounds more)
    const bool cut_mode = !sync_mode && s.tile_depth_cut != nullptr && s.binning == OLSR_BINNING_ELLIPSE;
    const SortPlan tile_plan = sort_plan(n_host, /*n_is_capacity=*/!sync_mode, cut_mode ? 25 : 85,
                                         (s.flags & OLSR_FLAG_FRAMES_IN_FLIGHT) != 0);
    // synchronisation words of the binning buffer that this frame uses (zeroed by the emission kernel)
    const int64_t bin_sync_words =
        (b.tile_status - b.sync_words) +
        (fused_tiles ? ((int64_t)tpasses * tile_plan.nblk * ((int64_t)1 << tdb) + 1) / 2 : 0);  // 16-bit words
    launch_emit(s, d, g, b, bin_sync_words, n_host, carry ? carry : (order ? order : g.depth_order),
                carry ? g.carry_totals : nullptr, carry ? g.carry_miss : nullptr, n_order_dev, st);
    STAGE("emit");
    if (n_host > 0) {
      // arrange the value ping-pong so that the last pass always lands in b.src
      const bool odd = tile_sort_where(d.ntiles) != 0;
      SortBuffers sb{b.key_a, b.key_b, odd ? b.val_b : b.src, odd ? b.src : b.val_b, b.radix_table, b.scan_partials};
      if (fused_tiles) {
        launch_sort_hist(b.key_a, n_host, n_dev, tbits, b.tile_hist, nullptr, tile_plan.threads, st);
        // the first pass also clears the liveness flags, the last one derives the tile ranges
        launch_sort_fused(sb, tile_plan, n_host, n_dev, tbits, true, b.tile_hist, b.tile_status, b.tickets, b.flags, im.ranges,
                          nullptr, nullptr, &g.counters[8], sort_knobs().fault.load(std::memory_order_relaxed) & 2, st);
      } else {
        const int where = launch_radix_sort(sb, n_host, n_dev, tbits, true, st);
        launch_tile_ranges(where ? b.key_b : b.key_a, n_host, n_dev, im.ranges, b.flags, st);
      }
      STAGE("tile_sort");
    }
  }
  g_rows_call = RowsMailbox{};
  if (sync_mode) {  // the drop-in entry: post the frame's gradient-row counts for the matching backward
    if (!g_rows.p.load(std::memory_order_acquire)) {
      std::lock_guard<std::mutex> lk(g_rows.init);
      if (!g_rows.p.load(std::memory_order_relaxed)) {
        int32_t* hp = nullptr;
        // (+ STICKY_SLOTS words behind the ring: the sticky synchronisation-error flags, one per (device, stream) —
        //  take_sticky_sync_error)
        HIP_TRY(hipHostMalloc((void**)&hp, (ROWS_RING * 4 + STICKY_SLOTS) * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(hipHostGetDevicePointer((void**)&g_rows.dp, hp, 0));
        std::memset(hp, 0, (ROWS_RING * 4 + STICKY_SLOTS) * sizeof(int32_t));
        g_rows.p.store(hp, std::memory_order_release);
      }
    }
    int32_t tok = g_rows.seq.fetch_add(1) + 1;
    if (tok <= 0 || tok >= 0x7FFFFF00) {  // (wrap: tokens stay positive)
      g_rows.seq.store(1);
      tok = 1;
    }
    g_last_token = tok;
    g_rows_call.dev = g_rows.dp + 4 * (tok % ROWS_RING);
    g_rows_call.seq = tok;
    g_rows_call.sticky = g_rows.dp + 4 * ROWS_RING + sticky_slot_of(st);
  }
  if (view_hints != nullptr && s.P > 0) {  // (the synchronising entry; the slot was picked by the depth sort's histogram kernel)
    g_rows_call.hint_slot = view_hints;
    tile_order_inout = view_hints + HINT_HDR;
  }
  // the caller announced its backward's row capacity: the forward's last launch compacts the rows as well (include/olsr.h)
  if (s.backward_row_capacity > 0 && s.P > 0 && bin_buf != nullptr) g_rows_call.compact_rows_n = n_host;
  stamp(st, 0);
  launch_render_forward(s, d, g, b, im, out_color, out_language, out_depth, out_opacity, n_touched, tile_order_inout,
                        num_rendered_dev, (s.P > 0) ? loss : nullptr, st);
  stamp(st, 1);
  g_rows_call = RowsMailbox{};
  STAGE("render_forward");

  if (num_rendered_dev && s.P == 0) HIP_TRY(hipMemsetAsync(num_rendered_dev, 0, 2 * sizeof(int32_t), st));
  (void)gb;
  (void)ib;
  (void)bb;
  return OLSR_OK;
}

// ---- launch-order hint of the synchronising entry ---------------------------------------------------------------
// olsr_forward keeps no caller-owned state between calls, but the forward composite finishes ~10 % sooner when a frame's
// heavy tiles are started first (tile_order_inout of olsr_forward_async).  The library therefore keeps, per (device,
// stream, tile count), the order measured on the previous frame issued on that stream.  Forwards on one stream execute in
// order, so the array is a complete permutation whenever a kernel reads it; no result depends on its content (a hint from
// another view of another scene is merely a worse guess).  A few KB each, at most 64 of them (least recently used evicted),
// allocated stream-ordered; the survivors live until the process ends.
// Several orders per (device, stream, tile count), one per VIEW the stream has rendered (round 4): the reference's mapping
// loop renders its window of keyframes one after the other on one stream (utils/slam_backend.py:510-670), so "the previous
// frame's order" belongs to another view — and a stale order costs the forward composite 25-35 % with one frame in flight
// (0.17 -> 0.23 ms at config 3, scripts/probe/arc_views.py).  A one-wave kernel in front of the composite compares the frame's
// view matrix with the HINT_SLOTS stored ones and names the nearest slot (within HINT_VIEW_TOL per entry), else recycles the
// least recently used one; the composite reads that slot's order, the tile-order kernel writes this frame's order back into
// it.  Buffer (32-bit words): [0] chosen slot, [1] use counter, [4, 4 + S) last use of every slot, then S x 16 floats (view
// matrices, NaN = empty), then S x ntiles orders (each a permutation at all times: iota initially).
struct OrderHints {
  struct Entry {
    int dev;
    hipStream_t st;
    int ntiles;
    uint32_t* buf;
  };
  std::mutex m;
  std::vector<Entry> v;
  // evicted buffers, never freed while anything may still use them: a thread that fetched the pointer a
  // moment before the eviction still launches kernels on it, and the evicted stream may have a forward in flight.  They are
  // handed to the next new key of the same (device, tile count) — any content is a legal hint (the header's slot index is
  // always 0..15, whoever wrote it last; orders that are no permutations are ignored entry by entry) — and only when more
  // than 64 of them wait is the oldest freed, after a synchronisation of its device.
  std::vector<Entry> spare;
} g_order_hints;

// the hint buffer of (current device, st, ntiles): header + HINT_SLOTS orders (layout above); nullptr: run without a hint
uint32_t* order_hint_of(int ntiles, hipStream_t st) {
  int dev = 0;
  if (ntiles <= 0 || hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_order_hints.m);
  auto& v = g_order_hints.v;
  for (size_t i = 0; i < v.size(); ++i)
    if (v[i].dev == dev && v[i].st == st && v[i].ntiles == ntiles) {
      const OrderHints::Entry e = v[i];  // most recently used last (LRU order)
      v.erase(v.begin() + (long)i);
      v.push_back(e);
      return e.buf;
    }
  auto& spare = g_order_hints.spare;
  if (v.size() >= 64) {
    // a caller that keeps creating streams / resolutions: the least recently used hint is retired to the spare list (see
    // above), not freed — its pointer may be in another thread's hands, its stream may still be writing it
    spare.push_back(v.front());
    v.erase(v.begin());
    if (spare.size() > 64) {
      const OrderHints::Entry old = spare.front();
      spare.erase(spare.begin());
      int cur = dev;
      (void)hipSetDevice(old.dev);
      (void)hipDeviceSynchronize();  // nothing that could still touch it is in flight after this (rare by construction)
      (void)hipFree(old.buf);
      (void)hipSetDevice(cur);
    }
  }
  const size_t words = (size_t)HINT_HDR + (size_t)HINT_SLOTS * (size_t)ntiles;
  for (size_t i = 0; i < spare.size(); ++i)
    if (spare[i].dev == dev && spare[i].ntiles == ntiles) {
      uint32_t* reuse = spare[i].buf;
      spare.erase(spare.begin() + (long)i);
      hint_init_kernel<<<(unsigned)((words + 255) / 256), 256, 0, st>>>(reuse, ntiles);
      v.push_back({dev, st, ntiles, reuse});
      return reuse;
    }
  uint32_t* buf = nullptr;
  // stream-ordered allocation: no device synchronisation on the hot olsr_forward path the first time a key is seen
  if (hipMallocAsync((void**)&buf, sizeof(uint32_t) * words, st) != hipSuccess) {
    (void)hipGetLastError();
    if (hipMalloc((void**)&buf, sizeof(uint32_t) * words) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
  }
  hint_init_kernel<<<(unsigned)((words + 255) / 256), 256, 0, st>>>(buf, ntiles);
  v.push_back({dev, st, ntiles, buf});
  return buf;
}

}  // namespace

namespace olsr {
RowsMailbox& rows_mailbox_of_this_call() { return g_rows_call; }
// The radix passes' knobs (olsr_state.h): seeded from the environment once, when the library is loaded
SortKnobs& sort_knobs() {
  static SortKnobs k;
  return k;
}
}  // namespace olsr

namespace {
struct SortKnobsFromEnv {
  SortKnobsFromEnv() {
    auto num = [](const char* name) {
      const char* e = std::getenv(name);
      return e ? std::atoi(e) : 0;
    };
    sort_knobs().kpt = num("OLSR_SORT_KPT");
    sort_knobs().resident = num("OLSR_SORT_RESIDENT");
    sort_knobs().legacy = num("OLSR_SORT_LEGACY") == 1 ? 1 : 0;
    if (std::getenv("OLSR_SORT_SMALL")) sort_knobs().small_sort = num("OLSR_SORT_SMALL") != 0 ? 1 : 0;
    if (std::getenv("OLSR_SORT_COMPACT")) sort_knobs().compact = num("OLSR_SORT_COMPACT") != 0 ? 1 : 0;
    if (num("OLSR_SORT_THREADS") == 1024 || num("OLSR_SORT_THREADS") == 256) sort_knobs().threads = num("OLSR_SORT_THREADS");
  }
} g_sort_knobs_from_env;
}  // namespace

extern "C" {

int32_t olsr_last_forward_token(void) { return g_last_token; }

int64_t olsr_live_rows(int32_t token, int32_t packed_survivor_waves) {
  int32_t* ring = g_rows.p.load(std::memory_order_acquire);
  if (token <= 0 || !ring) return -1;
  const volatile int32_t* slot = ring + 4 * (token % ROWS_RING);
  if (__atomic_load_n(&slot[2], __ATOMIC_ACQUIRE) != token) return -1;
  const int64_t v = (int64_t)(uint32_t)slot[packed_survivor_waves ? 1 : 0];
  return (__atomic_load_n(&slot[2], __ATOMIC_ACQUIRE) == token) ? v : -1;  // (not overwritten meanwhile)
}

int32_t olsr_live_rows_overwritten(int32_t token) {
  int32_t* ring = g_rows.p.load(std::memory_order_acquire);
  if (token <= 0 || !ring) return 0;
  const int32_t seen = __atomic_load_n(&ring[4 * (token % ROWS_RING) + 2], __ATOMIC_ACQUIRE);
  return (seen > token && seen - token < (1 << 30)) ? 1 : 0;
}

int64_t olsr_live_rows_wait(int32_t token, int32_t packed_survivor_waves, int32_t timeout_us) {
  int64_t v = olsr_live_rows(token, packed_survivor_waves);
  int32_t* ring = g_rows.p.load(std::memory_order_acquire);
  if (v >= 0 || timeout_us <= 0 || token <= 0 || !ring) return v;
  const volatile int32_t* slot = ring + 4 * (token % ROWS_RING);
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t spin = 0;; ++spin) {
    const int32_t seen = __atomic_load_n(&slot[2], __ATOMIC_ACQUIRE);
    if (seen == token) return olsr_live_rows(token, packed_survivor_waves);
    // tokens grow by one per forward (wrapping to 1 at INT32_MAX): a larger one in the slot means ours was overwritten
    if (seen > token && seen - token < (1 << 30)) return -1;
    __builtin_ia32_pause();
    if ((spin & 0xFFu) == 0xFFu &&
        std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(timeout_us))
      return -1;
  }
}

int64_t olsr_backward_rows(int32_t token, int32_t packed_survivor_waves, int64_t num_rendered, int32_t F) {
  static const int32_t wait_us = [] {
    const char* e = std::getenv("OLSR_ROWS_WAIT_US");
    return e ? std::atoi(e) : 5000;
  }();
  const int64_t bound = (num_rendered > 0 ? num_rendered : 0) * (packed_survivor_waves ? 2 : 4);
  int64_t v = olsr_live_rows(token, packed_survivor_waves);
  if (v < 0 && olsr_backward_scratch_bytes(bound, F) > ((size_t)64 << 20))
    v = olsr_live_rows_wait(token, packed_survivor_waves, wait_us);
  return (v >= 0 && v <= bound) ? v : bound;
}

size_t olsr_geometry_bytes(int32_t P, int32_t F) {
  size_t bytes = 0;
  GeometryState::carve(nullptr, (size_t)(P > 0 ? P : 0), grad_row(supported_F(F) ? F : 0), bytes);
  return bytes;
}

size_t olsr_image_bytes(int32_t width, int32_t height, int32_t tile) {
  size_t bytes = 0;
  if (tile <= 0) tile = 15;
  const size_t tiles = (size_t)((width + tile - 1) / tile) * (size_t)((height + tile - 1) / tile);
  ImageState::carve(nullptr, (size_t)width * (size_t)height, tiles, bytes);
  return bytes;
}

size_t olsr_binning_bytes(int64_t num_rendered, int32_t F) {
  (void)F;
  size_t bytes = 0;
  BinningState::carve(nullptr, (size_t)(num_rendered > 0 ? num_rendered : 0), bytes);
  return bytes;
}

size_t olsr_backward_scratch_bytes(int64_t rows, int32_t F) {
  return align_up((size_t)(rows > 0 ? rows : 0) * (size_t)grad_row(supported_F(F) ? F : 0) * sizeof(float)) + 2 * ALIGN;
}

int olsr_forward(const olsr_scene* scene, olsr_alloc_fn geometry_alloc, void* geometry_user,
                 olsr_alloc_fn binning_alloc, void* binning_user, olsr_alloc_fn image_alloc, void* image_user,
                 float* out_color, float* out_language, float* out_depth, float* out_opacity, int32_t* radii,
                 int32_t* n_touched, int32_t* num_rendered, void* hip_stream) {
  int rc = check_scene(scene, false);
  if (rc != OLSR_OK) return rc;
  if (!geometry_alloc || !binning_alloc || !image_alloc) return fail(OLSR_ERR_ARG, "allocation callbacks are required");
  // (before the allocation callbacks run: a call that fails for an EARLIER frame's error must not have resized the caller's
  //  buffers — ADVICE round 4)
  if (take_sticky_sync_error((hipStream_t)hip_stream)) return fail(OLSR_ERR_DEVICE, STICKY_MSG);
  void* geom = geometry_alloc(geometry_user, olsr_geometry_bytes(scene->P, scene->F));
  if (!geom) return fail(OLSR_ERR_ALLOC, "geometry allocation callback returned NULL");
  void* img = image_alloc(image_user, olsr_image_bytes(scene->width, scene->height, scene->tile));
  if (!img) return fail(OLSR_ERR_ALLOC, "image allocation callback returned NULL");
  BinningProvider bp;
  bp.fn = binning_alloc;
  bp.user = binning_user;
  if (num_rendered) *num_rendered = 0;
  const int tile = scene->tile > 0 ? scene->tile : 15;
  const int ntiles = ((scene->width + tile - 1) / tile) * ((scene->height + tile - 1) / tile);
  olsr_scene sc = *scene;
  sc.tile_depth_cut = nullptr;  // (per-tile depth cut-offs belong to the sync-free entries: their verdict is a device status word)
  sc.depth_order_carry = nullptr;  // (a caller-owned array: the reference-shaped entry has no place for one)
  return forward_impl(sc, geom, img, bp, out_color, out_language, out_depth, out_opacity, radii, n_touched,
                      num_rendered, nullptr, nullptr, (hipStream_t)hip_stream, nullptr,
                      order_hint_of(ntiles, (hipStream_t)hip_stream));
}

int olsr_forward_async(const olsr_scene* scene, void* geometry_buffer, void* binning_buffer, int64_t capacity,
                       void* image_buffer, float* out_color, float* out_language, float* out_depth, float* out_opacity,
                       int32_t* radii, int32_t* n_touched, int32_t* num_rendered_dev, uint32_t* tile_order_inout,
                       void* hip_stream) {
  int rc = check_scene(scene, false);
  if (rc != OLSR_OK) return rc;
  if (!geometry_buffer || !binning_buffer || !image_buffer || capacity < 0)
    return fail(OLSR_ERR_ARG, "state buffers and a non-negative capacity are required");
  if (scene->tile_depth_cut && scene->binning == OLSR_BINNING_ELLIPSE &&
      (scene->flags & (OLSR_FLAG_FWD_ACCUM_MFMA | OLSR_FLAG_FWD_ACCUM_WEIGHT)))
    return fail(OLSR_ERR_ARG, "tile_depth_cut: only with the default forward accumulation");
  BinningProvider bp;
  bp.fixed = binning_buffer;
  bp.capacity = capacity;
  return forward_impl(*scene, geometry_buffer, image_buffer, bp, out_color, out_language, out_depth, out_opacity,
                      radii, n_touched, nullptr, num_rendered_dev, tile_order_inout, (hipStream_t)hip_stream);
}

size_t olsr_fused_loss_scratch_bytes(int32_t width, int32_t height, int32_t tile) {
  if (width <= 0 || height <= 0 || (tile != 15 && tile != 16)) return 0;
  const size_t tiles = (size_t)((width + tile - 1) / tile) * (size_t)((height + tile - 1) / tile);
  return align_up(tiles * 5 * sizeof(float)) + ALIGN;
}

int olsr_forward_async_loss(const olsr_scene* scene, void* geometry_buffer, void* binning_buffer, int64_t capacity,
                            void* image_buffer, float* out_color, float* out_language, float* out_depth,
                            float* out_opacity, int32_t* radii, int32_t* n_touched, int32_t* num_rendered_dev,
                            uint32_t* tile_order_inout, const olsr_loss_fusion* loss, void* hip_stream) {
  if (!loss)
    return olsr_forward_async(scene, geometry_buffer, binning_buffer, capacity, image_buffer, out_color, out_language,
                              out_depth, out_opacity, radii, n_touched, num_rendered_dev, tile_order_inout, hip_stream);
  int rc = check_scene(scene, false);
  if (rc != OLSR_OK) return rc;
  if (!geometry_buffer || !binning_buffer || !image_buffer || capacity < 0)
    return fail(OLSR_ERR_ARG, "state buffers and a non-negative capacity are required");
  const olsr_loss_params& lp = loss->params;
  if (lp.width != scene->width || lp.height != scene->height)
    return fail(OLSR_ERR_ARG, "fused loss: params.width / height must equal the scene's");
  if (scene->flags & (OLSR_FLAG_FWD_ACCUM_MFMA | OLSR_FLAG_FWD_ACCUM_WEIGHT))
    return fail(OLSR_ERR_ARG, "fused loss: only with the default forward accumulation");
  if (!loss->gt_image || !loss->gt_depth || !loss->dL_dimage || !loss->dL_ddepth || !loss->loss || !loss->scratch)
    return fail(OLSR_ERR_ARG, "fused loss: gt_image, gt_depth, dL_dimage, dL_ddepth, loss and scratch are required");
  const bool lang_term = !loss->tracking && lp.F > 0 && loss->gt_language != nullptr;
  if (lang_term && (lp.F != scene->F || !loss->dL_dlanguage || lp.lang_width <= 0 || lp.lang_height <= 0))
    return fail(OLSR_ERR_ARG, "fused loss: a language term needs params.F == scene F, dL_dlanguage and the target's size");
  if (scene->tile_depth_cut && scene->binning == OLSR_BINNING_ELLIPSE && !loss->tracking)
    return fail(OLSR_ERR_ARG, "tile_depth_cut: with a fused loss only for the tracking loss (a mapping step sums views; a view "
                              "whose frame missed would change the sum)");
  if (scene->P <= 0)  // nothing is rendered (the images are the background): the stand-alone kernels define this case
    return fail(OLSR_ERR_ARG, "fused loss: P must be > 0 (use olsr_mapping_loss / olsr_tracking_loss on an empty render)");
  olsr_loss_fusion lf = *loss;
  lf.scratch = (void*)(((uintptr_t)loss->scratch + ALIGN - 1) / ALIGN * ALIGN);
  BinningProvider bp;
  bp.fixed = binning_buffer;
  bp.capacity = capacity;
  return forward_impl(*scene, geometry_buffer, image_buffer, bp, out_color, out_language, out_depth, out_opacity, radii,
                      n_touched, nullptr, num_rendered_dev, tile_order_inout, (hipStream_t)hip_stream, &lf);
}

int olsr_backward(const olsr_scene* scene, const int32_t* radii, void* geometry_buffer, int32_t num_rendered,
                  void* binning_buffer, const void* image_buffer, olsr_alloc_fn scratch_alloc, void* scratch_user,
                  void* scratch, int64_t scratch_rows, const float* dL_dout_color, const float* dL_dout_language,
                  const float* dL_dout_depth, float* dL_dmeans2D, float* dL_dconic, float* dL_dopacity,
                  float* dL_dcolors, float* dL_dlanguage, float* dL_ddepths, float* dL_dmeans3D, float* dL_dcov3D,
                  float* dL_dsh, float* dL_dscales, float* dL_drotations, float* dL_dtau, float* dL_dtau_sum,
                  const olsr_grad_bucket* bucket, int32_t* status_dev, void* hip_stream) {
  int rc = check_scene(scene, true);
  if (rc != OLSR_OK) return rc;
  const olsr_scene& s = *scene;
  hipStream_t st = (hipStream_t)hip_stream;
  if (s.bwd_mode != OLSR_BWD_REFERENCE && s.bwd_mode != OLSR_BWD_EXACT)
    return fail(OLSR_ERR_ARG, "bwd_mode must be OLSR_BWD_REFERENCE or OLSR_BWD_EXACT");
  if (take_sticky_sync_error(st)) return fail(OLSR_ERR_DEVICE, STICKY_MSG);
  mark("begin", st);
  if (s.P == 0) {
    if (dL_dtau_sum) HIP_TRY(hipMemsetAsync(dL_dtau_sum, 0, 6 * sizeof(float), st));
    if (status_dev) HIP_TRY(hipMemsetAsync(status_dev, 0, 2 * sizeof(int32_t), st));
    return OLSR_OK;
  }
  if (!radii || !geometry_buffer || !binning_buffer || !image_buffer || num_rendered < 0)
    return fail(OLSR_ERR_ARG, "radii, the three state buffers and num_rendered (>= 0) are required");
  if (!scratch_alloc && (!scratch || scratch_rows < 0))
    return fail(OLSR_ERR_ARG, "either a scratch allocation callback or a scratch buffer with its row capacity is required");
  if (!dL_dout_color) return fail(OLSR_ERR_ARG, "dL_dout_color must not be NULL");
  // A NULL language / depth cotangent means "the loss does not depend on that image" (what autograd hands the
  // reference's backward as None -> zeros, DGR/diff_gaussian_rasterization/__init__.py:296-345; the tracking loss has
  // no language term, utils/slam_utils.py:92-121).  Without a language cotangent the RGB instantiation of the composite
  // backward runs on the language forward's state: D - A == 0 and the rank-0 language row == 0, so every gradient
  // equals what zero-filled cotangents give, and dL_dlanguage is written as zeros.
  const int F_rows = (s.F > 0 && !dL_dout_language) ? 0 : s.F;
  if (bucket) {
    if (!bucket->flat || !bucket->densify || !bucket->max_radii)
      return fail(OLSR_ERR_ARG, "bucket.flat, bucket.densify and bucket.max_radii must not be NULL");
  } else if (!dL_dtau_sum && (!dL_dmeans2D || !dL_dopacity || !dL_dcolors || (s.F > 0 && !dL_dlanguage) ||
                              !dL_dmeans3D || !dL_dcov3D || (s.M > 0 && !dL_dsh) || !dL_dscales || !dL_drotations ||
                              !dL_dtau)) {
    return fail(OLSR_ERR_ARG, "gradient outputs must not be NULL (unless a gradient bucket or dL_dtau_sum is given)");
  }
  const FrameDims d = frame_dims(s);
  size_t gb, ib, bb;
  const GeometryState g = GeometryState::carve(geometry_buffer, (size_t)s.P, grad_row(s.F), gb);
  const ImageState im = ImageState::carve(const_cast<void*>(image_buffer), (size_t)d.W * d.H, (size_t)d.ntiles, ib);
  BinningState b = BinningState::carve(binning_buffer, (size_t)num_rendered, bb);

  // compact the partial-gradient rows: one row per (instance, slot) pair the forward blended — unless the forward did
  // (olsr_scene.backward_row_capacity: rowbase, counters[6] and counters[7] are in place, the backward's last kernel
  // copies them to status_dev)
  const bool packed_ref15 = (s.bwd_mode == OLSR_BWD_REFERENCE && s.tile == 15);
  const bool rows_compacted = s.backward_row_capacity > 0;
  if (rows_compacted && (scratch_alloc || scratch_rows != s.backward_row_capacity))
    return fail(OLSR_ERR_ARG, "backward_row_capacity: the backward needs a caller-owned scratch of exactly that many rows");
  if (!rows_compacted) {
    launch_row_compaction(b.flags, num_rendered, &g.counters[1], packed_ref15, b.rowbase, b.row_status, b.tickets + 8,
                          scratch_alloc ? 0x7FFFFFFFLL : scratch_rows, g.counters, status_dev, st);
    STAGE("row_compaction");
  }
  if (scratch_alloc) {
    int32_t c3[3] = {0, 0, 0};  // {live rows, row / instance overflow, synchronisation error}
    HIP_TRY(hipMemcpyAsync(c3, &g.counters[6], 3 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (c3[2] != 0)  // (nothing of this frame's lists can be trusted; no gradient has been written)
      return fail(OLSR_ERR_DEVICE, "device-side synchronisation error: a look-back of this frame's radix sort / row "
                                   "compaction never received a predecessor's counts (state buffer corrupted mid-frame?)");
    const int32_t L = c3[0];
    scratch_rows = L;
    scratch = scratch_alloc(scratch_user, olsr_backward_scratch_bytes(L, s.F));
    if (!scratch) return fail(OLSR_ERR_ALLOC, "backward scratch allocation callback returned NULL");
  }
  float* rows = (float*)(((uintptr_t)scratch + ALIGN - 1) / ALIGN * ALIGN);

  stamp(st, 2);
  if (s.bwd_mode == OLSR_BWD_REFERENCE)
    launch_render_backward_reference(s, F_rows, d, g, b, im, dL_dout_color, dL_dout_language, dL_dout_depth, rows, st);
  else
    launch_render_backward_exact(s, F_rows, d, g, b, im, dL_dout_color, dL_dout_language, dL_dout_depth, rows, st);
  stamp(st, 3);
  STAGE("render_backward");
  GradOut o{dL_dmeans2D, dL_dconic, dL_dopacity, dL_dcolors, dL_dlanguage, dL_ddepths, dL_dmeans3D,
            dL_dcov3D,   dL_dsh,    dL_dscales,  dL_drotations, dL_dtau,    dL_dtau_sum};
  if (bucket) {
    o.bucket_flat = bucket->flat;
    o.bucket_densify = bucket->densify;
    o.bucket_max_radii = bucket->max_radii;
    o.bucket_assign = bucket->assign;
    o.bucket_row_mask = reinterpret_cast<unsigned long long*>(bucket->row_mask);
  }
  o.status_dev = status_dev;
  o.status_rows = rows_compacted;
  // (a caller that passes status_dev reads the report there; one that does not — the reference-shaped bindings — gets it
  //  from the library's next call)
  o.sticky_error = status_dev ? nullptr : sticky_sync_error_dev(st);
  launch_preprocess_backward(s, F_rows, d, g, b, rows, radii, o, g.tau_partials, st);
  STAGE("preprocess_backward");
  (void)gb;
  (void)ib;
  (void)bb;
  return OLSR_OK;
}

int olsr_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                      uint8_t* present, void* hip_stream) {
  (void)projmatrix;  // the reference computes p_hom and discards it (CR/auxiliary.h:149-151)
  if (P < 0) return fail(OLSR_ERR_ARG, "P must be >= 0");
  if (P == 0) return OLSR_OK;
  if (!means3D || !viewmatrix || !present) return fail(OLSR_ERR_ARG, "means3D, viewmatrix and present are required");
  launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)hip_stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(OLSR_ERR_DEVICE, std::string("mark_visible launch: ") + hipGetErrorString(e));
  return OLSR_OK;
}

int olsr_adam_step(int32_t P, int32_t M, int32_t F, const olsr_adam_params* params, const float* flat, float* means3D,
                   float* shs, float* opacities, float* scales, float* rotations, float* language, float* exp_avg,
                   float* exp_avg_sq, void* hip_stream) {
  if (P < 0 || M < 0 || !supported_F(F)) return fail(OLSR_ERR_ARG, "P, M must be >= 0 and F one of 0, 3, 15, 16, 32");
  if (!params || params->step < 1) return fail(OLSR_ERR_ARG, "adam params are required and step must be >= 1");
  if (P == 0) return OLSR_OK;
  if (!flat || !means3D || !opacities || !scales || !rotations || !exp_avg || !exp_avg_sq || (M > 0 && !shs) ||
      (F > 0 && !language))
    return fail(OLSR_ERR_ARG, "the bucket, every parameter array and both moment buffers are required");
  const float* one[1] = {flat};
  launch_adam_step(P, M, F, *params, one, nullptr, 1, means3D, shs, opacities, scales, rotations, language, exp_avg, exp_avg_sq,
                   (hipStream_t)hip_stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(OLSR_ERR_DEVICE, std::string("adam_step launch: ") + hipGetErrorString(e));
  return OLSR_OK;
}

int olsr_adam_step_sum(int32_t P, int32_t M, int32_t F, const olsr_adam_params* params, int32_t n_flats,
                       const float* const* flats, float* means3D, float* shs, float* opacities, float* scales,
                       float* rotations, float* language, float* exp_avg, float* exp_avg_sq, void* hip_stream) {
  return olsr_adam_step_masked(P, M, F, params, n_flats, flats, nullptr, means3D, shs, opacities, scales, rotations, language,
                               exp_avg, exp_avg_sq, hip_stream);
}

int olsr_adam_step_masked(int32_t P, int32_t M, int32_t F, const olsr_adam_params* params, int32_t n_flats,
                          const float* const* flats, const uint64_t* const* row_masks, float* means3D, float* shs,
                          float* opacities, float* scales, float* rotations, float* language, float* exp_avg,
                          float* exp_avg_sq, void* hip_stream) {
  if (P < 0 || M < 0 || !supported_F(F)) return fail(OLSR_ERR_ARG, "P, M must be >= 0 and F one of 0, 3, 15, 16, 32");
  if (!params || params->step < 1) return fail(OLSR_ERR_ARG, "adam params are required and step must be >= 1");
  if (n_flats < 1 || n_flats > OLSR_ADAM_MAX_BUCKETS || !flats)
    return fail(OLSR_ERR_ARG, "adam_step_sum: between 1 and 8 gradient buckets");
  for (int b = 0; b < n_flats; ++b)
    if (!flats[b]) return fail(OLSR_ERR_ARG, "adam_step_sum: a gradient bucket is NULL");
  if (P == 0) return OLSR_OK;
  if (!means3D || !opacities || !scales || !rotations || !exp_avg || !exp_avg_sq || (M > 0 && !shs) ||
      (F > 0 && !language))
    return fail(OLSR_ERR_ARG, "every parameter array and both moment buffers are required");
  launch_adam_step(P, M, F, *params, flats, reinterpret_cast<const unsigned long long* const*>(row_masks), n_flats, means3D,
                   shs, opacities, scales, rotations, language, exp_avg, exp_avg_sq, (hipStream_t)hip_stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(OLSR_ERR_DEVICE, std::string("adam_step_sum launch: ") + hipGetErrorString(e));
  return OLSR_OK;
}

int olsr_pose_step(const olsr_pose_params* params, const float* dL_dtau_sum, const float* dL_dexposure,
                   const float* projection_matrix, float* state, int32_t* status, void* hip_stream) {
  return olsr_pose_step_gated(params, dL_dtau_sum, dL_dexposure, projection_matrix, state, status, nullptr, hip_stream);
}

int olsr_pose_step_gated(const olsr_pose_params* params, const float* dL_dtau_sum, const float* dL_dexposure,
                         const float* projection_matrix, float* state, int32_t* status, const int32_t* frame_status,
                         void* hip_stream) {
  if (!params || !projection_matrix || !state || !status)
    return fail(OLSR_ERR_ARG, "pose params, projection_matrix, state and status are required");
  // (step <= 0 with a gradient: the step count is status[1] + 1, kept on the device — see include/olsr.h)
  if (!dL_dtau_sum && dL_dexposure) return fail(OLSR_ERR_ARG, "an exposure gradient needs a pose gradient (one optimiser step)");
  launch_pose_step(*params, dL_dtau_sum, dL_dexposure, projection_matrix, state, status, frame_status,
                   (hipStream_t)hip_stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(OLSR_ERR_DEVICE, std::string("pose_step launch: ") + hipGetErrorString(e));
  return OLSR_OK;
}

size_t olsr_knn_scratch_bytes(int32_t P) { return knn_scratch_bytes(P); }

int olsr_knn_mean_dist2(int32_t P, const float* points, float* mean_dist2, void* scratch, void* hip_stream) {
  if (P < 0) return fail(OLSR_ERR_ARG, "P must be >= 0");
  if (P == 0) return OLSR_OK;
  if (!points || !mean_dist2 || !scratch) return fail(OLSR_ERR_ARG, "points, mean_dist2 and scratch are required");
  launch_knn(P, points, mean_dist2, scratch, (hipStream_t)hip_stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(OLSR_ERR_DEVICE, std::string("knn launch: ") + hipGetErrorString(e));
  return OLSR_OK;
}

size_t olsr_mapping_loss_scratch_bytes(int32_t width, int32_t height) {
  if (width <= 0 || height <= 0) return ALIGN;
  return (size_t)loss_blocks(width, height) * 5 * sizeof(float) + ALIGN;
}

int olsr_mapping_loss(const olsr_loss_params* params, const float* image, const float* depth, const float* language,
                      const float* gt_image, const float* gt_depth, const float* gt_language, const float* exposure,
                      float* dL_dimage, float* dL_ddepth, float* dL_dlanguage, float* loss, float* dL_dexposure,
                      void* scratch, void* hip_stream) {
  if (!params) return fail(OLSR_ERR_ARG, "loss params are NULL");
  const olsr_loss_params& p = *params;
  if (p.width <= 0 || p.height <= 0) return fail(OLSR_ERR_ARG, "image size must be positive");
  if (!supported_F(p.F)) return fail(OLSR_ERR_ARG, "F (language channels) must be one of 0, 3, 15, 16, 32");
  if (!image || !depth || !gt_image || !gt_depth || !dL_dimage || !dL_ddepth || !loss || !scratch)
    return fail(OLSR_ERR_ARG, "image, depth, their targets, their gradient outputs, loss and scratch are required");
  if (p.F > 0 && (!language || !dL_dlanguage)) return fail(OLSR_ERR_ARG, "language and dL_dlanguage are required when F > 0");
  if (p.F > 0 && gt_language && (p.lang_width <= 0 || p.lang_height <= 0))
    return fail(OLSR_ERR_ARG, "the language target size must be positive");
  hipStream_t st = (hipStream_t)hip_stream;
  float* partials = (float*)(((uintptr_t)scratch + ALIGN - 1) / ALIGN * ALIGN);
  launch_mapping_loss(p, image, depth, language, gt_image, gt_depth, p.F > 0 ? gt_language : nullptr, exposure, nullptr,
                      nullptr, false, dL_dimage, dL_ddepth, dL_dlanguage, loss, dL_dexposure, partials, st);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(OLSR_ERR_DEVICE, std::string("mapping_loss launch: ") + hipGetErrorString(e));
  return OLSR_OK;
}

int olsr_tracking_loss(const olsr_loss_params* params, const float* image, const float* depth, const float* opacity,
                       const float* gt_image, const float* gt_depth, const float* grad_mask, const float* exposure,
                       float* dL_dimage, float* dL_ddepth, float* loss, float* dL_dexposure, void* scratch,
                       void* hip_stream) {
  if (!params) return fail(OLSR_ERR_ARG, "loss params are NULL");
  const olsr_loss_params& p = *params;
  if (p.width <= 0 || p.height <= 0) return fail(OLSR_ERR_ARG, "image size must be positive");
  if (!image || !depth || !opacity || !gt_image || !gt_depth || !dL_dimage || !dL_ddepth || !loss || !scratch)
    return fail(OLSR_ERR_ARG, "image, depth, opacity, their targets, the gradient outputs, loss and scratch are required");
  hipStream_t st = (hipStream_t)hip_stream;
  float* partials = (float*)(((uintptr_t)scratch + ALIGN - 1) / ALIGN * ALIGN);
  launch_mapping_loss(p, image, depth, nullptr, gt_image, gt_depth, nullptr, exposure, opacity, grad_mask, true, dL_dimage,
                      dL_ddepth, nullptr, loss, dL_dexposure, partials, st);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(OLSR_ERR_DEVICE, std::string("tracking_loss launch: ") + hipGetErrorString(e));
  return OLSR_OK;
}

int olsr_accumulate_gradients(int32_t P, int32_t M, int32_t F, int32_t assign, const float* dL_dmeans3D,
                              const float* dL_dsh,
                              const float* dL_dopacity, const float* dL_dscales, const float* dL_drotations,
                              const float* dL_dlanguage, const float* dL_dmeans2D, const int32_t* radii, float* flat,
                              float* densify, int32_t* max_radii, void* hip_stream) {
  if (P < 0 || M < 0 || F < 0) return fail(OLSR_ERR_ARG, "P, M, F must be >= 0");
  if (P == 0) return OLSR_OK;
  if (!dL_dmeans3D || !dL_dopacity || !dL_dscales || !dL_drotations || !dL_dmeans2D || !radii || !flat || !densify ||
      !max_radii || (M > 0 && !dL_dsh) || (F > 0 && !dL_dlanguage))
    return fail(OLSR_ERR_ARG, "gradient, radii and accumulator pointers must not be NULL");
  launch_accumulate(P, M, F, assign != 0, dL_dmeans3D, dL_dsh, dL_dopacity, dL_dscales, dL_drotations, dL_dlanguage, dL_dmeans2D,
                    radii, flat, densify, max_radii, (hipStream_t)hip_stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(OLSR_ERR_DEVICE, std::string("accumulate launch: ") + hipGetErrorString(e));
  return OLSR_OK;
}

int olsr_bucket_add(int32_t P, int32_t width, float* dst_flat, float* dst_densify, int32_t* dst_max_radii,
                    uint64_t* dst_row_mask, const float* src_flat, const float* src_densify, const int32_t* src_max_radii,
                    const uint64_t* src_row_mask, void* hip_stream) {
  if (P < 0 || width <= 0) return fail(OLSR_ERR_ARG, "P must be >= 0, width > 0");
  if (P == 0) return OLSR_OK;
  if (!dst_flat || !dst_densify || !dst_max_radii || !src_flat || !src_densify || !src_max_radii)
    return fail(OLSR_ERR_ARG, "bucket_add: flat, densify and max_radii of both buckets are required");
  launch_bucket_add(P, width, dst_flat, src_flat, reinterpret_cast<unsigned long long*>(dst_row_mask),
                    reinterpret_cast<const unsigned long long*>(src_row_mask), dst_densify, src_densify, dst_max_radii,
                    src_max_radii, (hipStream_t)hip_stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(OLSR_ERR_DEVICE, std::string("bucket_add launch: ") + hipGetErrorString(e));
  return OLSR_OK;
}

int olsr_sparse_exchange_mask(int32_t P, int32_t width, const float* flat, const uint64_t* row_mask, const int32_t* max_radii,
                              int32_t* imax, void* hip_stream) {
  if (P < 0 || width <= 0) return fail(OLSR_ERR_ARG, "P must be >= 0, width > 0");
  if (P == 0) return OLSR_OK;
  if (!flat || !max_radii || !imax) return fail(OLSR_ERR_ARG, "flat, max_radii and imax must not be NULL");
  launch_exchange_mask(P, width, flat, reinterpret_cast<const unsigned long long*>(row_mask), max_radii, imax,
                       (hipStream_t)hip_stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(OLSR_ERR_DEVICE, std::string("sparse exchange (mask) launch: ") + hipGetErrorString(e));
  return OLSR_OK;
}

int64_t olsr_sparse_exchange_scratch_ints(int32_t P) { return P <= 0 ? 0 : ((int64_t)P + 1023) / 1024; }

int olsr_sparse_exchange_pack(int32_t P, int32_t width, int32_t capacity, const float* flat, const int32_t* imax,
                              int32_t* max_radii, uint64_t* row_mask, const float* densify, int32_t* idx, float* fsum,
                              int32_t* scratch, int32_t* status_dev, void* hip_stream) {
  if (P < 0 || width <= 0 || capacity <= 0) return fail(OLSR_ERR_ARG, "P must be >= 0, width and capacity > 0");
  if (P == 0) return OLSR_OK;
  if (!flat || !imax || !max_radii || !densify || !scratch || !status_dev || ((idx == nullptr) != (fsum == nullptr)))
    return fail(OLSR_ERR_ARG, "sparse exchange (pack): only row_mask may be NULL, or idx and fsum together (count only)");
  launch_exchange_pack(P, width, capacity, flat, imax, max_radii, reinterpret_cast<unsigned long long*>(row_mask), densify, idx,
                       fsum, scratch, status_dev, (hipStream_t)hip_stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(OLSR_ERR_DEVICE, std::string("sparse exchange (pack) launch: ") + hipGetErrorString(e));
  return OLSR_OK;
}

int olsr_sparse_exchange_unpack(int32_t P, int32_t width, int32_t capacity, const int32_t* idx, const float* fsum, float* flat,
                                float* densify, void* hip_stream) {
  if (P < 0 || width <= 0 || capacity <= 0) return fail(OLSR_ERR_ARG, "P must be >= 0, width and capacity > 0");
  if (P == 0) return OLSR_OK;
  if (!idx || !fsum || !flat || !densify) return fail(OLSR_ERR_ARG, "sparse exchange (unpack): pointers must not be NULL");
  launch_exchange_unpack(P, width, capacity, idx, fsum, flat, densify, (hipStream_t)hip_stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(OLSR_ERR_DEVICE, std::string("sparse exchange (unpack) launch: ") + hipGetErrorString(e));
  return OLSR_OK;
}

const void* olsr_geometry_field(const void* geometry_buffer, int32_t P, int32_t F, const char* name) {
  size_t bytes;
  const GeometryState g =
      GeometryState::carve(const_cast<void*>(geometry_buffer), (size_t)P, grad_row(supported_F(F) ? F : 0), bytes);
  if (!std::strcmp(name, "depths")) return g.depths;
  if (!std::strcmp(name, "means2D")) return g.means2D;
  if (!std::strcmp(name, "cov3D")) return g.cov3D;
  if (!std::strcmp(name, "conic_opacity")) return g.conic_opacity;
  if (!std::strcmp(name, "rgb")) return g.rgb;
  if (!std::strcmp(name, "clamped")) return g.clamped;
  if (!std::strcmp(name, "tiles_touched")) return g.tiles_touched;
  if (!std::strcmp(name, "depth_order")) return g.depth_order;  // (the sort's own buffer: every Gaussian, compaction off)
  if (!std::strcmp(name, "depth_order_compacted")) return g.gacc;  // u32[counters[12]]: the emitting Gaussians in depth order
  if (!std::strcmp(name, "counters")) return g.counters;
  if (!std::strcmp(name, "emit_totals")) return g.emit_status;
  if (!std::strcmp(name, "inst_start")) return g.inst_start;
  if (!std::strcmp(name, "blended")) return g.blended;  // u8[P]: some pixel blended the Gaussian in this frame's forward
  if (!std::strcmp(name, "carry_miss")) return g.carry_miss;  // u32: != 0 = this frame's carried depth order was not repairable
  if (!std::strcmp(name, "carry_totals")) return g.carry_totals;
  if (!std::strcmp(name, "sort_keys")) return g.key_a;  // u32[P] (valid after a forward whose carried order was repaired)
  return nullptr;
}

const void* olsr_binning_field(const void* binning_buffer, int64_t num_rendered, int32_t F, const char* name) {
  size_t bytes;
  (void)F;
  const BinningState b = BinningState::carve(const_cast<void*>(binning_buffer), (size_t)num_rendered, bytes);
  if (!std::strcmp(name, "inst_gid")) return b.inst_gid;
  if (!std::strcmp(name, "flags")) return b.flags;
  if (!std::strcmp(name, "rowbase")) return b.rowbase;
  if (!std::strcmp(name, "row_sync")) return b.tickets + 8;  // {ticket, finished blocks} of the row compaction
  if (!std::strcmp(name, "key_a")) return b.key_a;
  if (!std::strcmp(name, "key_b")) return b.key_b;
  if (!std::strcmp(name, "src")) return b.src;
  if (!std::strcmp(name, "val_b")) return b.val_b;
  return nullptr;
}

const void* olsr_image_field(const void* image_buffer, int32_t width, int32_t height, int32_t tile, const char* name) {
  size_t bytes;
  if (tile <= 0) tile = 15;
  const size_t tiles = (size_t)((width + tile - 1) / tile) * (size_t)((height + tile - 1) / tile);
  const ImageState im = ImageState::carve(const_cast<void*>(image_buffer), (size_t)width * height, tiles, bytes);
  if (!std::strcmp(name, "final_T")) return im.final_T;
  if (!std::strcmp(name, "n_contrib")) return im.n_contrib;
  if (!std::strcmp(name, "ranges")) return im.ranges;
  if (!std::strcmp(name, "tile_work")) return im.tile_work;    // [2][tiles]
  if (!std::strcmp(name, "tile_order")) return im.tile_order;  // [tiles]
  return nullptr;
}

void olsr_set_profiling(int enable) {
  g_profiling = enable != 0;
  marks_reset();
}

int olsr_get_stage_times(const char** names, float* ms, int max) {
  if (g_marks.size() < 2) return 0;
  (void)hipEventSynchronize(g_marks.back().ev);
  int n = 0;
  for (size_t i = 1; i < g_marks.size() && n < max; ++i) {
    if (!std::strcmp(g_marks[i].name, "begin")) continue;  // interval between two calls
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_marks[i - 1].ev, g_marks[i].ev) != hipSuccess) t = -1.f;
    names[n] = g_marks[i].name;
    ms[n] = t;
    ++n;
  }
  return n;
}

void olsr_debug_sort_timing(unsigned long long* device_buffer, int max_blocks, int max_launches) {
  debug_set_sort_timing(device_buffer, max_blocks, max_launches);
}

void olsr_debug_sync_fault(int fault_bits, int spin_limit) {
  if (fault_bits >= 0) sort_knobs().fault = fault_bits & 3;
  if (spin_limit >= 0) sort_knobs().spin_limit = spin_limit > 0 ? spin_limit : (1 << 22);
}

void olsr_debug_sort_knobs(int keys_per_thread, int resident_blocks, int legacy) {
  if (keys_per_thread >= 0) sort_knobs().kpt = keys_per_thread;
  if (resident_blocks >= 0) sort_knobs().resident = resident_blocks;
  if (legacy >= 0) sort_knobs().legacy = legacy ? 1 : 0;
}

void olsr_debug_composite_stamps(unsigned long long* device_buffer, int capacity) {
  g_stamps.buf = device_buffer;
  g_stamps.capacity = device_buffer ? capacity : 0;
  g_stamps.next = 0;
}

int olsr_debug_sort_threads(int threads) {
  if (threads == 0 || threads == 256 || threads == 1024) sort_knobs().threads = threads;
  return sort_knobs().threads.load();
}

void olsr_debug_sort_compact(int enable) {
  if (enable >= 0) sort_knobs().compact.store(enable ? 1 : 0, std::memory_order_relaxed);
}

void olsr_debug_sort_small(int enable) {
  if (enable >= 0) sort_knobs().small_sort = enable ? 1 : 0;
}

int olsr_debug_sort_plan(int64_t n, int n_is_capacity, int32_t* keys_per_thread, int32_t* blocks) {
  const SortPlan p = sort_plan((long long)n, n_is_capacity != 0);
  if (keys_per_thread) *keys_per_thread = p.kpt;
  if (blocks) *blocks = p.nblk;
  return fused_sort_applicable(n, 32) ? 1 : 0;
}

size_t olsr_debug_backward_ordered_scratch_bytes(int64_t num_rendered, int32_t F) {
  if (num_rendered < 0 || !supported_F(F)) return 0;
  return align_up((size_t)num_rendered * (size_t)grad_row(F) * sizeof(float)) + align_up((size_t)num_rendered) + 2 * ALIGN;
}

int olsr_debug_backward_ordered(const olsr_scene* scene, const void* geometry_buffer, int32_t num_rendered,
                                const void* binning_buffer, const void* image_buffer, const float* dL_dout_color,
                                const float* dL_dout_language, const float* dL_dout_depth, void* scratch, float* dL_dmeans2D,
                                float* dL_dconic, float* dL_dopacity, float* dL_dcolors, float* dL_dlanguage,
                                float* dL_ddepths, int32_t condition, void* hip_stream) {
  int rc = check_scene(scene, true);
  if (rc != OLSR_OK) return rc;
  const olsr_scene& s = *scene;
  if (s.bwd_mode != OLSR_BWD_REFERENCE && s.bwd_mode != OLSR_BWD_EXACT)
    return fail(OLSR_ERR_ARG, "bwd_mode must be OLSR_BWD_REFERENCE or OLSR_BWD_EXACT");
  if (s.P <= 0) return OLSR_OK;
  if (!geometry_buffer || !binning_buffer || !image_buffer || num_rendered < 0 || !scratch || !dL_dout_color || !dL_dmeans2D ||
      !dL_dconic || !dL_dopacity || !dL_dcolors || !dL_ddepths || (s.F > 0 && !dL_dlanguage))
    return fail(OLSR_ERR_ARG, "ordered backward: state buffers, scratch, dL_dout_color and the six outputs are required");
  const FrameDims d = frame_dims(s);
  size_t gb, ib, bb;
  const GeometryState g = GeometryState::carve(const_cast<void*>(geometry_buffer), (size_t)s.P, grad_row(s.F), gb);
  const ImageState im = ImageState::carve(const_cast<void*>(image_buffer), (size_t)d.W * d.H, (size_t)d.ntiles, ib);
  const BinningState b = BinningState::carve(const_cast<void*>(binning_buffer), (size_t)num_rendered, bb);
  float* rows = (float*)(((uintptr_t)scratch + ALIGN - 1) / ALIGN * ALIGN);
  uint8_t* used = (uint8_t*)rows + align_up((size_t)num_rendered * (size_t)grad_row(s.F) * sizeof(float));
  launch_render_backward_ordered(s, d, g, b, im, num_rendered, dL_dout_color, dL_dout_language, dL_dout_depth, rows, used,
                                 dL_dmeans2D, dL_dconic, dL_dopacity, dL_dcolors, dL_dlanguage, dL_ddepths, condition != 0,
                                 (hipStream_t)hip_stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(OLSR_ERR_DEVICE, std::string("ordered backward launch: ") + hipGetErrorString(e));
  (void)gb;
  (void)ib;
  (void)bb;
  return OLSR_OK;
}

const char* olsr_last_error(void) { return g_err.c_str(); }
const char* olsr_version(void) { return "olsr 0.1 (gfx950)"; }

}  // extern "C"
