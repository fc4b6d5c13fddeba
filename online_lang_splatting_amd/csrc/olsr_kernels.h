// olsr_kernels.h — host-side launchers of the gfx950 kernels (one translation unit each).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/olsr.h"
#include "olsr_state.h"

namespace olsr {

// Large-footprint Gaussians go through wave-per-Gaussian kernels.  The forward's emission builds two work lists in
// geometry big_list: listed in more than OLSR_BIG_FOOTPRINT tiles (from the front, count in counters[5]: emission
// and row sums) and in OLSR_MID_FOOTPRINT+1 .. OLSR_BIG_FOOTPRINT tiles (from the back, count in counters[4]: row
// sums only — a lane still emits 32 instances faster than a wave does, but sums that many gradient rows slower).
constexpr uint32_t OLSR_BIG_FOOTPRINT = 32;
constexpr uint32_t OLSR_MID_FOOTPRINT = 12;

struct FrameDims {
  int W, H, tile, gx, gy, ntiles;
  float focal_x, focal_y;
};

// k_preprocess.hip
void launch_preprocess(const olsr_scene& s, const FrameDims& d, const GeometryState& g, int32_t* radii,
                       int32_t* n_touched, hipStream_t st);
void launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t st);

// k_binning.hip
// Stable LSD radix sort of (key, val) on `bits` low bits of key, ceil(bits/8) passes of equal digit width.
// n_dev (nullable) bounds the element count on the device; n_host sizes the grid.
struct SortBuffers {
  uint32_t *key_a, *key_b, *val_a, *val_b, *table, *partials;
};
// Returns 0 if the result ends in (key_a,val_a), 1 if in (key_b,val_b).
int launch_radix_sort(const SortBuffers& b, int64_t n_host, const int32_t* n_dev, int bits, bool vals_in_identity,
                      hipStream_t st);
// fused scan of the per-Gaussian instance counts (depth order) + emission of the instances; also zeroes the first
// bin_sync_words words of the binning buffer's synchronisation area
// (n_host: the instance count, or the caller's capacity when the count is only known on the device)
// order: the depth order to emit in (g.depth_order, or the carried one); alt_totals / alt_unless (may be null): the per-block
// instance totals come from alt_totals instead of g.emit_status when *alt_unless == 0 (the carried order was repaired)
// n_order_dev (may be null = P): the number of entries of `order` when the depth sort compacted its input
void launch_emit(const olsr_scene& s, const FrameDims& d, const GeometryState& g, const BinningState& b,
                 int64_t bin_sync_words, int64_t n_host, const uint32_t* order, const uint32_t* alt_totals,
                 const uint32_t* alt_unless, const int32_t* n_order_dev, hipStream_t st);
// exclusive scan of popcount(flags) over [0, n] -> rowbase[0..n] in one kernel; counters[6] = total live rows,
// counters[7] = (total > row_capacity) or the forward's instance overflow
// packed_ref15: rows per instance = packed survivor waves (flag bits 4-5) instead of forward slots (bits 0-3)
// row_status / sync: BinningState::row_status and tickets + 8 (zeroed by the forward, re-zeroed by the kernel itself)
void launch_row_compaction(const uint8_t* flags, int64_t n_host, const int32_t* n_dev, bool packed_ref15,
                           uint32_t* rowbase, uint32_t* row_status, uint32_t* sync, int64_t row_capacity,
                           int32_t* counters, int32_t* status_dev, hipStream_t st);
// backward launch order: inside each XCD's contiguous chunk of tiles, heaviest (most live pairs) first
// tile_work: [2][ntiles] (ImageState).  rows_mailbox (may be null): device view of four host words that receive {sum of
// tile_work[0], sum of tile_work[1], rows_seq, 0} (the drop-in path sizes its backward scratch from them without a
// synchronisation); live_rows: ImageState::live_rows, zero on entry, the staging of those sums
// (reduced by one block of the tile-order kernel when a forward carries the fused loss epilogue; see k_loss.hip below)
struct LossFinalArgs {
  const float* partials;  // [nb][LOSS_SUMS], or null: nothing to do
  int nb, W, H, F, has_lang;
  float alpha, lamda;
  float* loss;
  float* d_exposure;       // written if use_exposure, zeroed if zero_exposure, may be null
  int use_exposure, zero_exposure;
};
// rows (may be null): the row compaction of launch_row_compaction in the same launch (forward_tail_kernel) — for a forward
// whose caller announced the backward's row capacity (olsr_scene.backward_row_capacity); the backward then skips its own
struct ForwardTailRows {
  const uint8_t* flags;
  int64_t n_host;
  const int32_t* n_dev;
  bool packed_ref15;
  uint32_t* rowbase;
  uint32_t* row_status;
  uint32_t* sync;
  int64_t row_capacity;
  int32_t* counters;
};
void launch_tile_order(const uint32_t* tile_work, uint32_t* tile_order, uint32_t* order_copy, int ntiles,
                       uint32_t* live_rows, int32_t* rows_mailbox, int32_t rows_seq, const int32_t* counters,
                       int32_t* num_rendered_dev, int32_t* sticky_error, const uint32_t* hint_slot, float* depth_cut,
                       int gx, int gy, const LossFinalArgs& loss_final, const ForwardTailRows* rows, hipStream_t st);
struct RowsMailbox {  // set by olsr_forward for the duration of one call (thread-local in olsr_api.hip)
  int32_t* dev = nullptr;
  int32_t seq = 0;
  int32_t* sticky = nullptr;  // device view of the process-wide "a frame had a synchronisation error" host word (may be null)
  const uint32_t* hint_slot = nullptr;  // word 0 = which of the stream's per-view tile orders this frame uses (olsr_api.hip)
  int64_t compact_rows_n = -1;  // >= 0: the forward's last launch also compacts the backward's rows (the instance capacity)
};
RowsMailbox& rows_mailbox_of_this_call();
// ranges must have been zeroed (launch_instance_offsets); also clears flags[0, n)
void launch_tile_ranges(const uint32_t* sorted_keys, int64_t n_host, const int32_t* n_dev, uint32_t* ranges,
                        uint8_t* flags, hipStream_t st);

// k_sort.hip — one kernel per radix pass (digit counts published per block, see the file header)
struct FusedHouse {  // the frame's bookkeeping done by block 0 of the depth sort's histogram kernel
  const uint32_t* part_rect;
  const uint32_t* part_count;
  int nparts;
  long long capacity;
  int32_t* counters;
  int32_t* num_rendered_dev;
  uint32_t* ranges;  // initialised to {UINT_MAX, 0} per tile (empty)
  int nranges;
  int32_t* host_mailbox;  // (may be null) device view of two host words: receives {R, host_seq}, in this order
  int32_t host_seq;
  uint32_t* live_rows;    // ImageState::live_rows (zeroed by the kernel)
  uint32_t* hint_base;    // (may be null) the stream's per-view tile orders (olsr_api.hip: order_hint_of): the kernel picks this
  const float* view;      //   frame's slot by its view matrix (hint_pick_wave)
};
// per-view tile orders of the synchronising entry (olsr_api.hip): [0] chosen slot, [1] use counter, [4, 4 + S) last use of every
// slot, then S x 16 floats (view matrices, NaN = empty), then S x ntiles orders
constexpr int HINT_SLOTS = 16;
constexpr int HINT_HDR = 4 + HINT_SLOTS + 16 * HINT_SLOTS;  // words in front of the orders
static_assert(HINT_HDR % 4 == 0, "the orders stay 16-byte aligned");
constexpr float HINT_VIEW_TOL = 0.03f;
bool fused_sort_applicable(int64_t n_host, int bits);
int fused_sort_digit_bits(int bits, int* passes_out);
// digit totals of all passes (hist[pass][256], zeroed beforehand); optionally the frame's bookkeeping
// run_if (may be null; also launch_sort_fused / launch_small_depth_sort): a device word — the kernels do their work only when it
// is non-zero (the carried depth order could not be repaired, k_order_carry.hip); the frame's bookkeeping happens either way
// compaction (may be null; depth sort only): the kernel also writes the (key, index) of the Gaussians that emit instances
// (inst_count != 0) densely, in index order, into out_keys / out_gid, their number into *n_out, and counts digits of those
// only — the passes then sort out_keys / out_gid with n_dev = n_out (k_sort.hip: "Visible-set compaction")
struct SortCompaction {
  const uint32_t* inst_count;
  const uint32_t* part_vis;  // per block of 256 Gaussians: how many emit (preprocess)
  uint32_t* out_keys;
  uint32_t* out_gid;
  int32_t* n_out;
};
bool depth_sort_compaction_applicable(int64_t P);
void launch_sort_hist(const uint32_t* keys, int64_t n_host, const int32_t* n_dev, int bits, uint32_t* hist,
                      const FusedHouse* house, int threads /* 256 or 1024: SortPlan::threads */, hipStream_t st,
                      const uint32_t* run_if = nullptr, const SortCompaction* compaction = nullptr);
// stable sort on the low `bits` bits; status: [passes][plan.nblk][1 << digit bits] zeroed 16-bit words (plan = sort_plan(n_host, ...)), tickets:
// one zeroed word per pass.  flags_clear (with vals_in_identity): byte i is cleared for every element; ranges: the
// last pass derives per-key [start, end) (keys must then be tile ids).  Returns where the result ends (0: a, 1: b);
// the sorted keys of the LAST pass are not written.  emit_totals (depth sort): the last pass adds every value's
// instance count (inst_count[v]) to emit_totals[final position / EMIT_CHUNK] (zeroed beforehand).
int launch_sort_fused(const SortBuffers& b, const SortPlan& plan, int64_t n_host, const int32_t* n_dev, int bits,
                      bool vals_in_identity, const uint32_t* hist, uint32_t* status, uint32_t* tickets, uint8_t* flags_clear, uint32_t* ranges,
                      const uint32_t* inst_count, uint32_t* emit_totals, int32_t* sync_error, int fault, hipStream_t st,
                      uint32_t* final_vals_out = nullptr, const uint32_t* run_if = nullptr);
void debug_set_sort_timing(unsigned long long* buf, int max_blocks, int max_launches);
// the same totals for a depth order produced by the multi-kernel passes
// the whole depth sort of n <= 8 192 Gaussians in one launch, incl. the frame's bookkeeping (k_sort.hip: sort_small_kernel)
bool small_depth_sort_applicable(int64_t n, bool as_fallback = false);
void launch_small_depth_sort(const uint32_t* keys, int n, uint32_t* order_out, const uint32_t* inst_count,
                             uint32_t* emit_totals, const FusedHouse* house, const uint32_t* run_if, hipStream_t st);
// k_order_carry.hip — the previous frame's depth order repaired under this frame's keys (olsr_scene.depth_order_carry):
// carry [P] in: any content, out: the new order if *miss stays 0 (else a partial repair the radix passes then overwrite);
// keys [P] by Gaussian; tmp_key / tmp_gid [P] scratch; totals: instances per emission block of EMIT_CHUNK ranks;
// miss: a zeroed device word
void launch_order_repair(int P, uint32_t* carry, const uint32_t* keys, uint32_t* tmp_key, uint32_t* tmp_gid,
                         const uint32_t* inst_count, uint32_t* totals, uint32_t* miss, bool frames_in_flight,
                         hipStream_t st);
void launch_emit_totals(const uint32_t* order, int P, const uint32_t* inst_count, uint32_t* emit_totals, hipStream_t st);

// k_render_fwd.hip
// loss (may be null): evaluate the mapping / tracking loss in the composite's epilogue (olsr_forward_async_loss)
void launch_render_forward(const olsr_scene& s, const FrameDims& d, const GeometryState& g, const BinningState& b,
                           const ImageState& im, float* out_color, float* out_language, float* out_depth,
                           float* out_opacity, int32_t* n_touched, uint32_t* tile_order_inout, int32_t* num_rendered_dev,
                           const olsr_loss_fusion* loss, hipStream_t st);

// k_render_bwd.hip
// (two translation units, one per backward mode, so they compile in parallel)
// F_rows: language channels of the partial-gradient rows: s.F, or 0 when there is no language cotangent
void launch_render_backward_reference(const olsr_scene& s, int F_rows, const FrameDims& d, const GeometryState& g,
                                      const BinningState& b, const ImageState& im, const float* dL_dcolor,
                                      const float* dL_dlanguage, const float* dL_ddepth, float* rows, hipStream_t st);
void launch_render_backward_exact(const olsr_scene& s, int F_rows, const FrameDims& d, const GeometryState& g,
                                  const BinningState& b, const ImageState& im, const float* dL_dcolor,
                                  const float* dL_dlanguage, const float* dL_ddepth, float* rows, hipStream_t st);

// k_render_bwd_ordered.hip — test instrument: the composite backward in the reference's own association (olsr_debug_backward_ordered)
void launch_render_backward_ordered(const olsr_scene& s, const FrameDims& d, const GeometryState& g, const BinningState& b,
                                    const ImageState& im, int64_t num_rendered, const float* dL_dcolor,
                                    const float* dL_dlanguage, const float* dL_ddepth, float* rows, uint8_t* used,
                                    float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolors,
                                    float* dL_dlanguage_out, float* dL_ddepths, bool condition, hipStream_t st);

// k_preprocess_bwd.hip
struct GradOut {
  float *dL_dmeans2D, *dL_dconic, *dL_dopacity, *dL_dcolors, *dL_dlanguage, *dL_ddepths, *dL_dmeans3D, *dL_dcov3D,
      *dL_dsh, *dL_dscales, *dL_drotations, *dL_dtau, *dL_dtau_sum;
  // optional fused accumulation (olsr_grad_bucket); every pointer above may then be NULL
  float* bucket_flat = nullptr;
  float* bucket_densify = nullptr;
  int32_t* bucket_max_radii = nullptr;
  int bucket_assign = 0;
  unsigned long long* bucket_row_mask = nullptr;  // olsr_grad_bucket.row_mask (include/olsr.h)
  int32_t* status_dev = nullptr;  // olsr_backward's {L, overflow}: the last kernel raises [1] to 2 on a synchronisation error
  int32_t* sticky_error = nullptr;  // ... and sets this mapped host word (RowsMailbox::sticky), if there is one
  bool status_rows = false;  // the rows were compacted by the forward: the last kernel also writes status_dev = {L, overflow}
};
// F_rows: language channels of `rows` (see above); with F_rows == 0 < s.F the language gradients are written as zeros
void launch_preprocess_backward(const olsr_scene& s, int F_rows, const FrameDims& d, const GeometryState& g,
                                const BinningState& b, const float* rows, const int32_t* radii, const GradOut& o,
                                float* tau_partials, hipStream_t st);
int tau_partial_blocks(int P);

// k_accumulate.hip
void launch_exchange_mask(int P, int width, const float* flat, const unsigned long long* row_mask, const int32_t* max_radii,
                          int32_t* imax, hipStream_t st);
void launch_exchange_pack(int P, int width, int cap, const float* flat, const int32_t* imax, int32_t* max_radii,
                          unsigned long long* row_mask, const float* densify, int32_t* idx, float* fsum, int32_t* counts,
                          int32_t* status, hipStream_t st);
void launch_exchange_unpack(int P, int width, int cap, const int32_t* idx, const float* fsum, float* flat, float* densify,
                            hipStream_t st);
void launch_accumulate(int P, int M, int F, bool assign, const float* dmeans3D, const float* dsh,
                       const float* dopacity, const float* dscales, const float* drot, const float* dlang,
                       const float* dmeans2D, const int32_t* radii, float* flat, float* densify, int32_t* max_radii,
                       hipStream_t st);
void launch_bucket_add(int P, int width, float* dst, const float* src, unsigned long long* dst_mask,
                       const unsigned long long* src_mask, float* dst_densify, const float* src_densify,
                       int32_t* dst_radii, const int32_t* src_radii, hipStream_t st);

// k_adam.hip
constexpr int OLSR_ADAM_MAX_BUCKETS = 8;
void launch_adam_step(int P, int M, int F, const olsr_adam_params& hp, const float* const* flats,
                      const unsigned long long* const* masks, int n_flats, float* means3D, float* shs, float* opacities, float* scales, float* rotations, float* language,
                      float* exp_avg, float* exp_avg_sq, hipStream_t st);

// k_pose.hip
void launch_pose_step(const olsr_pose_params& p, const float* dL_dtau_sum, const float* dL_dexposure, const float* proj,
                      float* state, int32_t* status, const int32_t* frame_status, hipStream_t st);

// k_knn.hip
size_t knn_scratch_bytes(int P);
void launch_knn(int P, const float* points, float* mean_dist2, void* scratch, hipStream_t st);

// k_loss.hip
int loss_blocks(int W, int H);
// the final reduction of a loss's partial sums (olsr_loss_device.h: loss_final_block); for the fused epilogue of the forward
// composite it runs in one block of the tile-order kernel (partials == nullptr: nothing to do)
LossFinalArgs loss_final_args(const float* partials, int nb, const olsr_loss_params& p, bool tracking, bool has_lang,
                              bool use_exposure, float* loss, float* dL_dexposure);
void launch_mapping_loss(const olsr_loss_params& p, const float* image, const float* depth, const float* language,
                         const float* gt_image, const float* gt_depth, const float* gt_language, const float* exposure,
                         const float* opacity, const float* grad_mask, bool tracking, float* dL_dimage,
                         float* dL_ddepth, float* dL_dlanguage, float* loss, float* dL_dexposure, float* partials,
                         hipStream_t st);

}  // namespace olsr
