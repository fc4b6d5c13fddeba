/*
 * olsr.h — C-ABI of the MI355X-native language-Gaussian rasterizer ("olsr").
 *
 * This is the drop-in boundary for the one hot path of rpng/online_lang_splatting:
 * the differentiable tile rasterizer behind `diff_gaussian_rasterization`.
 * Every entry point takes plain device pointers, sizes and a HIP stream handle
 * (no torch types).  The reference interface each entry point replaces is cited
 * as file:line relative to
 *   DGR = /root/reference/submodules/diff-gaussian-rasterization
 *   CR  = DGR/cuda_rasterizer
 *
 * Conventions (identical to the reference):
 *   - all float arrays are fp32, row-major, contiguous, resident on the GPU;
 *   - viewmatrix / projmatrix / projmatrix_raw are the 16 floats the Python caller
 *     holds (i.e. the transposes W2C^T, (P*W2C)^T, P^T) — column-major to the kernels;
 *   - opacities are post-sigmoid, scales post-exp, rotations are NOT re-normalised (unless
 *     scene->activations says the arrays are raw, see OLSR_ACT_*);
 *   - outputs are fully overwritten by the library (the caller need not zero them).
 *
 * Error behaviour: every function returns OLSR_OK (0) or a negative code and
 * records a message retrievable with olsr_last_error() (thread-local).
 *
 * Contents: the rasterizer itself (olsr_forward, olsr_forward_async, olsr_backward, olsr_mark_visible and
 * their size / introspection / profiling helpers — SURVEY.md section 8 rows a-e), then the callers and
 * data either side of it (rows f1-f3): olsr_mapping_loss, olsr_tracking_loss, olsr_pose_step,
 * olsr_accumulate_gradients, olsr_sparse_exchange_mask / _pack / _unpack, olsr_adam_step, olsr_knn_mean_dist2.
 */
#ifndef OLSR_H_INCLUDED
#define OLSR_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OLSR_OK 0
#define OLSR_ERR_ARG (-1)      /* bad argument (shape / exclusivity / unsupported F, tile) */
#define OLSR_ERR_DEVICE (-2)   /* HIP runtime error (message holds hipGetErrorString) */
#define OLSR_ERR_ALLOC (-3)    /* allocation callback returned NULL */
#define OLSR_ERR_CAPACITY (-4) /* async mode: instance count exceeded caller capacity */

/* Device-side status words of the sync-free entries (num_rendered_dev[1] of olsr_forward_async, status_dev[1] of
 * olsr_backward):  0 = fine;  1 = capacity overflow (instances resp. gradient rows; nothing rendered / zero gradients);
 * 2 = synchronisation error: a block of the frame's radix passes or of the row compaction waited for a predecessor's counts
 * until its spin bound ran out — the frame's synchronisation words were overwritten from outside mid-frame.  The library
 * never hangs on that and never writes out of bounds, but the tile lists are garbage: the forward's images must not be used,
 * the backward writes zero gradients.  The synchronising olsr_backward (scratch_alloc != NULL) returns OLSR_ERR_DEVICE; for
 * olsr_forward and for an olsr_backward without status_dev (the reference-shaped bindings) the frame's last kernel raises a
 * flag in mapped host memory and the FIRST olsr_forward / olsr_backward ON THE SAME DEVICE AND STREAM after the GPU got there
 * returns OLSR_ERR_DEVICE (once), the way an asynchronous HIP error surfaces (round 6: the flag is keyed by device and stream —
 * up to 63 pairs, further ones share one word —, so a broken frame on one stream does not fail healthy calls on another). */
#define OLSR_STATUS_OK 0
#define OLSR_STATUS_OVERFLOW 1
#define OLSR_STATUS_SYNC_ERROR 2

/* Backward flavour.  REFERENCE reproduces what the shipped CUDA computes with its
 * 15x15 tiles (CR/config.h:17-18): the 225-lane tree reduction of
 * CR/backward.cu:684-702 keeps 128 of 225 pixel ranks, language gradients come from
 * tile rank 0 only (CR/backward.cu:1137,1194-1197) and the language recursion is not
 * skip-guarded (CR/backward.cu:1127-1139).  EXACT is the true gradient.
 * Both modes are tolerance-level, not bit-level, restatements of the reference's backward: the value path is re-associated
 * (running form of the behind-colour recursion, dot form of the language recursion, a wave-level reduction tree instead of
 * the shared-memory tree + float atomics), so gradients agree with the reference's arithmetic to the north-star tolerance
 * (>= 99.99 % of the elements of every tensor within 1e-4), not bit for bit; recursion-only visits are skipped while a
 * wave's last_alpha is still zero, which presumes finite features and cotangents (0 * x == 0). */
#define OLSR_BWD_REFERENCE 0
#define OLSR_BWD_EXACT 1

/* Allocation callback: must return a device pointer to >= nbytes (256-byte aligned),
 * valid until the matching backward has run.  Mirrors the std::function<char*(size_t)>
 * resize functionals of CR/rasterizer.h:34-36 and DGR/rasterize_points.cu:27-33. */
typedef void *(*olsr_alloc_fn)(void *user, size_t nbytes);

/* Caller-side activations folded into the kernels.  The reference applies
 * sigmoid / exp / normalize in PyTorch before every render (GaussianModel.get_opacity / get_scaling /
 * get_rotation, gaussian_splatting/scene/gaussian_model.py:95-105) and autograd chains back through them.
 * With a bit set, the corresponding array holds the RAW parameter, preprocess applies
 *   opacity = 1 / (1 + exp(-x))      scale = exp(x)      rotation = q / max(|q|, 1e-12)
 * and the backward returns the gradient with respect to the RAW parameter (dL_dopacity, dL_dscales,
 * dL_drotations and the bucket rows).  With a raw opacity the backward needs scene->opacities too. */
#define OLSR_ACT_OPACITY_SIGMOID 1
#define OLSR_ACT_SCALE_EXP 2
#define OLSR_ACT_ROTATION_NORMALIZE 4

/* olsr_scene.flags.
 *   OLSR_FLAG_SIGNED_EMPTY_RADII  (forward) a Gaussian inside the frustum whose bounding square covers no tile gets
 *       radii[i] = -radius instead of 0 (it still emits nothing, and n_touched stays 0).  The disentangled rasterizer
 *       (DGR-D = submodules/diff-gaussian-rasterization-disentangle-optim) needs it: its
 *       preprocess writes BOTH radii of a Gaussian as soon as ONE of its two covariance sets covers a tile
 *       (DGR-D/cuda_rasterizer/forward.cu:391-431), so the caller that composes the two passes must know the radius of
 *       the set that covered none.  A backward must be given max(radii, 0). */
#define OLSR_FLAG_SIGNED_EMPTY_RADII 1
/*   OLSR_FLAG_FWD_ACCUM_MFMA  (forward) the per-pixel feature accumulation C += f alpha T of the forward composite — a dense
 *       [64 pixels x K entries] x [K x (4 + F) channels] contraction per wave — runs on the matrix cores
 *       (v_mfma_f32_16x16x4_f32: exact fp32, a k-ordered fma chain) instead of the vector ALU.  Every decision (alpha
 *       floor, saturation, n_touched, the backward's liveness flags) stays on the vector ALU and is unchanged, so
 *       final_T / n_contrib / radii / n_touched are bit-identical; the images are rounded as fma(alpha T, f, C) where the
 *       reference's source order (CR/forward.cu:479-484) gives fma(f alpha, T, C): equal to ~1e-7 relative, not bit for
 *       bit.  Non-finite features then poison the 16-pixel block instead of the blending pixels only. */
#define OLSR_FLAG_FWD_ACCUM_MFMA 2
/*   OLSR_FLAG_FWD_ACCUM_WEIGHT  (forward) the same rounding as the MFMA variant on the vector ALU: w = alpha T once per pixel,
 *       then one fma(w, f, C) per channel instead of the reference's mul + fma (half the lane operations of the
 *       accumulation).  Decisions unchanged and bit-identical; images to ~1e-7 relative of the reference's rounding. */
#define OLSR_FLAG_FWD_ACCUM_WEIGHT 4
/*   OLSR_FLAG_FRAMES_IN_FLIGHT  (forward; round 5) the caller keeps SEVERAL frames in flight on several HIP streams (the views of
 *       a mapping iteration, a benchmark's lanes).  One composite kernel fills the chip, and the workgroup dispatcher keeps
 *       feeding it while it has workgroups left: a 1024-thread workgroup of ANOTHER frame's radix sort needs sixteen free wave
 *       slots on one CU and gets them only in the composite's tail, so the frames' dependent binning chains stalled behind each
 *       other's composites (kernel trace, four frames in flight: radix passes of 15 us stretched to 147 us, no composite
 *       executing during 26 % of the wall time).  With the flag the histogram and pass kernels of this frame run as four-wave
 *       workgroups with sixteen keys per thread, which are placed sooner: + 2 % frames/s with four frames in flight at config 3
 *       (2 304 -> 2 350, K = 60) — and - 11 % with ONE frame in flight (the depth sort 69 -> 108 us), which is why it is a
 *       statement about the caller and not the default.  Lists, images and gradients are bit-identical either way. */
#define OLSR_FLAG_FRAMES_IN_FLIGHT 8

/* How Gaussians are binned into tiles.
 *   OLSR_BINNING_RECT     every tile of the reference's bounding square (getRect, CR/auxiliary.h:46-56):
 *                         instance lists, num_rendered and n_contrib equal the reference's bit for bit.
 *   OLSR_BINNING_ELLIPSE  only the tiles that the ellipse alpha >= 1/255 can reach (a subset of the
 *                         square, same order).  The dropped (tile, Gaussian) pairs blend no pixel in the
 *                         forward and are whole-tile skips in the backward, so every image, radii,
 *                         n_touched and every gradient is unchanged; num_rendered and the state buffers'
 *                         list positions (n_contrib) count the kept pairs. */
#define OLSR_BINNING_RECT 0
#define OLSR_BINNING_ELLIPSE 1

/* Per-call scene description shared by forward and backward.
 * Replaces the positional argument lists of
 *   CudaRasterizer::Rasterizer::forward / backward           CR/rasterizer.h:33-104
 *   CudaRasterizer::LanguageRasterizer::forward / backward   CR/rasterizer.h:115-197
 * F == 0 selects the RGB-only rasterizer (GaussianRasterizer), F > 0 the language one. */
typedef struct olsr_scene {
  int32_t P;           /* number of Gaussians */
  int32_t D;           /* active SH degree 0..3 */
  int32_t M;           /* SH coefficients per channel in `shs` (0 when shs == NULL) */
  int32_t F;           /* language channels: 0, 3, 15, 16 or 32 */
  int32_t width;
  int32_t height;
  int32_t tile;        /* logical tile edge in pixels: 15 (reference, CR/config.h) or 16 */
  int32_t prefiltered; /* CR/auxiliary.h:154-161 */
  int32_t debug;       /* synchronise + check after every stage (CR/auxiliary.h:166-173) */
  int32_t bwd_mode;    /* OLSR_BWD_REFERENCE or OLSR_BWD_EXACT (used by backward only) */
  float tan_fovx;
  float tan_fovy;
  float scale_modifier;
  int32_t binning;     /* OLSR_BINNING_RECT or OLSR_BINNING_ELLIPSE (used by forward only) */
  const float *background;       /* [3] */
  const float *means3D;          /* [P,3] */
  const float *shs;              /* [P,M,3] or NULL */
  const float *colors_precomp;   /* [P,3] or NULL (exactly one of shs / colors_precomp) */
  const float *language_precomp; /* [P,F]; required when F > 0 (CR/rasterizer_impl.cu:500-502) */
  const float *opacities;        /* [P] */
  const float *scales;           /* [P,3] or NULL */
  const float *rotations;        /* [P,4] or NULL */
  const float *cov3D_precomp;    /* [P,6] or NULL (exactly one of scales+rotations / cov3D) */
  const float *viewmatrix;       /* [16] */
  const float *projmatrix;       /* [16] */
  const float *projmatrix_raw;   /* [16] (backward only; may be NULL in forward) */
  const float *cam_pos;          /* [3] */
  int32_t activations; /* OLSR_ACT_* bit mask: which parameter arrays are RAW (pre-activation); 0 = the
                        * reference's calling convention (already activated) */
  int32_t flags;       /* OLSR_FLAG_* bit mask, 0 = the reference's behaviour */
  float *tile_depth_cut; /* NULL (the default), or device float[2 x tiles], in/out — see "Per-tile depth cut-offs" below.
                          * Used by olsr_forward_async / olsr_forward_async_loss only. */
  int64_t backward_row_capacity; /* 0 (the default), or — for a caller that owns a fixed backward scratch — its row capacity
                          * (`scratch_rows` of the olsr_backward that will follow, with this bwd_mode and tile).  The forward's last
                          * launch then also compacts the backward's rows (which depend on the forward alone), and an
                          * olsr_backward given a scene with the same non-zero value skips its own compaction launch: one launch
                          * and ~8 us less per frame.  Both calls must see the same value, bwd_mode and tile; the backward
                          * verifies scratch_rows == backward_row_capacity (OLSR_ERR_ARG) and needs scratch_alloc == NULL. */
  uint32_t *depth_order_carry; /* NULL (the default), or device uint32[P], in/out — see "Carried depth order" below.  Used by
                          * olsr_forward_async / olsr_forward_async_loss only. */
} olsr_scene;

/* Carried depth order (round 6; a performance hint for SEQUENCES of nearly identical views of the same Gaussians: the ~100
 * dependent tracking iterations of a frame, utils/slam_frontend.py:163-277, and the ~150 mapping iterations over one window of
 * keyframes, utils/slam_backend.py:499-670 — one array per VIEW).  The depth sort (hist + four radix passes: five dependent
 * launches, ~70 us whatever it sorts) orders the P Gaussians by (depth bits, index); between two iterations of such a loop
 * the pose or the parameters move by a step of the optimiser and the order changes only locally.  With
 * scene->depth_order_carry the forward REPAIRS the order the previous forward left in the array instead of sorting from
 * scratch: two launches of independent workgroups (no look-back, no ticket) merge-sort windows of 2048 ranks of the old
 * order under the new keys, first aligned, then offset by half a window, which is a complete sort whenever no Gaussian moved
 * by more than 1024 ranks; the second launch proves the result — every adjacent pair of (depth bits, index) strictly
 * ascending, which also proves that the array is a permutation — and otherwise raises a device-side flag that makes the
 * radix passes, which are enqueued behind the repair in any case, run instead of returning at once.  The lists are therefore
 * ALWAYS the reference's, bit for bit, whatever the array held on entry (zeros, another view's order, garbage: the frame then
 * simply pays for both); on exit the array holds this frame's order.  Sort keys are defined for all P Gaussians (the view-space
 * depth also for Gaussians outside the frustum, a monotone function of it behind the near plane), so that the order is a
 * property of the pose and not of what is visible.  olsr_forward (the synchronising entry) ignores the field, and so do
 * sorts that take the multi-kernel passes (more than 67 M Gaussians). */

/* Per-tile depth cut-offs (round 4; an opt-in for SEQUENCES of nearly identical views: the ~100 tracking iterations of a
 * frame).  The forward composite reads a tile's depth-sorted list only until every pixel of the tile is saturated — at config 3
 * that is 14 % of the instances; the rest is emitted, tile-sorted and compacted for nothing.  With scene->tile_depth_cut
 * (device float[2 x tiles]: [0, tiles) the cut-offs in force, in/out — initialise with +infinity = no cut; [tiles, 2 tiles)
 * library scratch) the forward drops every Gaussian that lies behind the cut-off of EVERY tile its footprint reaches, and
 * leaves behind, per tile, the cut-off for the NEXT frame: 1.1 x the depth of the entry at which the tile saturated (+ 0.01),
 * +infinity if it did not saturate, then the largest of the tile's 3 x 3 neighbourhood (a depth edge may move into the tile).
 * A tile's list is complete up to its cut-off depth, so the result is EXACT — bit-identical images, radii and n_touched,
 * gradients equal up to the order of their per-Gaussian sums; num_rendered counts the kept instances — whenever every tile
 * saturated at an entry in front of its cut-off.  If one did not (the view or the scene moved too much) the frame is flagged
 * OLSR_STATUS_CUT_MISS in num_rendered_dev[1]: its images and gradients may lack contributions in that tile, and the caller
 * re-renders.  The array has been updated by then: a tile that did not saturate at all carries +infinity again (and so do
 * its neighbours, through the 3 x 3 maximum); a tile that saturated, but BEHIND its cut-off, carries the finite
 * 1.1 x (the depth it saturated at) + 0.01 — taken from a list that may have had holes, hence possibly still too near: such a
 * tile can miss once more and converges from below, every miss moving its cut-off outwards.  Exact tile binning
 * only (OLSR_BINNING_ELLIPSE; ignored with OLSR_BINNING_RECT), default forward accumulation only, and with a fused loss only
 * the tracking loss (OLSR_ERR_ARG otherwise); olsr_forward (the synchronising entry) ignores the field.
 * Without a host read-back: olsr_backward on the state buffers of a CUT_MISS frame writes zero gradients (status_dev[1] = 3),
 * and olsr_pose_step_gated given that frame's num_rendered_dev takes no step — an iteration whose frame missed is a no-op on
 * the device, and the k-th COUNTED step sees the pose the k-th step of the loop without cut-offs sees.  A missed iteration
 * still consumes one of a caller's fixed budget of iterations (the reference's tracking_itr_num): a caller that wants the
 * same NUMBER of optimiser steps iterates until status[1] of olsr_pose_step_gated (steps done, on the device) has reached
 * it — slam_iterations.TrackingLoop.run(steps) does, reading the count back every few iterations. */
#define OLSR_STATUS_CUT_MISS 3

/* Sizes of the three opaque state buffers (bytes).  Replace
 * CudaRasterizer::required<GeometryState|ImageState|BinningState>, CR/rasterizer_impl.h:84-90. */
size_t olsr_geometry_bytes(int32_t P, int32_t F);
size_t olsr_image_bytes(int32_t width, int32_t height, int32_t tile);
size_t olsr_binning_bytes(int64_t num_rendered, int32_t F);

/* Forward.  Replaces RasterizeGaussiansCUDA / RasterizeLanguageGaussiansCUDA ->
 * Rasterizer::forward / LanguageRasterizer::forward
 * (DGR/rasterize_points.cu:35-123,125-241; CR/rasterizer_impl.cu:216-362,364-525).
 * The three callbacks are invoked once each (geometry, image, then — after the
 * instance count is known, one host sync like CR/rasterizer_impl.cu:454-455 —
 * binning).  Outputs: out_color[3,H,W], out_language[F,H,W] (ignored when F == 0),
 * out_depth[H,W], out_opacity[H,W], radii[P] (int32), n_touched[P] (int32).
 * *num_rendered receives R, the number of (Gaussian, tile) instances. */
/* Process-wide state olsr_forward keeps (results never depend on it; tested): per (device, stream, tile count) the
 * heaviest-first tile orders of up to 16 VIEWS the stream has rendered (launch-order hints: a one-wave kernel in front of the
 * composite picks the stored view matrix nearest to this frame's, else recycles the least recently used slot — the
 * reference's mapping loop renders its window of keyframes in turn on one stream; 16 x tiles x 4 bytes, stream-ordered
 * allocation, at most 64 keys with least-recently-used eviction, never freed otherwise); a ring of 256 mapped host slots for the gradient-row counts
 * (olsr_live_rows); two mapped host words per calling thread for the instance count (leaked at thread exit by design: a
 * thread_local destructor could run after the HIP runtime's teardown). */
int olsr_forward(const olsr_scene *scene,
                 olsr_alloc_fn geometry_alloc, void *geometry_user,
                 olsr_alloc_fn binning_alloc, void *binning_user,
                 olsr_alloc_fn image_alloc, void *image_user,
                 float *out_color, float *out_language, float *out_depth, float *out_opacity,
                 int32_t *radii, int32_t *n_touched,
                 int32_t *num_rendered, void *hip_stream);

/* Forward without a host sync: the caller provides all three buffers, the binning
 * buffer sized for `capacity` instances (olsr_binning_bytes(capacity, F)).  R stays on
 * the device; `num_rendered_dev` (device int32[2]) receives {R, status}: OLSR_STATUS_OK, _OVERFLOW
 * (nothing is rendered when R > capacity) or _SYNC_ERROR.  This is the entry the
 * benchmark and the frame-sharded trainer use; it has no reference counterpart
 * (SURVEY.md §7 step 8).
 *
 * tile_order_inout (device uint32[tiles], may be NULL): launch-order hint.  On entry it must hold a
 * permutation of [0, tiles) in which, inside each eighth of the range (one XCD's share), tiles are
 * sorted by expected work, heaviest first — the identity is fine for a first frame; on exit it holds
 * exactly that order measured on THIS frame.  Consecutive frames of a SLAM sequence (and the ~100
 * tracking iterations per frame) see nearly the same per-tile load, so feeding the array back lets the
 * forward composite start its stragglers first.  The result never depends on the order. */
int olsr_forward_async(const olsr_scene *scene,
                       void *geometry_buffer, void *binning_buffer, int64_t capacity,
                       void *image_buffer,
                       float *out_color, float *out_language, float *out_depth, float *out_opacity,
                       int32_t *radii, int32_t *n_touched,
                       int32_t *num_rendered_dev, uint32_t *tile_order_inout, void *hip_stream);

/* Backward.  Replaces RasterizeGaussiansBackwardCUDA / RasterizeLanguageGaussiansBackwardCUDA ->
 * Rasterizer::backward / LanguageRasterizer::backward
 * (DGR/rasterize_points.cu:243-331,333-455; CR/rasterizer_impl.cu:529-636,638-756).
 * `num_rendered` is the R returned by olsr_forward — or, for buffers filled by
 * olsr_forward_async, the `capacity` that call was given (it fixes the carving of the
 * binning buffer).
 *
 * Scratch.  The composite backward writes one partial-gradient row per (instance, 64-pixel
 * slot) pair that the forward actually blended; the per-Gaussian reduction then sums them in a
 * fixed order (this replaces the reference's float atomicAdd, CR/backward.cu:1176-1198, and
 * makes the gradients bit-reproducible).  The number of such pairs, L <= 4 R, is known on the
 * device after the forward.  Two ways to provide the row buffer, mirroring the forward:
 *   - scratch_alloc != NULL: the function computes L, synchronises once, and asks the callback
 *     for olsr_backward_scratch_bytes(L, F) bytes (what the drop-in Python path does);
 *   - scratch_alloc == NULL: `scratch` holds olsr_backward_scratch_bytes(scratch_rows, F) bytes;
 *     no synchronisation.  If L > scratch_rows nothing is written and the overflow is reported
 *     through status_dev.
 * status_dev (device int32[2], may be NULL) receives {L, status}: OLSR_STATUS_OK / _OVERFLOW / _SYNC_ERROR.
 *
 * Cotangents.  dL_dout_color[3,H,W] is required.  dL_dout_language[F,H,W] and dL_dout_depth[H,W] may be NULL: "the loss
 * does not depend on that image" — what autograd hands the reference's backward as None and PyTorch turns into zeros
 * (DGR/diff_gaussian_rasterization/__init__.py:296-345).  The front end's tracking loss has no language term
 * (utils/slam_utils.py:92-121): without a language cotangent the RGB instantiation of the composite backward runs on the
 * language forward's state (the language terms of dL_dalpha vanish identically), every gradient equals what a zero-filled
 * cotangent gives, and dL_dlanguage / the bucket's language columns are written as zeros.
 *
 * Gradient outputs, all fully overwritten:
 *   dL_dmeans2D[P,3]  dL_dcolors[P,3]  dL_dlanguage[P,F]  dL_dopacity[P]
 *   dL_dmeans3D[P,3]  dL_dcov3D[P,6]   dL_dsh[P,M,3]      dL_dscales[P,3]
 *   dL_drotations[P,4]  dL_dtau[P,6]
 * dL_dconic[P,4] and dL_ddepths[P] are the reference's internal buffers
 * (DGR/rasterize_points.cu:390-391); they may be NULL.  The geometry and binning buffers carry
 * scratch regions the backward writes (the reference passes them as char* too).
 * dL_dtau_sum[6] (may be NULL) receives the sum over P that the Python layer computes at
 * DGR/diff_gaussian_rasterization/__init__.py:383-385.
 *
 * bucket (may be NULL): additionally write (assign != 0) or add (assign == 0) this view's gradients
 * straight into the flat all-reduce buffer and the densification statistics described at
 * olsr_accumulate_gradients — the same result as calling that function afterwards, without the
 * round trip through the separate arrays.  With a bucket, every per-Gaussian output above may be
 * NULL and is then not written (a mapping step only consumes the bucket and dL_dtau_sum).  Likewise with
 * dL_dtau_sum alone: tracking (utils/slam_frontend.py) optimises only the camera pose, so a pose-only backward
 * passes NULL for every per-Gaussian array. */
typedef struct olsr_grad_bucket {
  float *flat;        /* [P][3 xyz | 3M sh | 1 opacity | 3 scale | 4 rotation | F language] */
  float *densify;     /* [P][2]  {sum of ||dL_dmeans2D.xy|| over views, number of views that saw it} */
  int32_t *max_radii; /* [P] */
  int32_t assign;     /* 1: first view of a step (overwrite), 0: add */
  int32_t _pad0;
  uint64_t *row_mask; /* NULL, or device uint64[ceil(P / 64)], in/out: bit g is set when row g of `flat` MAY be non-zero.
                       * Saturation leaves ~98 % of a view's Gaussians without gradient (config 3), and an overwrite that
                       * rewrites their zero rows is 150 MB of stores per view for nothing: with a mask, assign != 0 writes
                       * only the rows that have a gradient now or whose bit was set (those are zeroed), and leaves the mask =
                       * rows with a gradient now; assign == 0 ORs the rows it adds to into the mask.  The result in `flat`
                       * is the same, bit for bit, provided the mask covers every non-zero row when the call is made: start
                       * with all ones (= unknown: the first overwrite is dense), and set it to all ones again after writing
                       * `flat` by any other means (a collective, olsr_accumulate_gradients, a sum of buckets).  `densify` and
                       * `max_radii` are always written for every Gaussian. */
} olsr_grad_bucket;
size_t olsr_backward_scratch_bytes(int64_t rows, int32_t F);
/* Exact scratch rows for the backward of an olsr_forward (the synchronising entry) without a synchronisation.  The
 * forward's last kernel posts the frame's gradient-row counts into mapped host memory; olsr_last_forward_token() names the
 * olsr_forward this thread issued last, olsr_live_rows(token, packed) returns that frame's row count — `packed` != 0: the
 * reference-mode backward of 15x15 tiles (one row per packed survivor wave), else one row per 64-pixel slot — or -1 while
 * the forward has not finished or after 256 later forwards of the thread reused the slot: size the scratch by the bound
 * (2 resp. 4 rows per instance) then.  No reference counterpart (its backward accumulates with atomics). */
int32_t olsr_last_forward_token(void);
int64_t olsr_live_rows(int32_t token, int32_t packed_survivor_waves);
/* The same, for a caller that runs ahead of the GPU (a training loop reaches its backward while the forward is still
 * executing): waits — polling the mapped word, the GPU is busy with the forward meanwhile — until the forward has posted
 * its counts, at most timeout_us microseconds, and not at all when the slot already belongs to a later forward.
 * Returns the row count or -1 (then size by the bound).  The drop-in bindings call it when the bound would cost more than
 * 64 MB of scratch (OLSR_ROWS_WAIT_US, read once at load, default 5000; 0: never wait). */
int64_t olsr_live_rows_wait(int32_t token, int32_t packed_survivor_waves, int32_t timeout_us);
/* 1 when the ring slot of `token` already belongs to a LATER forward (more than 255 forwards were issued before this
 * backward): the count will never be readable, so a caller sizes its scratch by the bound at once instead of guessing and
 * then waiting for a count that cannot arrive. */
int32_t olsr_live_rows_overwritten(int32_t token);
/* The bindings' policy in one call: rows to size the backward scratch of (token, R instances, F) with. */
int64_t olsr_backward_rows(int32_t token, int32_t packed_survivor_waves, int64_t num_rendered, int32_t F);
int olsr_backward(const olsr_scene *scene, const int32_t *radii,
                  void *geometry_buffer, int32_t num_rendered,
                  void *binning_buffer, const void *image_buffer,
                  olsr_alloc_fn scratch_alloc, void *scratch_user,
                  void *scratch, int64_t scratch_rows,
                  const float *dL_dout_color, const float *dL_dout_language,
                  const float *dL_dout_depth,
                  float *dL_dmeans2D, float *dL_dconic, float *dL_dopacity,
                  float *dL_dcolors, float *dL_dlanguage, float *dL_ddepths,
                  float *dL_dmeans3D, float *dL_dcov3D, float *dL_dsh,
                  float *dL_dscales, float *dL_drotations, float *dL_dtau,
                  float *dL_dtau_sum, const olsr_grad_bucket *bucket,
                  int32_t *status_dev, void *hip_stream);

/* ---- optimiser step on the gradient bucket -------------------------------
 * One fused Adam step over all Gaussian parameters, fed by the flat bucket olsr_backward / the all-reduce
 * produced: replaces `self.gaussians.optimizer.step()` — torch.optim.Adam(param_groups, lr=0.0, eps=1e-15) over
 * xyz, f_dc, f_rest, opacity, scaling, rotation, f_language (gaussian_splatting/scene/gaussian_model.py:393-440,
 * utils/slam_backend.py:747-749) — with the same arithmetic (torch/optim/adam.py, single-tensor path, no
 * amsgrad / weight decay) and the same DENSE semantics (rows with zero gradient decay their moments and
 * move).  `step` is the 1-based step count AFTER the increment.  shs is [P,M,3] (k = 0: f_dc, k >= 1:
 * f_rest); parameter arrays hold whatever the caller optimises (raw parameters with scene.activations).
 * exp_avg / exp_avg_sq: [P, 11 + 3M + F] in bucket layout, zero before the first step. */
typedef struct olsr_adam_params {
  /* doubles, like the Python floats torch.optim.Adam holds: every scalar the update uses (1 - beta1, 1 - beta2,
   * the bias corrections, step_size = lr / bias_correction1, sqrt(bias_correction2)) is formed in double on the
   * host and rounded to fp32 once, exactly where torch rounds it */
  double lr_xyz, lr_sh_dc, lr_sh_rest, lr_opacity, lr_scale, lr_rotation, lr_language;
  double beta1, beta2, eps; /* 0.9, 0.999, 1e-15 in the reference */
  int32_t step;
  int32_t _pad0;
} olsr_adam_params;
int olsr_adam_step(int32_t P, int32_t M, int32_t F, const olsr_adam_params *params, const float *flat,
                   float *means3D, float *shs, float *opacities, float *scales, float *rotations,
                   float *language, float *exp_avg, float *exp_avg_sq, void *hip_stream);

/* The same step fed by SEVERAL buckets (1 <= n_flats <= 8, `flats` a host array of device pointers): the gradient of a row
 * is ((flats[0] + flats[1]) + flats[2]) + ... — exactly what adding the buckets into flats[0] first would leave, without
 * the read-modify-write passes over P x width floats (a caller that renders a step's views on several HIP streams keeps one
 * bucket per stream: frame_shard.FrameLanes). */
int olsr_adam_step_sum(int32_t P, int32_t M, int32_t F, const olsr_adam_params *params, int32_t n_flats,
                       const float *const *flats, float *means3D, float *shs, float *opacities, float *scales,
                       float *rotations, float *language, float *exp_avg, float *exp_avg_sq, void *hip_stream);

/* The same step for buckets that carry a row mask (olsr_grad_bucket.row_mask; round 5): row_masks[b] (host array of device
 * pointers, an entry or the array itself may be NULL = "every row may be non-zero") names the rows of flats[b] that may hold
 * a gradient; a row whose bit is clear is NOT READ — its gradient is the +0.0 the row holds by the mask's invariant.  The
 * update stays dense (every parameter and both moments of every row are read and written): parameters and moments equal
 * olsr_adam_step_sum's — torch.optim.Adam's — bit for bit.  What goes is the read of known-zero gradient rows: on a surface
 * map a view leaves 20 % of the rows live, on the i.i.d. volume of SURVEY 8(d) 2 %.  P must start at a multiple of 64 rows
 * of the masks (a block of the kernel owns one mask word). */
int olsr_adam_step_masked(int32_t P, int32_t M, int32_t F, const olsr_adam_params *params, int32_t n_flats,
                          const float *const *flats, const uint64_t *const *row_masks, float *means3D, float *shs,
                          float *opacities, float *scales, float *rotations, float *language, float *exp_avg,
                          float *exp_avg_sq, void *hip_stream);

/* dst bucket += src bucket — the sum of the per-stream buckets of a step before its exchange (what autograd's `.grad +=`
 * over the views of BackEnd.map does, utils/slam_backend.py:510-670) — reading and writing only the gradient rows that
 * src_row_mask says may be non-zero (NULL: all of them); dst_row_mask (may be NULL) |= src_row_mask.  densify (SUM) and
 * max_radii (MAX) are combined for every Gaussian.  Same bits as a dense dst += src (a row left alone is dst + 0.0). */
int olsr_bucket_add(int32_t P, int32_t width, float *dst_flat, float *dst_densify, int32_t *dst_max_radii,
                    uint64_t *dst_row_mask, const float *src_flat, const float *src_densify, const int32_t *src_max_radii,
                    const uint64_t *src_row_mask, void *hip_stream);

/* ---- the reference's other native dependency ----------------------------
 * mean_dist2[i] = mean of the squared distances from point i to its 3 nearest neighbours (FLT_MAX counts
 * for a missing neighbour when P < 4).  Replaces simple_knn._C.distCUDA2 -> SimpleKNN::knn
 * (submodules/simple-knn/spatial.cu:15-26, simple_knn.cu:185-221), used to initialise Gaussian scales
 * (gaussian_splatting/scene/gaussian_model.py:256-263).  points[P,3], mean_dist2[P], scratch of
 * olsr_knn_scratch_bytes(P) bytes; everything stays on the device (the reference copies the bounding box
 * to the host twice). */
size_t olsr_knn_scratch_bytes(int32_t P);
int olsr_knn_mean_dist2(int32_t P, const float *points, float *mean_dist2, void *scratch, void *hip_stream);

/* ---- caller side of the path -----------------------------------------
 * Mapping loss of one view and its gradient with respect to the rendered images, in one pass over the
 * pixels.  Replaces, with their autograd backward,
 *   get_loss_mapping / get_loss_mapping_rgbd                        utils/slam_utils.py:124-165
 *   F.interpolate(gt_lang_feat, size=(H, W), mode='bilinear', align_corners=False)
 *   l1_loss(language_feat, gt_lang_feat_resize), lamda_lang weighting   utils/slam_backend.py:579-597
 *                                                        gaussian_splatting/utils/loss_utils.py:21-22
 *   loss = alpha * mean|m_rgb * (exp(a) * image + b) - m_rgb * gt_image|
 *        + (1 - alpha) * mean|m_d * depth - m_d * gt_depth|
 *        + lamda_lang * mean|language - resize(gt_language)|
 * with m_rgb = (sum_c gt_image > rgb_boundary_threshold), m_d = (gt_depth > 0.01), means over all
 * elements of the respective tensor.  The outputs dL_dimage[3,H,W], dL_ddepth[1,H,W] and
 * dL_dlanguage[F,H,W] are exactly the cotangents olsr_backward consumes, so the rendered images
 * make one round trip instead of the ~20 elementwise kernels of the PyTorch formulation.
 *   exposure      device float[2] {exposure_a, exposure_b}, or NULL / initialization != 0 for a = b = 0
 *   gt_language   [F, lang_height, lang_width] or NULL (no language term; dL_dlanguage zero-filled)
 *   loss          device float[4]: {total, rgb term, depth term, language term} (each already weighted)
 *   dL_dexposure  device float[2] {dL/da, dL/db} or NULL
 *   scratch       olsr_mapping_loss_scratch_bytes(width, height) bytes */
typedef struct olsr_loss_params {
  int32_t width, height;
  int32_t F;                       /* language channels (0: none) */
  int32_t lang_width, lang_height; /* size of gt_language (192 x 192 in the reference) */
  int32_t initialization;          /* != 0: skip the exposure transform (get_loss_mapping, :125-126) */
  float alpha;                     /* config["Training"]["alpha"], default 0.95 */
  float rgb_boundary_threshold;    /* config["Training"]["rgb_boundary_threshold"] */
  float lamda_lang;                /* BackEnd.lamda_lang (1.0), utils/slam_backend.py:80 */
  int32_t _pad0;
} olsr_loss_params;
size_t olsr_mapping_loss_scratch_bytes(int32_t width, int32_t height);
int olsr_mapping_loss(const olsr_loss_params *params, const float *image, const float *depth,
                      const float *language, const float *gt_image, const float *gt_depth,
                      const float *gt_language, const float *exposure,
                      float *dL_dimage, float *dL_ddepth, float *dL_dlanguage,
                      float *loss, float *dL_dexposure, void *scratch, void *hip_stream);

/* ---- the loss in the forward composite's epilogue ------------------------------------------
 * olsr_forward_async with olsr_mapping_loss / olsr_tracking_loss evaluated where the composite kernel still holds every
 * pixel's colour, depth, language features and transmittance in registers (the equivalent of CR/forward.cu:490-512): the
 * kernel writes the cotangents olsr_backward consumes (dL_dimage, dL_ddepth, dL_dlanguage) next to — or, with skip_images,
 * instead of — the images, and one partial sum per tile that a one-block kernel turns into loss[4] / dL_dexposure[2].
 * The rendered images make NO round trip: the separate loss kernel re-reads 4 (4 + F) bytes per pixel and writes as many.
 * Same arithmetic per pixel as the two stand-alone entries (shared source, csrc/olsr_loss_device.h): the cotangents are
 * bit-identical to theirs, the loss differs in the summation order only (per-tile instead of per-256-pixel partials).
 *   params        width / height must equal the scene's; params.F: language channels of the loss term — the scene's F, or 0
 *                 for "no language term" (then dL_dlanguage is not written: hand olsr_backward a NULL language cotangent)
 *   tracking      0: the mapping loss;  != 0: the tracking loss (opacity-weighted, grad_mask; params.F is ignored)
 *   skip_images   != 0: out_color / out_language / out_depth / out_opacity are not written and may be NULL
 *   gt_language   [F, lang_height, lang_width] or NULL (no language term)
 *   scratch       olsr_fused_loss_scratch_bytes(width, height, tile) bytes
 * Every normaliser of these losses is the pixel count (means over all elements): nothing has to be precomputed per keyframe.
 * Supported with the default forward accumulation (not with OLSR_FLAG_FWD_ACCUM_MFMA / _WEIGHT). */
typedef struct olsr_loss_fusion {
  olsr_loss_params params;
  int32_t tracking;
  int32_t skip_images;
  const float *gt_image;    /* [3,H,W] */
  const float *gt_depth;    /* [H,W] */
  const float *gt_language; /* [F,lang_height,lang_width] or NULL */
  const float *exposure;    /* device float[2] or NULL */
  const float *grad_mask;   /* [H,W] float or NULL (tracking only) */
  float *dL_dimage;         /* [3,H,W] */
  float *dL_ddepth;         /* [H,W] */
  float *dL_dlanguage;      /* [F,H,W]; required when a language term is evaluated */
  float *loss;              /* device float[4] {total, rgb, depth, language} */
  float *dL_dexposure;      /* device float[2] or NULL */
  void *scratch;
} olsr_loss_fusion;
size_t olsr_fused_loss_scratch_bytes(int32_t width, int32_t height, int32_t tile);
int olsr_forward_async_loss(const olsr_scene *scene,
                            void *geometry_buffer, void *binning_buffer, int64_t capacity,
                            void *image_buffer,
                            float *out_color, float *out_language, float *out_depth, float *out_opacity,
                            int32_t *radii, int32_t *n_touched,
                            int32_t *num_rendered_dev, uint32_t *tile_order_inout,
                            const olsr_loss_fusion *loss, void *hip_stream);

/* Tracking loss of one view (front end: the pose is optimised, the Gaussians are fixed) and its image cotangents.
 * Replaces get_loss_tracking / get_loss_tracking_rgb / get_loss_tracking_rgbd (utils/slam_utils.py:92-121):
 *   loss = alpha * mean(opacity * |m * (exp(a) * image + b) - m * gt_image|)
 *        + (1 - alpha) * mean|dm * depth - dm * gt_depth|
 * m = (sum_c gt_image > rgb_boundary_threshold) * grad_mask (grad_mask[H,W] float, NULL = all ones),
 * dm = (gt_depth > 0.01) * (opacity > 0.95).  params->F, lang_* and lamda_lang are ignored.
 * loss = device float[4] {total, rgb term, depth term, 0}.  No gradient is produced for `opacity`: the
 * rasterizer's autograd function discards the cotangent of its opacity output
 * (DGR/diff_gaussian_rasterization/__init__.py:333-343), so the reference's backward drops it as well.
 * Same scratch as olsr_mapping_loss.  Follow with olsr_backward in pose-only mode (dL_dtau_sum alone). */
int olsr_tracking_loss(const olsr_loss_params *params, const float *image, const float *depth,
                       const float *opacity, const float *gt_image, const float *gt_depth,
                       const float *grad_mask, const float *exposure,
                       float *dL_dimage, float *dL_ddepth, float *loss, float *dL_dexposure,
                       void *scratch, void *hip_stream);

/* ---- one tracking iteration's pose update (the front end) ----------------------
 * Replaces, per iteration of the reference's tracking loop (utils/slam_frontend.py:216-243),
 *   pose_optimizer.step()           torch.optim.Adam over cam_rot_delta (lr config Training.lr.cam_rot_delta = 0.003),
 *                                   cam_trans_delta (0.001), exposure_a / exposure_b (0.01); eps 1e-8, betas (0.9, 0.999)
 *   converged = update_pose(cam)    utils/pose_utils.py:79-97: tau = [cam_trans_delta | cam_rot_delta],
 *                                   new_w2c = SE3_exp(tau) @ T_w2c (:61-76), converged = |tau| < 1e-4, deltas zeroed
 * and the camera properties the next render reads (utils/camera_utils.py:103-117): world_view_transform = W2C^T,
 * full_proj_transform = world_view_transform @ projection_matrix, camera_center = world_view_transform^-1 [3, :3]
 * — one launch instead of ~40 one-element PyTorch kernels and a host read-back between two dependent renders.
 * Tolerance: Adam, SE3_exp and the convergence test follow the reference operation for operation; the view matrix is formed
 * directly as W2C^T, where the reference's Camera goes through getWorld2View2 (gaussian_splatting/utils/graphics_utils.py:
 * 33-46), which inverts [R|t] twice, and update_pose re-reads R and T from that.  T_w2c and the matrices therefore agree with
 * the reference to fp32 rounding per iteration (tests/test_gpu_pose.py holds them to 5e-7 x the iteration number), not
 * bit for bit.
 *   dL_dtau_sum   device float[6] = [rho | theta] as olsr_backward leaves it (rho: gradient of cam_trans_delta,
 *                 theta: of cam_rot_delta), or NULL: no step, only the matrices of the current pose are (re)derived
 *   dL_dexposure  device float[2] from olsr_tracking_loss, or NULL (exposure not optimised)
 *   projection_matrix  device float[16], the P^T the callers hold (Camera.projection_matrix)
 *   state         device float[80], all zero before the first call except T_w2c:
 *                 [0,16) T_w2c row-major (in/out) | [16,32) world_view_transform | [32,48) full_proj_transform |
 *                 [48,51) camera_center | [52,58) exp_avg of tau | [58,64) exp_avg_sq | [64,70) tau this step applied |
 *                 [70,72) exposure a, b (in/out) | [72,74) their exp_avg | [74,76) exp_avg_sq
 *   status        device int32[2]: {converged flag of this step, steps done}
 * params->step is the 1-based Adam step count of this call; step <= 0: the count is status[1] + 1, kept on the device (the
 * launch is then the same every iteration and can be replayed from a HIP graph; reset status to restart the optimiser). */
typedef struct olsr_pose_params {
  double lr_rot, lr_trans, lr_exposure;
  double beta1, beta2, eps;  /* torch.optim.Adam defaults in the reference: 0.9, 0.999, 1e-8 */
  double converged_threshold; /* 1e-4 */
  int32_t step;
  int32_t _pad0;
} olsr_pose_params;
int olsr_pose_step(const olsr_pose_params *params, const float *dL_dtau_sum, const float *dL_dexposure,
                   const float *projection_matrix, float *state, int32_t *status, void *hip_stream);
/* olsr_pose_step, skipped on the device when the frame its gradient came from was not usable:
 *   frame_status  device int32[2], the num_rendered_dev of that frame's olsr_forward_async[_loss] (or NULL: always step).
 * frame_status[1] != 0 (OLSR_STATUS_OVERFLOW / SYNC_ERROR / CUT_MISS): no optimiser step — state [0,16) and [52,76) and
 * status[1] keep their values, status[0] = 0 — only the matrices are re-derived, as with dL_dtau_sum == NULL. */
int olsr_pose_step_gated(const olsr_pose_params *params, const float *dL_dtau_sum, const float *dL_dexposure,
                         const float *projection_matrix, float *state, int32_t *status, const int32_t *frame_status,
                         void *hip_stream);

/* Adds one view's per-Gaussian gradients into the flat fp32 buffer
 *   flat[P][3 xyz | 3M sh | 1 opacity | 3 scale | 4 rotation | F language]
 * that a frame-sharded trainer all-reduces once per optimisation step, and updates the
 * densification statistics densify[P][2] += {||dL_dmeans2D.xy||, 1} for visible Gaussians and
 * max_radii[P] = max(max_radii, radii).  With assign != 0 the accumulators are overwritten
 * instead (first view of a step; saves zero-filling them).  No reference counterpart in native code: it is what
 * autograd's `.grad +=` over the views of BackEnd.map (utils/slam_backend.py:510-670) and
 * GaussianModel.add_densification_stats (gaussian_splatting/scene/gaussian_model.py:965-969) do
 * with separate PyTorch kernels. */
int olsr_accumulate_gradients(int32_t P, int32_t M, int32_t F, int32_t assign,
                              const float *dL_dmeans3D,
                              const float *dL_dsh, const float *dL_dopacity,
                              const float *dL_dscales, const float *dL_drotations,
                              const float *dL_dlanguage, const float *dL_dmeans2D,
                              const int32_t *radii, float *flat, float *densify,
                              int32_t *max_radii, void *hip_stream);

/* ---- the capacity-bound sparse exchange of the bucket (SURVEY.md section 8, rows e and f2) ---------------------------
 * Saturation leaves ~2 % of a view's Gaussians with a gradient row (config 3: 9 471 of 500 000), so a frame-sharded
 * step exchanges the UNION of the ranks' non-zero rows instead of the whole bucket — without a host synchronisation,
 * so that several frames stay in flight.  The caller issues two collectives (RCCL through torch.distributed, or any
 * other all-reduce) and these three calls around them, all on one stream:
 *   olsr_sparse_exchange_mask    imax[0..P) = 1 where row g of `flat` holds a non-zero element (only rows whose bit
 *                                of row_mask is set are looked at; row_mask may be NULL = look at every row),
 *                                imax[P..2P) = max_radii
 *   -- all-reduce MAX over imax (int32[2P]): the union of the rows, and the reduced radii --
 *   olsr_sparse_exchange_pack    max_radii <- imax[P..2P); row_mask (if given) <- the union; the union's rows, in
 *                                ascending row order, are copied into fsum[slot][width] for slot < capacity, unused
 *                                slots are zero-filled, idx[slot] = the row (P for an unused slot);
 *                                fsum[capacity * width ..) <- densify[P][2]; status_dev = {rows in the union,
 *                                1 if they did not fit};  scratch: olsr_sparse_exchange_scratch_ints(P) int32.
 *                                With idx == fsum == NULL nothing is packed: the call only counts (status_dev, max_radii
 *                                and row_mask as above) — for a caller that reads the count back and sizes the packed
 *                                buffer exactly (FrameShardedStep's "sparse" exchange, one host synchronisation per step)
 *   -- all-reduce SUM over fsum (fp32[capacity * width + 2P]) --
 *   olsr_sparse_exchange_unpack  flat[idx[slot]] <- fsum[slot], densify <- the tail
 * The bucket then holds what a dense all-reduce of {flat, densify} (SUM) and max_radii (MAX) would have left, bit for
 * bit, provided the union fitted; on overflow only the first `capacity` rows of the union were exchanged and the
 * caller must repeat the step with a larger capacity or densely (the contract of an instance overflow).  `width` is
 * the bucket's row width 11 + 3M + F.  No reference counterpart: the reference is single-GPU and autograd's `.grad`
 * accumulates the views of BackEnd.map (utils/slam_backend.py:510-670) in one process. */
int olsr_sparse_exchange_mask(int32_t P, int32_t width, const float *flat, const uint64_t *row_mask,
                              const int32_t *max_radii, int32_t *imax, void *hip_stream);
int64_t olsr_sparse_exchange_scratch_ints(int32_t P);
int olsr_sparse_exchange_pack(int32_t P, int32_t width, int32_t capacity, const float *flat, const int32_t *imax,
                              int32_t *max_radii, uint64_t *row_mask, const float *densify, int32_t *idx,
                              float *fsum, int32_t *scratch, int32_t *status_dev, void *hip_stream);
int olsr_sparse_exchange_unpack(int32_t P, int32_t width, int32_t capacity, const int32_t *idx, const float *fsum,
                                float *flat, float *densify, void *hip_stream);

/* Near-plane visibility test.  Replaces markVisible / checkFrustum
 * (DGR/rasterize_points.cu:457-476; CR/rasterizer_impl.cu:54-66,141-153).
 * present[P] is one byte per Gaussian (bool). */
int olsr_mark_visible(int32_t P, const float *means3D, const float *viewmatrix,
                      const float *projmatrix, uint8_t *present, void *hip_stream);

/* Introspection of the opaque state for stage-by-stage parity tests and for the
 * benchmark's byte model.  Each returns a device pointer inside the given buffer
 * (or NULL for an unknown name).  Names: geometry — "depths" f32[P], "means2D"
 * f32[P,2], "cov3D" f32[P,6], "conic_opacity" f32[P,4], "rgb" f32[P,3], "clamped"
 * u8[P,3], "tiles_touched" u32[P], "depth_order" u32[P]; binning — "src" u32[R]
 * (sorted position -> emission index), "inst_gid" u32[R] (emission index -> Gaussian id; the
 * reference's point_list is inst_gid[src]), "flags" u8[R], "rowbase" u32[R+1], "row_sync" u32[2] (ticket and done-counter of the
 * row compaction, zero between kernels); image — "final_T" f32[H*W], "n_contrib" u32[H*W], "ranges" u32[tiles,2]. */
const void *olsr_geometry_field(const void *geometry_buffer, int32_t P, int32_t F, const char *name);
const void *olsr_binning_field(const void *binning_buffer, int64_t num_rendered, int32_t F,
                               const char *name);
const void *olsr_image_field(const void *image_buffer, int32_t width, int32_t height, int32_t tile,
                             const char *name);

/* Per-stage HIP-event timing (events recorded on the stream the kernels run on).
 * olsr_set_profiling(1) clears the log and starts recording one event per stage of every
 * forward/backward issued by this thread (no synchronisation is added);
 * olsr_get_stage_times waits for the last event and returns, in issue order, one
 * (stage name, milliseconds) entry per stage executed since then (up to `max`). */
void olsr_set_profiling(int enable);
int olsr_get_stage_times(const char **names, float *ms, int max);

/* Debug aid (kernel tuning): while a device buffer of max_launches * max_blocks * 8 uint64 is set, every block of the
 * next radix-sort passes issued by this thread records the shader clock at its phase boundaries ({ticket, keys
 * counted, counts published, ranked, predecessors summed, written}; index (launch * max_blocks + block) * 8 + phase).
 * NULL switches it off. */
void olsr_debug_sort_timing(unsigned long long *device_buffer, int max_blocks, int max_launches);

/* Host logic of the radix passes, exposed for tests: the keys-per-thread and the number of 1024-thread blocks a sort of
 * n keys is launched with (n_is_capacity != 0: n bounds a count only known on the device, see olsr_forward_async), and
 * whether n is sorted by the one-kernel passes at all (return value: 1) or by the multi-kernel fallback (0). */
int olsr_debug_sort_plan(int64_t n, int n_is_capacity, int32_t *keys_per_thread, int32_t *blocks);

/* Test hook for the synchronisation-error path (OLSR_STATUS_SYNC_ERROR), process-wide.  fault_bits: bit 0 / bit 1 = in every
 * later forward the block holding ticket 0 of the first depth-sort / tile-sort pass never publishes its digit counts, which is
 * what its successors see when a status word is lost; spin_limit: polls a look-back makes before it gives up (0 restores the
 * default, 2^22 — seconds; tests use a few thousand).  A negative argument leaves that knob as it is. */
void olsr_debug_sync_fault(int fault_bits, int spin_limit);

/* Tuning / test knobs of the radix passes, process-wide.  keys_per_thread in {2, 4, 8, 12, 16} pins the instantiation (0:
 * chosen from the input size); resident_blocks > 0 replaces the 256 blocks a round is planned for; legacy != 0 forces the
 * multi-kernel passes at any size.  A negative argument leaves that knob as it is.  The library reads the environment
 * (OLSR_SORT_KPT, OLSR_SORT_RESIDENT, OLSR_SORT_LEGACY) ONCE, when it is loaded, to seed them; nothing on a call path
 * touches getenv.  Not for use while frames are in flight (a forward and its backward must see the same plan). */
void olsr_debug_sort_knobs(int keys_per_thread, int resident_blocks, int legacy);

/* The depth sort of at most 8 192 Gaussians is ONE launch of one workgroup (histogram, the frame's bookkeeping and the four
 * 8-bit passes inside the block; round 5) instead of a histogram launch and four radix passes with a ~10 us floor each; same
 * order bit for bit.  enable = 0 sends small frames through the pass kernels as well (tests compare the two), 1 restores the
 * default, negative leaves it.  Process-wide; seeded once from OLSR_SORT_SMALL.  A keys_per_thread pinned through
 * olsr_debug_sort_knobs also selects the pass kernels. */
void olsr_debug_sort_small(int enable);

/* Visible-set compaction of the depth sort (round 6): the sort orders only the Gaussians that emit instances — the histogram
 * kernel, which reads every key anyway, writes their (key, index) densely in index order and counts digits of those only; the
 * passes sort that many keys.  Same lists bit for bit.  enable = 0 sorts every Gaussian as rounds 1-5 did (tests compare the
 * two), 1 restores the default, negative leaves it.  Process-wide; seeded once from OLSR_SORT_COMPACT.  Not used with a carried
 * depth order (whose domain is every Gaussian), by the one-launch sort of <= 8 192 Gaussians, nor beyond 2 M Gaussians. */
void olsr_debug_sort_compact(int enable);

/* Diagnostic: while a device buffer of 2 x capacity uint64 is set, every forward / backward composite launch of this process
 * is bracketed by two one-thread kernels on its stream that write {device wall clock (100 MHz), stream << 8 | kind} — kind 0 / 1:
 * in front of / behind the forward composite (+ its tile-order kernel), 2 / 3: the backward composite — into consecutive
 * entries.  Reads the overlap of several frames in flight without a profiler (scripts/probe/composite_overlap.py).  The stamps
 * cost four tiny launches per frame.  NULL switches it off. */
void olsr_debug_composite_stamps(unsigned long long *device_buffer, int capacity);

/* Test instrument (round 6; never on a product path): the composite backward — CR/backward.cu:932-1201 (language_render_cuda),
 * 706-930 (renderCUDA), 684-702 (render_cuda_reduce_sum) — restated on the GPU in the reference's OWN association: its per-lane
 * state and expressions in source order, the 225-lane integer-halving tree through shared memory in the tree's own pairing
 * (OLSR_BWD_REFERENCE; lane order in double for OLSR_BWD_EXACT, like the oracle), one row per instance the tile does not skip,
 * a Gaussian's rows added one after the other in sorted-list order.  It runs on the state buffers a forward left (either
 * entry; num_rendered as for olsr_backward) and writes the composite-level gradients only: dL_dmeans2D [P,3], dL_dconic [P,4],
 * dL_dopacity [P], dL_dcolors [P,3], dL_dlanguage [P,F], dL_ddepths [P].  These EQUAL the CPU oracle's bit for bit
 * (tests/test_gpu_bwd_ordered.py); the product's fast kernel, which re-associates the same sums, is compared with this one on the
 * GPU at sizes the oracle needs minutes for.  scene->bwd_mode and tile select the variant; a NULL language / depth cotangent
 * counts as zeros.  scratch: olsr_debug_backward_ordered_scratch_bytes(num_rendered, F) bytes.  Tens of milliseconds per frame.
 * condition != 0: the outputs are not the gradients but their CONDITION A — the same sums with every product replaced by the
 * product of magnitudes and every difference by the sum of magnitudes.  To first order any re-association of the sums and
 * any few-ulp variation of the value path moves an element by at most K x 2^-24 x A, K of the order of the operations a term
 * went through; the tests hold the fast kernel to that, element by element ("it differs by rounding only"). */
size_t olsr_debug_backward_ordered_scratch_bytes(int64_t num_rendered, int32_t F);
int olsr_debug_backward_ordered(const olsr_scene *scene, const void *geometry_buffer, int32_t num_rendered,
                                const void *binning_buffer, const void *image_buffer,
                                const float *dL_dout_color, const float *dL_dout_language, const float *dL_dout_depth,
                                void *scratch,
                                float *dL_dmeans2D, float *dL_dconic, float *dL_dopacity, float *dL_dcolors,
                                float *dL_dlanguage, float *dL_ddepths, int32_t condition, void *hip_stream);

/* Test / experiment knob: threads per workgroup of the radix passes and their histogram kernels.  0 (default): the call
 * decides — 1024, or 256 for a scene that carries OLSR_FLAG_FRAMES_IN_FLIGHT; 256 / 1024: forced for every later forward.
 * Any other argument only reads the value back.  Same lists bit for bit.  Process-wide; seeded once from OLSR_SORT_THREADS.
 * Returns the value in force. */
int olsr_debug_sort_threads(int threads);

const char *olsr_last_error(void);
const char *olsr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* OLSR_H_INCLUDED */
