// k_accumulate.hip — accumulate one view's per-Gaussian gradients into the flat buffer that the
// frame-sharded step all-reduces, plus the densification statistics.
//
// Counterpart of what autograd's `.grad +=` over the 12 views of a mapping iteration and
// GaussianModel.add_densification_stats (gaussian_splatting/scene/gaussian_model.py:965-969) do
// with ~10 separate elementwise PyTorch kernels; here it is one pass: every gradient array is
// read once, the flat row [3 xyz | 3M sh | 1 opacity | 3 scale | 4 rot | F lang] is read-modify-
// written once (or just written, for the first view of a step: `assign`).
#include <algorithm>

#include "olsr_device.h"
#include "olsr_kernels.h"

namespace olsr {

// A block covers ACC_G consecutive Gaussians = ACC_G * width consecutive floats of `flat`, so the
// read-modify-write of `flat` is perfectly coalesced; (gaussian, column) of an element come from a
// small-range division done in fp32 (exact for e < 2^22).
constexpr int ACC_G = 64;

template <bool ASSIGN>
__global__ __launch_bounds__(256) void accumulate_kernel(
    int P, int M, int F, int width, const float* __restrict__ dmeans3D, const float* __restrict__ dsh,
    const float* __restrict__ dopacity, const float* __restrict__ dscales, const float* __restrict__ drot,
    const float* __restrict__ dlang, const float* __restrict__ dmeans2D, const int32_t* __restrict__ radii,
    float* __restrict__ flat, float* __restrict__ densify, int32_t* __restrict__ max_radii) {
  const int g0 = blockIdx.x * ACC_G;
  const int ng = min(ACC_G, P - g0);
  const int count = ng * width;
  const int sh_w = 3 * M;
  const float inv_w = 1.0f / (float)width;
  float* out = flat + (size_t)g0 * width;
  for (int e = threadIdx.x; e < count; e += 256) {
    const int gl = (int)(((float)e + 0.5f) * inv_w);
    const int c = e - gl * width;
    const int g = g0 + gl;
    float v;
    if (c < 3) v = dmeans3D[3 * (size_t)g + c];
    else if (c < 3 + sh_w) v = dsh[(size_t)g * sh_w + (c - 3)];
    else if (c < 4 + sh_w) v = dopacity[g];
    else if (c < 7 + sh_w) v = dscales[3 * (size_t)g + (c - 4 - sh_w)];
    else if (c < 11 + sh_w) v = drot[4 * (size_t)g + (c - 7 - sh_w)];
    else v = dlang[(size_t)g * F + (c - 11 - sh_w)];
    out[e] = ASSIGN ? v : out[e] + v;
  }
  if (threadIdx.x < ng) {
    const int g = g0 + threadIdx.x;
    const int r = radii[g];
    const bool vis = r > 0;
    const float gx = dmeans2D[3 * (size_t)g], gy = dmeans2D[3 * (size_t)g + 1];
    const float nrm = vis ? sqrtf(gx * gx + gy * gy) : 0.f;  // ||viewspace grad||, taken per view
    const float cnt = vis ? 1.f : 0.f;
    float2* dz = reinterpret_cast<float2*>(densify) + g;
    if (ASSIGN) {
      *dz = make_float2(nrm, cnt);
      max_radii[g] = r;
    } else {
      const float2 o = *dz;
      *dz = make_float2(o.x + nrm, o.y + cnt);
      max_radii[g] = max(max_radii[g], r);
    }
  }
}

void launch_accumulate(int P, int M, int F, bool assign, const float* dmeans3D, const float* dsh,
                       const float* dopacity, const float* dscales, const float* drot, const float* dlang,
                       const float* dmeans2D, const int32_t* radii, float* flat, float* densify, int32_t* max_radii,
                       hipStream_t st) {
  if (P <= 0) return;
  const int width = 11 + 3 * M + F;
  const unsigned nb = (unsigned)((P + ACC_G - 1) / ACC_G);
  if (assign)
    accumulate_kernel<true><<<nb, 256, 0, st>>>(P, M, F, width, dmeans3D, dsh, dopacity, dscales, drot, dlang, dmeans2D,
                                                radii, flat, densify, max_radii);
  else
    accumulate_kernel<false><<<nb, 256, 0, st>>>(P, M, F, width, dmeans3D, dsh, dopacity, dscales, drot, dlang,
                                                 dmeans2D, radii, flat, densify, max_radii);
}

// ---- dst bucket += src bucket, reading only the rows src's row mask says may be non-zero ------------------------------
// The sum of the lane buckets of a step (frame_shard.FrameLanes: one bucket per HIP stream) before an exchange.  torch's
// dst.add_(src) reads and writes P x width floats whatever they hold; with row masks (olsr_grad_bucket.row_mask) a block
// of 64 Gaussians whose src word is zero touches neither bucket, and inside a block only the flagged rows move:
// dst[g] + 0.0 == dst[g] for the rows left alone.  dst_mask |= src_mask.  The densification statistics and radii are dense
// (every visible Gaussian has them): densify += , max_radii = max.
__global__ __launch_bounds__(256) void bucket_add_kernel(int P, int width, float* __restrict__ dst,
                                                         const float* __restrict__ src,
                                                         unsigned long long* __restrict__ dst_mask,
                                                         const unsigned long long* __restrict__ src_mask,
                                                         float* __restrict__ dst_densify,
                                                         const float* __restrict__ src_densify,
                                                         int32_t* __restrict__ dst_radii,
                                                         const int32_t* __restrict__ src_radii) {
  static_assert(ACC_G == 64, "one row-mask word per block");
  const int g0 = blockIdx.x * ACC_G;
  const int ng = min(ACC_G, P - g0);
  if (threadIdx.x < ng) {
    const int g = g0 + threadIdx.x;
    float2* dz = reinterpret_cast<float2*>(dst_densify) + g;
    const float2 a = *dz, b = reinterpret_cast<const float2*>(src_densify)[g];
    *dz = make_float2(a.x + b.x, a.y + b.y);
    dst_radii[g] = max(dst_radii[g], src_radii[g]);
  }
  const unsigned long long w = src_mask ? src_mask[blockIdx.x] : ~0ull;
  if (w == 0ull) return;
  if (threadIdx.x == 0 && dst_mask) dst_mask[blockIdx.x] |= w;
  const int count = ng * width;
  const float inv_w = 1.0f / (float)width;
  const size_t base = (size_t)g0 * width;
  for (int e = threadIdx.x; e < count; e += 256) {
    const int gl = (int)(((float)e + 0.5f) * inv_w);
    if ((w >> gl) & 1ull) dst[base + e] = dst[base + e] + src[base + e];
  }
}

void launch_bucket_add(int P, int width, float* dst, const float* src, unsigned long long* dst_mask,
                       const unsigned long long* src_mask, float* dst_densify, const float* src_densify,
                       int32_t* dst_radii, const int32_t* src_radii, hipStream_t st) {
  if (P <= 0) return;
  bucket_add_kernel<<<(unsigned)((P + ACC_G - 1) / ACC_G), 256, 0, st>>>(P, width, dst, src, dst_mask, src_mask, dst_densify,
                                                                        src_densify, dst_radii, src_radii);
}


// ---- the capacity-bound sparse exchange of the bucket (frame_shard.py: sparse_all_reduce_capped) ----
// Saturation leaves ~2 % of a view's Gaussians with a gradient row, so a frame-sharded step exchanges the UNION of the
// ranks' non-zero rows instead of the bucket: collective 1 (MAX over int32 [row flags | max_radii]) makes the union known
// to every rank, collective 2 (SUM over fp32 [capacity packed rows | densification statistics]) carries the rows.  The
// kernels below are the local work around those two collectives — four small launches per step where round 4's first
// form spent ~15 PyTorch kernels (a scan of the whole 62 MB bucket among them).  No reference counterpart (the reference
// is single-GPU: autograd's .grad over the views of BackEnd.map, utils/slam_backend.py:510-670).
constexpr int EX_ROWS = 1024;  // rows per block of the count / pack kernels: 256 threads x 4, each wave on 64 consecutive rows

// imax[g] = 1 when row g of `flat` holds a non-zero element, imax[P + g] = max_radii[g].  With a row mask (bit g: the row
// MAY be non-zero) only flagged rows are looked at; a row's first element usually decides.
__global__ __launch_bounds__(256) void exchange_mask_kernel(int P, int width, const float* __restrict__ flat,
                                                            const unsigned long long* __restrict__ row_mask,
                                                            const int32_t* __restrict__ max_radii, int32_t* __restrict__ imax) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= P) return;
  int nz = 0;
  if (row_mask == nullptr || ((row_mask[g >> 6] >> (g & 63)) & 1ull)) {
    const float* r = flat + (size_t)g * width;
    for (int c = 0; c < width; ++c)
      if (r[c] != 0.0f) {  // (NaN counts as non-zero, -0 as zero: what `flat != 0` says)
        nz = 1;
        break;
      }
  }
  imax[g] = nz;
  imax[(size_t)P + g] = max_radii[g];
}

// after collective 1: counts[b] = rows of the union in block b's 1024 rows; max_radii <- the reduced radii; the row mask
// becomes the union (exactly the rows that can be non-zero once the packed rows are scattered back)
__global__ __launch_bounds__(256) void exchange_count_kernel(int P, const int32_t* __restrict__ imax,
                                                             int32_t* __restrict__ max_radii,
                                                             unsigned long long* __restrict__ row_mask,
                                                             int32_t* __restrict__ counts) {
  __shared__ int s_cnt[4];
  const int tid = threadIdx.x, w = tid >> 6;
  int c = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int g = blockIdx.x * EX_ROWS + k * 256 + tid;
    const bool on = g < P && imax[g] != 0;
    if (g < P) max_radii[g] = imax[(size_t)P + g];
    const u64 m = ballot(on);
    c += __popcll(m);
    if ((tid & 63) == 0 && row_mask != nullptr && g < P) row_mask[g >> 6] = m;  // (lane 0: g is a multiple of 64)
  }
  if ((tid & 63) == 0) s_cnt[w] = c;
  __syncthreads();
  if (tid == 0) counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// slot of every union row = its rank among the union's rows (ascending row index); rows with a slot < cap are copied
// into packed[slot]; the slots nobody took are zero-filled and point at no row (idx = P); tail <- densify;
// status = {rows in the union, overflow}
__global__ __launch_bounds__(256) void exchange_pack_kernel(int P, int width, int cap, int nb, const float* __restrict__ flat,
                                                            const int32_t* __restrict__ imax,
                                                            const int32_t* __restrict__ counts,
                                                            const float* __restrict__ densify, int32_t* __restrict__ idx,
                                                            float* __restrict__ packed, float* __restrict__ tail,
                                                            int32_t* __restrict__ status) {
  __shared__ int s_red[2][4];
  __shared__ int s_pop[16];
  __shared__ int s_g[EX_ROWS];  // the block's selected rows in slot order: local rank -> row
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, b = blockIdx.x;
  // rows of the union in front of this block, and in total
  int before = 0, total = 0;
  for (int i = tid; i < nb; i += 256) {
    const int c = counts[i];
    total += c;
    before += (i < b) ? c : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    before += __shfl_xor(before, o);
    total += __shfl_xor(total, o);
  }
  if (lane == 0) {
    s_red[0][w] = before;
    s_red[1][w] = total;
  }
  u64 m[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int g = b * EX_ROWS + k * 256 + tid;
    m[k] = ballot(g < P && imax[g] != 0);
    if (lane == 0) s_pop[k * 4 + w] = __popcll(m[k]);
  }
  __syncthreads();
  before = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
  total = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int pre = before;
    for (int i = 0; i < k * 4 + w; ++i) pre += s_pop[i];
    if ((m[k] >> lane) & 1ull) {
      const int slot = pre + __popcll(m[k] & ((1ull << lane) - 1ull));
      if (slot < cap) {
        const int g = b * EX_ROWS + k * 256 + tid;
        idx[slot] = g;
        s_g[slot - before] = g;  // (the block's slots are consecutive: before, before + 1, ...)
      }
    }
  }
  __syncthreads();
  // the block's rows land in consecutive slots: packed[before * width ...) is ONE contiguous run of nsel * width floats.
  // One element per thread and step — independent loads, perfectly coalesced stores.  (Round 4 gave every selected row a
  // wave: with 20 % of the rows live — a surface map — a wave copied ~50 rows one after the other, each a dependent
  // load -> store: 75 us for 11 MB.)
  int in_block = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) in_block += s_pop[i];
  const int nsel = max(0, min(in_block, cap - before));
  {
    const float inv_w = 1.0f / (float)width;
    float* dst = packed + (size_t)before * width;
    const int count = nsel * width;
    for (int e = tid; e < count; e += 256) {
      const int i = (int)(((float)e + 0.5f) * inv_w);  // (exact below 2^22 elements: nsel <= 1024, width <= 107)
      const int c = e - i * width;
      dst[e] = flat[(size_t)s_g[i] * width + c];
    }
  }
  // the slots behind the union: zeros, so that the SUM leaves zeros and the scatter skips them
  const int used = min(total, cap);
  const size_t gtid = (size_t)b * 256 + tid, gsz = (size_t)nb * 256;
  for (size_t s = (size_t)used + gtid; s < (size_t)cap; s += gsz) idx[s] = P;
  for (size_t e = (size_t)used * width + gtid; e < (size_t)cap * width; e += gsz) packed[e] = 0.0f;
  for (size_t e = gtid; e < 2 * (size_t)P; e += gsz) tail[e] = densify[e];
  if (b == 0 && tid == 0) {
    status[0] = total;
    status[1] = total > cap ? 1 : 0;
  }
}

// after collective 2: the summed rows go back to their rows, the summed statistics to `densify`
__global__ __launch_bounds__(256) void exchange_unpack_kernel(int P, int width, int cap, const int32_t* __restrict__ idx,
                                                              const float* __restrict__ packed, const float* __restrict__ tail,
                                                              float* __restrict__ flat, float* __restrict__ densify) {
  // one element per thread and step: the packed rows are read as one contiguous stream (round 4 walked them a wave per
  // row, eight dependent idx -> row -> store trips per wave)
  const size_t gtid = (size_t)blockIdx.x * 256 + threadIdx.x, gsz = (size_t)gridDim.x * 256;
  const size_t total = (size_t)cap * width;
  for (size_t e = gtid; e < total; e += gsz) {
    const size_t s = e / (size_t)width;
    const int g = idx[s];
    if ((unsigned)g < (unsigned)P) flat[(size_t)g * width + (e - s * width)] = packed[e];
  }
  for (size_t e = gtid; e < 2 * (size_t)P; e += gsz) densify[e] = tail[e];
}

void launch_exchange_mask(int P, int width, const float* flat, const unsigned long long* row_mask, const int32_t* max_radii,
                          int32_t* imax, hipStream_t st) {
  if (P <= 0) return;
  exchange_mask_kernel<<<(unsigned)((P + 255) / 256), 256, 0, st>>>(P, width, flat, row_mask, max_radii, imax);
}

// a count without a pack (idx == NULL): status = {rows in the union, whether they exceed cap}
__global__ __launch_bounds__(256) void exchange_total_kernel(int nb, int cap, const int32_t* __restrict__ counts,
                                                             int32_t* __restrict__ status) {
  __shared__ int s_t[4];
  int total = 0;
  for (int i = threadIdx.x; i < nb; i += 256) total += counts[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
  if ((threadIdx.x & 63) == 0) s_t[threadIdx.x >> 6] = total;
  __syncthreads();
  if (threadIdx.x == 0) {
    total = s_t[0] + s_t[1] + s_t[2] + s_t[3];
    status[0] = total;
    status[1] = total > cap ? 1 : 0;
  }
}

void launch_exchange_pack(int P, int width, int cap, const float* flat, const int32_t* imax, int32_t* max_radii,
                          unsigned long long* row_mask, const float* densify, int32_t* idx, float* fsum, int32_t* counts,
                          int32_t* status, hipStream_t st) {
  if (P <= 0) return;
  const int nb = (P + EX_ROWS - 1) / EX_ROWS;
  exchange_count_kernel<<<(unsigned)nb, 256, 0, st>>>(P, imax, max_radii, row_mask, counts);
  if (idx == nullptr) {
    exchange_total_kernel<<<1, 256, 0, st>>>(nb, cap, counts, status);
    return;
  }
  exchange_pack_kernel<<<(unsigned)nb, 256, 0, st>>>(P, width, cap, nb, flat, imax, counts, densify, idx, fsum,
                                                     fsum + (size_t)cap * width, status);
}

void launch_exchange_unpack(int P, int width, int cap, const int32_t* idx, const float* fsum, float* flat, float* densify,
                            hipStream_t st) {
  if (P <= 0) return;
  const size_t work = std::max((size_t)cap * width, (size_t)2 * P);
  const unsigned nb = (unsigned)std::min<size_t>((work + 255) / 256, 4096);
  exchange_unpack_kernel<<<nb, 256, 0, st>>>(P, width, cap, idx, fsum, fsum + (size_t)cap * width, flat, densify);
}

}  // namespace olsr
