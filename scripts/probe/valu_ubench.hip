// valu_ubench.hip — what does a wave64 VALU instruction cost on gfx950?
//
// Settles the issue-peak constant of bench.py's VALU roofline: MI355X_MICROARCH.md gives
// `v_fma_f32 (wave64): 2 cyc (SIMD-32)`, bench.py assumed one instruction per quad-cycle (4 clocks).  The probe runs a
// stream of N x UNROLL instructions of one kind per wave, as CHAINS independent dependency chains (1 = every instruction
// waits for the previous one, 8 = eight interleaved accumulators), with W waves resident per SIMD (W blocks of 256
// threads per CU; a 256-thread block puts one wave on each of the CU's four SIMDs), and reports
//     cycles per instruction per SIMD = W-wave elapsed shader clocks (s_memtime) / (instructions per wave x W)
// i.e. the reciprocal issue throughput of ONE SIMD with W waves competing for it, plus the same figure from the
// wall clock (hipEvents) at the nominal 2.4 GHz.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_ubench.out valu_ubench.hip && ./valu_ubench.out > r3_valu_ubench.json
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(e)                                                                     \
  do {                                                                               \
    hipError_t _e = (e);                                                             \
    if (_e != hipSuccess) {                                                          \
      std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e));                   \
      std::exit(1);                                                                  \
    }                                                                                \
  } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

enum Op {
  OP_FMA, OP_PK_FMA, OP_MUL, OP_ADD, OP_PK_MUL, OP_EXP, OP_RCP, OP_SQRT, OP_CNDMASK, OP_CMP, OP_MAX, OP_DPP_ADD,
  OP_PERMLANE32_SWAP, OP_READLANE, OP_LDS_B128, OP_CVT_I32, OP_RNDNE, OP_MOV, OP_LSHL_ADD, OP_MFMA_16x16x4_F32, OP_CNDMASK_SGPR, OP_CNDMASK_VCC_BLOCK, OP_CMP_VCC_CNDMASK, OP_ADD_U32, OP_AND_B32, OP_FMAC, OP_MIN, OP_PERMLANE16_SWAP, OP_DPP_MOV, OP_BPERMUTE, OP_SALU_AND, OP_FMA_MIX_EXP, OP_CNDMASK_E64_VCC, OP_CNDMASK_E32_SRCS, OP_CNDMASK_SGPR_BLOCK, OP_COUNT
};
static const char* op_name[OP_COUNT] = {
    "v_fma_f32", "v_pk_fma_f32", "v_mul_f32", "v_add_f32", "v_pk_mul_f32", "v_exp_f32", "v_rcp_f32", "v_sqrt_f32",
    "v_cndmask_b32", "v_cmp_lt_f32 (-> sgpr pair)", "v_max_f32", "v_add_f32 dpp row_ror:4",
    "v_permlane32_swap", "v_readlane_b32", "ds_read_b128 (broadcast)", "v_cvt_i32_f32", "v_rndne_f32", "v_mov_b32",
    "v_lshl_add_u32", "v_mfma_f32_16x16x4_f32", "v_cndmask_b32_e64 (sgpr-pair mask)", "v_cndmask_b32_e32 vcc (8 per asm block)", "v_cmp_lt_f32 vcc + v_cndmask_b32 vcc (pair = 1 inst)", "v_add_u32", "v_and_b32", "v_fmac_f32", "v_min_f32", "v_permlane16_swap", "v_mov_b32 dpp row_ror:4", "ds_bpermute_b32", "s_and_b64 (SALU)", "mix: 3 v_fma + 1 v_exp (per inst)", "v_cndmask_b32_e64 vcc as explicit mask (8 per asm block)", "v_cndmask_b32_e32 vcc, distinct src1 regs (8 per asm block)", "v_cndmask_b32_e64 sgpr-pair mask (8 per asm block)"};

constexpr int UNROLL = 64;  // instructions per loop iteration (a multiple of every CHAINS)

template <int OP, int CHAINS>
__device__ __forceinline__ void body(float (&a)[8], v2f (&p)[8], float b, float c, const float4* lds,
                                     unsigned long long& sink) {
  typedef float v4f __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int i = 0; i < UNROLL; ++i) {
    const int k = i % CHAINS;
    if constexpr (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
    else if constexpr (OP == OP_PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(v2f{b, b}), "v"(v2f{c, c}));
    else if constexpr (OP == OP_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
    else if constexpr (OP == OP_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
    else if constexpr (OP == OP_PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(v2f{b, b}));
    else if constexpr (OP == OP_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k]));
    else if constexpr (OP == OP_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
    else if constexpr (OP == OP_SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
    else if constexpr (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : );
    else if constexpr (OP == OP_CMP) {
      unsigned long long m;
      asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m) : "v"(a[k]), "v"(b));
      sink ^= m;  // (SALU: issues beside the VALU stream)
    } else if constexpr (OP == OP_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
    else if constexpr (OP == OP_DPP_ADD) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
    else if constexpr (OP == OP_PERMLANE32_SWAP) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[k]), "+v"(a[(k + 4) % 8]));
    else if constexpr (OP == OP_READLANE) {
      unsigned s;
      asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(a[k]));
      sink += s;
    } else if constexpr (OP == OP_LDS_B128) {
      v4f q;
      asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"((unsigned)(16 * (i & 63))));
      // (no wait per read: throughput of the LDS pipe for wave-uniform 16-byte reads; the last one is waited for below)
      if (i == UNROLL - 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        a[0] += q.x;
      }
    } else if constexpr (OP == OP_CVT_I32) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[k]));
    else if constexpr (OP == OP_RNDNE) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[k]));
    else if constexpr (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a[k]) : "v"(b));
    else if constexpr (OP == OP_LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[k]) : "v"(b));
    else if constexpr (OP == OP_CNDMASK_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "s"(0x5555555555555555ull ^ sink));
    else if constexpr (OP == OP_CNDMASK_VCC_BLOCK) {
      if (i % 8 == 0)
        asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n"
                     "v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc"
                     : "+v"(a[0]), "+v"(a[1 % CHAINS]), "+v"(a[2 % CHAINS]), "+v"(a[3 % CHAINS]), "+v"(a[4 % CHAINS]), "+v"(a[5 % CHAINS]), "+v"(a[6 % CHAINS]), "+v"(a[7 % CHAINS]) : "v"(b) : "vcc");
    } else if constexpr (OP == OP_CMP_VCC_CNDMASK) {
      if (i % 2 == 0) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %0\n v_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(a[k]) : "v"(b), "v"(c) : "vcc");
    } else if constexpr (OP == OP_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
    else if constexpr (OP == OP_AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
    else if constexpr (OP == OP_FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
    else if constexpr (OP == OP_MIN) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
    else if constexpr (OP == OP_PERMLANE16_SWAP) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[k]), "+v"(a[(k + 4) % 8]));
    else if constexpr (OP == OP_DPP_MOV) asm volatile("v_mov_b32_dpp %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
    else if constexpr (OP == OP_BPERMUTE) {
      asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a[k]) : "v"((unsigned)(4 * ((threadIdx.x + 1) & 63))));
    } else if constexpr (OP == OP_SALU_AND) {
      asm volatile("s_and_b64 %0, %0, %1" : "+s"(sink) : "s"(0x7777777777777777ull));
    } else if constexpr (OP == OP_FMA_MIX_EXP) {
      if (i % 4 == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k]));
      else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
    }
    else if constexpr (OP == OP_CNDMASK_E64_VCC) {
      if (i % 8 == 0)
        asm volatile("v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n"
                     "v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc"
                     : "+v"(a[0]), "+v"(a[1 % CHAINS]), "+v"(a[2 % CHAINS]), "+v"(a[3 % CHAINS]), "+v"(a[4 % CHAINS]), "+v"(a[5 % CHAINS]), "+v"(a[6 % CHAINS]), "+v"(a[7 % CHAINS]) : "v"(b) : "vcc");
    } else if constexpr (OP == OP_CNDMASK_E32_SRCS) {
      if (i % 8 == 0)
        asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %9, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %9, vcc\n"
                     "v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %9, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %9, vcc"
                     : "+v"(a[0]), "+v"(a[1 % CHAINS]), "+v"(a[2 % CHAINS]), "+v"(a[3 % CHAINS]), "+v"(a[4 % CHAINS]), "+v"(a[5 % CHAINS]), "+v"(a[6 % CHAINS]), "+v"(a[7 % CHAINS]) : "v"(b), "v"(c) : "vcc");
    } else if constexpr (OP == OP_CNDMASK_SGPR_BLOCK) {
      if (i % 8 == 0)
        asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n"
                     "v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9"
                     : "+v"(a[0]), "+v"(a[1 % CHAINS]), "+v"(a[2 % CHAINS]), "+v"(a[3 % CHAINS]), "+v"(a[4 % CHAINS]), "+v"(a[5 % CHAINS]), "+v"(a[6 % CHAINS]), "+v"(a[7 % CHAINS]) : "v"(b), "s"(0x5555555555555555ull));
    }
    else if constexpr (OP == OP_MFMA_16x16x4_F32) {
      // D[16x16] += A[16x4] B[4x16]: 4 accumulator registers per lane; chains = independent accumulators
      v4f* acc = reinterpret_cast<v4f*>(&p[0]);  // p[0..7] = 16 floats = 4 accumulators
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[k % 4]) : "v"(b), "v"(c));
    }
  }
  (void)lds;
}

template <int OP, int CHAINS>
__global__ __launch_bounds__(256) void probe_kernel(int iters, float b, float c, unsigned long long* stamps,
                                                    float* out) {
  __shared__ float4 s_lds[64];
  if (threadIdx.x < 64) s_lds[threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f);
  __syncthreads();
  float a[8];
  v2f p[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a[k] = 1.0f + 1e-3f * (float)((threadIdx.x + k) & 7);
    p[k] = v2f{a[k], a[k]};
  }
  unsigned long long sink = 0;
  // vcc for v_cndmask: half the lanes
  asm volatile("s_mov_b32 vcc_lo, 0x55555555\n s_mov_b32 vcc_hi, 0x55555555" ::: "vcc");
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) body<OP, CHAINS>(a, p, b, c, s_lds, sink);
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = (float)sink;
#pragma unroll
  for (int k = 0; k < 8; ++k) r += a[k] + p[k].x + p[k].y;
  if ((threadIdx.x & 63) == 0) {
    const size_t wv = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * wv] = t0;
    stamps[2 * wv + 1] = t1;
  }
  if (r == 123.456f) out[0] = r;  // keep the results alive
}

struct Result {
  double cyc_per_inst_simd;   // from s_memtime: mean wave elapsed / (insts per wave * W)
  double cyc_per_inst_wall;   // from hipEvents at 2.4 GHz
  double wave_elapsed_mean;
};

template <int OP, int CHAINS>
Result run(int waves_per_simd, int ncu, int iters, unsigned long long* d_stamps, float* d_out) {
  const int nblk = ncu * waves_per_simd;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  probe_kernel<OP, CHAINS><<<nblk, 256>>>(iters / 8 + 1, 0.999f, 1e-3f, d_stamps, d_out);  // warm-up (code fetch, clocks)
  CHECK(hipEventRecord(e0));
  probe_kernel<OP, CHAINS><<<nblk, 256>>>(iters, 0.999f, 1e-3f, d_stamps, d_out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> st((size_t)nblk * 8);
  CHECK(hipMemcpy(st.data(), d_stamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double sum = 0.0;
  for (int w = 0; w < nblk * 4; ++w) sum += (double)(st[2 * w + 1] - st[2 * w]);
  const double mean = sum / (nblk * 4);
  const double insts = (double)iters * UNROLL;
  Result r;
  r.wave_elapsed_mean = mean;
  r.cyc_per_inst_simd = mean / (insts * waves_per_simd);
  r.cyc_per_inst_wall = (ms * 1e-3 * 2.4e9) / (insts * waves_per_simd);
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return r;
}

template <int OP>
void run_op(int ncu, unsigned long long* d_stamps, float* d_out, bool& first) {
  const int iters = (OP == OP_EXP || OP == OP_RCP || OP == OP_SQRT || OP == OP_MFMA_16x16x4_F32) ? 400 : 1000;
  static const int W[4] = {1, 2, 4, 8};
  for (int ci = 0; ci < 2; ++ci) {
    for (int wi = 0; wi < 4; ++wi) {
      const Result r = ci == 0 ? run<OP, 1>(W[wi], ncu, iters, d_stamps, d_out) : run<OP, 8>(W[wi], ncu, iters, d_stamps, d_out);
      std::printf("%s\n    {\"op\": \"%s\", \"chains\": %d, \"waves_per_simd\": %d, \"cycles_per_inst_per_simd\": %.3f, "
                  "\"wave_memtime_ticks_per_own_inst\": %.3f, \"insts_per_wave\": %d}",
                  first ? "" : ",", op_name[OP], ci == 0 ? 1 : 8, W[wi], r.cyc_per_inst_wall,
                  r.wave_elapsed_mean / ((double)iters * UNROLL), iters * UNROLL);
      std::fflush(stdout);
      first = false;
    }
  }
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  unsigned long long* d_stamps;
  float* d_out;
  CHECK(hipMalloc(&d_stamps, sizeof(unsigned long long) * 2 * 4 * (size_t)ncu * 8));
  CHECK(hipMalloc(&d_out, 64));
  std::printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"compute_units\": %d, \"clock_khz\": %d,\n \"note\": \"cycles_per_inst_per_simd = kernel "
              "duration (hipEvents) x 2.4 GHz / (instructions per wave x waves per SIMD): the reciprocal issue throughput of one SIMD with that many "
              "waves on it (the f32 MFMA row reproduces the guide's 32 cycles per SIMD, so the nominal clock holds in these runs); "
              "wave_memtime_ticks_per_own_inst = s_memtime ticks a wave spends per instruction of its own; chains = independent dependency "
              "chains per wave; blocks of 256 threads (one wave per SIMD), waves_per_simd blocks per CU\",\n \"rows\": [",
              prop.name, prop.gcnArchName, ncu, prop.clockRate);
  bool first = true;
  run_op<OP_FMA>(ncu, d_stamps, d_out, first);
  run_op<OP_PK_FMA>(ncu, d_stamps, d_out, first);
  run_op<OP_MUL>(ncu, d_stamps, d_out, first);
  run_op<OP_ADD>(ncu, d_stamps, d_out, first);
  run_op<OP_PK_MUL>(ncu, d_stamps, d_out, first);
  run_op<OP_MAX>(ncu, d_stamps, d_out, first);
  run_op<OP_MOV>(ncu, d_stamps, d_out, first);
  run_op<OP_CNDMASK>(ncu, d_stamps, d_out, first);
  run_op<OP_CMP>(ncu, d_stamps, d_out, first);
  run_op<OP_LSHL_ADD>(ncu, d_stamps, d_out, first);
  run_op<OP_CVT_I32>(ncu, d_stamps, d_out, first);
  run_op<OP_RNDNE>(ncu, d_stamps, d_out, first);
  run_op<OP_EXP>(ncu, d_stamps, d_out, first);
  run_op<OP_RCP>(ncu, d_stamps, d_out, first);
  run_op<OP_SQRT>(ncu, d_stamps, d_out, first);
  run_op<OP_DPP_ADD>(ncu, d_stamps, d_out, first);
  run_op<OP_PERMLANE32_SWAP>(ncu, d_stamps, d_out, first);
  run_op<OP_READLANE>(ncu, d_stamps, d_out, first);
  run_op<OP_LDS_B128>(ncu, d_stamps, d_out, first);
  run_op<OP_MFMA_16x16x4_F32>(ncu, d_stamps, d_out, first);
  run_op<OP_CNDMASK_SGPR>(ncu, d_stamps, d_out, first);
  run_op<OP_CNDMASK_VCC_BLOCK>(ncu, d_stamps, d_out, first);
  run_op<OP_CMP_VCC_CNDMASK>(ncu, d_stamps, d_out, first);
  run_op<OP_ADD_U32>(ncu, d_stamps, d_out, first);
  run_op<OP_AND_B32>(ncu, d_stamps, d_out, first);
  run_op<OP_FMAC>(ncu, d_stamps, d_out, first);
  run_op<OP_MIN>(ncu, d_stamps, d_out, first);
  run_op<OP_PERMLANE16_SWAP>(ncu, d_stamps, d_out, first);
  run_op<OP_DPP_MOV>(ncu, d_stamps, d_out, first);
  run_op<OP_FMA_MIX_EXP>(ncu, d_stamps, d_out, first);
  run_op<OP_CNDMASK_E64_VCC>(ncu, d_stamps, d_out, first);
  run_op<OP_CNDMASK_E32_SRCS>(ncu, d_stamps, d_out, first);
  run_op<OP_CNDMASK_SGPR_BLOCK>(ncu, d_stamps, d_out, first);
  std::printf("\n ]}\n");
  return 0;
}
