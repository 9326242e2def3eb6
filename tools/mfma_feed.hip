// Micro-benchmark: how much of the fp16 16x16x32 MFMA rate survives when the B fragments come from LDS the way k_conv_s3x<.,.,2> reads them
// (two ds_read_b64 per term and column block, every read touching each bank once), as a function of how many MFMAs share a fragment pair:
//   mode 0: no LDS reads (operands in registers): the power / clock ceiling of this instruction on random data
//   mode 1: wave tile 32 channels x 128 positions  (2 row blocks): 4 reads per 6 MFMAs   -- k_conv_s3x<3, 8, 2> today
//   mode 2: wave tile 64 channels x 64 positions   (4 row blocks): 4 reads per 12 MFMAs  -- the k_conv_c8x tile with two terms
//   mode 3: mode 1 + the A fragments of every k-step fetched from global memory (4 x 1 KiB per wave, L2-resident)
//   mode 4: mode 2 + its A fragments (8 x 1 KiB per wave and k-step)
//   mode 5: mode 1 + the A fragments of every k-step read from LDS (4 x ds_read_b128 per wave: the weights-through-LDS design of conv_h.hip)
//   mode 6: ONE-term (the 16-bit path of configs[3], k_conv_c8x's tile): 64 channels x 128 positions per wave, 2 reads per 4 MFMAs, 4 waves x 2
//           workgroups per CU are modelled as 8 waves; A fragments (4 x 1 KiB per k-step and wave) from global memory
//   mode 7: mode 6 with the A fragments in registers (no global loads): what the LDS reads alone cost the one-term kernel
// Round 6 (kernel k2): the corners the round-5 table left open, and the price of STAGING (LDS-DMA pieces + one workgroup barrier per two k-steps):
//   mode 8:  wave tile 64 ch x 64 pos, B AND A from LDS (A: 8 x ds_read_b128 per k-step and wave, the 8 KiB of a k-step shared by the eight waves)
//   mode 9:  wave tile 64 ch x 128 pos (128 accumulator registers), B and A from LDS: 24 KiB of LDS reads per 96 MFMAs
//   mode 10: mode 3 + staging of the B bricks (21 pieces per two k-steps, issued by waves 0-3 behind a vmcnt(0) + s_barrier): today's kernel
//   mode 11: mode 8 + staging of the B bricks AND of the A fragments (21 + 16 pieces per two k-steps)
//   mode 12: mode 9 + staging (a 1024-position tile: 40 + 16 pieces per two k-steps)
//   mode 13: mode 5 (32 ch x 128 pos, A from LDS) + staging (21 + 16 pieces)
//   modes 14-18 (kernel k3): the same tiles out of v_mfma_f32_32x32x16_f16 (three times the issue room per FLOP)
// 256 workgroups of 8 waves (one per CU, two waves per SIMD), 24 KiB of random fp16 in LDS per workgroup (k2: 160 KiB).
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_feed.hip -o tools/mfma_feed.bin   Run: tools/mfma_feed.bin  (tools/prof_r06_feed.sh: plain + SQ / LDS counters)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
typedef const volatile __attribute__((address_space(3))) unsigned long long* lds64_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE>
__global__ void __launch_bounds__(512, 1) k(const uint4* __restrict__ in, float* __restrict__ out, int ksteps) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  constexpr bool ONE = MODE >= 6;
  constexpr int RB = (MODE == 2 || MODE == 4 || ONE) ? 4 : 2, NCB = ONE ? 8 : RB == 4 ? 4 : 8;
  constexpr bool AG = MODE == 3 || MODE == 4 || MODE == 6, AL = MODE == 5, BL = MODE != 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m16 = lane & 15, g = lane >> 4;
  for (int i = tid; i < 24576 / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = in[(blockIdx.x * 1536 + i) & 65535];
  __syncthreads();
  u32x4 A[RB][2];
  for (int r = 0; r < RB; ++r) for (int t = 0; t < 2; ++t) A[r][t] = __builtin_bit_cast(u32x4, in[(tid + (r * 2 + t) * 4096) & 65535]);
  f32x4 acc[RB][NCB];
  for (int r = 0; r < RB; ++r) for (int c = 0; c < NCB; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned base = (unsigned)(unsigned long long)(lptr_t)lds;
  const unsigned lane_b = (unsigned)((((wave >> 1) * NCB * 16 + m16) * 16) + (g & 1) * 8);
  u32x4 B[2][2];
  auto read_b = [&](u32x4 (&Bd)[2], unsigned vo, int cb) {
    for (int t = 0; t < (ONE ? 1 : 2); ++t) {
      u64x2 v;
      const unsigned lo = base + vo + t * 12288 + cb * 256;
      v.x = *(lds64_t)(lo); v.y = *(lds64_t)(lo ^ 8u);
      Bd[t] = __builtin_bit_cast(u32x4, v);
    }
  };
  int tap = g;
  for (int s = 0; s < ksteps; ++s) {
    if (AG) {
      for (int r = 0; r < RB; ++r) for (int t = 0; t < 2; ++t)
        A[r][t] = __builtin_bit_cast(u32x4, in[((s & 63) * 64 + (r * 2 + t) * 4096 + lane + (wave & 1) * 8192) & 65535]);
    }
    if (AL) {
      typedef const volatile __attribute__((address_space(3))) u32x4* lds128_t;
      for (int r = 0; r < RB; ++r) for (int t = 0; t < 2; ++t)
        A[r][t] = *(lds128_t)(base + (unsigned)((((s & 3) * 4 + r * 2 + t) * 64 + lane) * 16));
    }
    const unsigned vo = lane_b + (unsigned)(((tap % 3) * 110 + tap / 3) * 16);
    tap = (tap + 4) % 9;
    if (BL) read_b(B[0], vo, 0);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      u32x4(&Bc)[2] = B[cb & 1];
      u32x4(&Bn)[2] = B[(cb + 1) & 1];
      if (BL && cb + 1 < NCB) read_b(Bn, vo, cb + 1);
      if (!BL) { Bc[0] = A[0][0]; Bc[1] = A[0][1]; }
      constexpr int TA[3] = {ONE ? 0 : 1, 0, 0}, TB[3] = {0, 1, 0};
#pragma unroll
      for (int m = 0; m < (ONE ? 1 : 3); ++m)
#pragma unroll
        for (int r = 0; r < RB; ++r)
          acc[r][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[r][TA[m]]), __builtin_bit_cast(f16x8, Bc[TB[m]]), acc[r][cb], 0, 0, 0);
    }
  }
  float sum = 0.f;
  for (int r = 0; r < RB; ++r) for (int c = 0; c < NCB; ++c) for (int e = 0; e < 4; ++e) sum += acc[r][c][e];
  out[blockIdx.x * 512 + tid] = sum;
}

template <int MODE>
void run(const char* name, const uint4* in, float* out) {
  const int ksteps = 4000, grid = 256;
  constexpr bool ONE = MODE >= 6;
  constexpr int RB = (MODE == 2 || MODE == 4 || ONE) ? 4 : 2, NCB = ONE ? 8 : RB == 4 ? 4 : 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 24576, 0, in, out, ksteps);
  hipEventRecord(e0);
  const int reps = 8;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 24576, 0, in, out, ksteps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double flop = 2.0 * 16 * 16 * 32 * (ONE ? 1 : 3) * RB * NCB * (double)ksteps * grid * 8;
  printf("{\"mode\": \"%s\", \"ms\": %.3f, \"mfma_tflops\": %.0f, \"fp32_equiv_tflops\": %.0f}\n", name, ms, flop / (ms * 1e-3) / 1e12, flop / 3 / (ms * 1e-3) / 1e12);
}

// ---- round 6: wave tiles with BOTH operands from LDS, with and without the LDS-DMA staging + barrier that feeds them
// RB row blocks x NCB column blocks per wave; ASRC 0 = registers, 1 = global (per wave), 2 = LDS; STAGE = DMA pieces per two k-steps (0: none)
// LDS: [0, 80K) B image (term stride 40 KiB), [80K, 112K) A ring (4 k-steps x 8 KiB), [112K, 160K) landing area of the brick pieces
// WAVES = 4 (modes 19, 20): ONE wave per SIMD with the whole register file (512) -- the 64 x 128 wave tile WITH room for a second accumulator set
template <int RB, int NCB, int ASRC, int STAGE, int WAVES = 8>
__global__ void __launch_bounds__(WAVES * 64, 1) k2(const uint4* __restrict__ in, float* __restrict__ out, int ksteps) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  constexpr int NA = RB * 2;  // A fragments per k-step and wave
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), m16 = lane & 15, g = lane >> 4;
  for (int i = tid; i < 163840 / 16; i += WAVES * 64) reinterpret_cast<uint4*>(lds)[i] = in[(blockIdx.x * 1536 + i) & 65535];
  __syncthreads();
  const unsigned base = (unsigned)(unsigned long long)(lptr_t)lds;
  // position group: RB = 4 -> every wave its own NCB * 16 positions; RB = 2 -> waves 2p, 2p + 1 share a group (the two channel halves)
  const int pg = RB == 4 ? wave : wave >> 1;
  const unsigned lane_b = (unsigned)(((pg * NCB * 16 + m16) * 16) + (g & 1) * 8);
  constexpr unsigned kTerm = 40960, kARing = 81920, kLand = 114688;
  u32x4 A[2][NA];
  for (int i = 0; i < NA; ++i) { A[0][i] = __builtin_bit_cast(u32x4, in[(tid + i * 4096) & 65535]); A[1][i] = A[0][i]; }
  f32x4 acc[RB][NCB];
  for (int r = 0; r < RB; ++r) for (int c = 0; c < NCB; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 B[2][2];
  auto read_b = [&](u32x4 (&Bd)[2], unsigned vo, int cb) {
    for (int t = 0; t < 2; ++t) {
      u64x2 v;
      const unsigned lo = base + vo + t * kTerm + cb * 256;
      v.x = *(lds64_t)(lo); v.y = *(lds64_t)(lo ^ 8u);
      Bd[t] = __builtin_bit_cast(u32x4, v);
    }
  };
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(in), 0, 65536u * 16u, 0x00020000);
  int tap = g;
  auto load_a = [&](u32x4 (&Ad)[NA], int s) {
    if constexpr (ASRC == 1) {
      for (int i = 0; i < NA; ++i) Ad[i] = __builtin_bit_cast(u32x4, in[((s & 63) * 64 + i * 4096 + lane + (wave & 1) * 8192) & 65535]);
    } else if constexpr (ASRC == 2) {
      typedef const volatile __attribute__((address_space(3))) u32x4* lds128_t;
      // RB = 2: the two channel halves read different halves of the k-step's 8 KiB
      const unsigned ab = base + kARing + (unsigned)((s & 3) * 8192 + (RB == 2 ? (wave & 1) * 4096 : 0) + lane * 16);
      for (int i = 0; i < NA; ++i) Ad[i] = *(lds128_t)(ab + i * 1024);
    }
  };
  auto kstep = [&](u32x4 (&Ac)[NA], u32x4 (&An)[NA], int s) {
    load_a(An, s + 1);
    const unsigned vo = lane_b + (unsigned)(((tap % 3) * 110 + tap / 3) * 16);
    tap = (tap + 4) % 9;
    read_b(B[0], vo, 0);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      u32x4(&Bc)[2] = B[cb & 1];
      u32x4(&Bn)[2] = B[(cb + 1) & 1];
      if (cb + 1 < NCB) read_b(Bn, vo, cb + 1);
      constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < RB; ++r)
          acc[r][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, Ac[r * 2 + TA[m]]), __builtin_bit_cast(f16x8, Bc[TB[m]]), acc[r][cb], 0, 0, 0);
    }
  };
  int piece = 0;
  for (int s = 0; s < ksteps; s += 2) {
    if constexpr (STAGE > 0) {
      // own pieces of the pair before last have landed; everybody's after the barrier; then the next pair's pieces (waves 0-3: one per SIMD pair)
      if (wave < 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (wave < 4) {
#pragma unroll 1
        for (int pc = wave; pc < STAGE; pc += 4) {
          const unsigned dst = ASRC == 2 && pc >= STAGE - 16 ? kARing + (unsigned)((((s + 2) & 3) * 8 + (pc - (STAGE - 16))) * 1024)
                                                             : kLand + (unsigned)(((piece + pc) % 48) * 1024);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(lds + dst), 16, (unsigned)(((piece * 64 + pc * 64 + lane) & 65535) * 16), 0, 0, 0);
        }
        piece += STAGE;
      }
    }
    kstep(A[0], A[1], s);
    kstep(A[1], A[0], s + 1);
  }
  float sum = 0.f;
  for (int r = 0; r < RB; ++r) for (int c = 0; c < NCB; ++c) for (int e = 0; e < 4; ++e) sum += acc[r][c][e];
  out[blockIdx.x * 512 + tid] = sum;
}

template <int RB, int NCB, int ASRC, int STAGE, int WAVES = 8>
void run2(const char* name, const uint4* in, float* out) {
  const int ksteps = 4000, grid = 256;
  auto kern = k2<RB, NCB, ASRC, STAGE, WAVES>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), 163840, 0, in, out, ksteps);
  hipEventRecord(e0);
  const int reps = 8;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), 163840, 0, in, out, ksteps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double flop = 2.0 * 16 * 16 * 32 * 3 * RB * NCB * (double)ksteps * grid * WAVES;
  printf("{\"mode\": \"%s\", \"ms\": %.3f, \"mfma_tflops\": %.0f, \"fp32_equiv_tflops\": %.0f}\n", name, ms, flop / (ms * 1e-3) / 1e12, flop / 3 / (ms * 1e-3) / 1e12);
}

// ---- round 6, second question: is the pipe ISSUE-bound?  (the SQ counters of modes 0-13: 1.9-2.08 GHz in every mode, MFMA pipe busy 0.93 -> 0.65 as LDS
// read instructions per MFMA go 0 -> 0.75: the clock is not what the reads cost.)  v_mfma_f32_32x32x16_f16 blocks the SIMD's vector issue for 8 of its 32
// cycles where 16x16x32 blocks 8 of 16: three times the issue room per FLOP for the same fragment reads.
// k3: wave tile (32 RB) channels x (32 NCB) positions out of 32x32x16 MFMAs, k-step = 16 (two lane groups = two taps); B from LDS as two ds_read_b64 per
// term and column block (32 lanes x 8 B: every read touches each bank once), A from global (ASRC 1) or LDS (2); STAGE as in k2 per 64 K-values.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int RB, int NCB, int ASRC, int STAGE>
__global__ void __launch_bounds__(512, 1) k3(const uint4* __restrict__ in, float* __restrict__ out, int ksteps) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  constexpr int NA = RB * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n32 = lane & 31, g = lane >> 5;
  for (int i = tid; i < 163840 / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = in[(blockIdx.x * 1536 + i) & 65535];
  __syncthreads();
  const unsigned base = (unsigned)(unsigned long long)(lptr_t)lds;
  const int pg = RB == 2 ? wave : wave >> 1;
  const unsigned lane_b = (unsigned)(((pg * NCB * 32 + n32) * 16) + (g & 1) * 8);
  constexpr unsigned kTerm = 40960, kARing = 81920, kLand = 114688;
  u32x4 A[2][NA];
  for (int i = 0; i < NA; ++i) { A[0][i] = __builtin_bit_cast(u32x4, in[(tid + i * 4096) & 65535]); A[1][i] = A[0][i]; }
  f32x16 acc[RB][NCB];
  for (int r = 0; r < RB; ++r) for (int c = 0; c < NCB; ++c) for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;
  u32x4 B[2][2];
  auto read_b = [&](u32x4 (&Bd)[2], unsigned vo, int cb) {
    for (int t = 0; t < 2; ++t) {
      u64x2 v;
      const unsigned lo = base + vo + t * kTerm + cb * 512;
      v.x = *(lds64_t)(lo); v.y = *(lds64_t)(lo ^ 8u);
      Bd[t] = __builtin_bit_cast(u32x4, v);
    }
  };
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(in), 0, 65536u * 16u, 0x00020000);
  int tap = g;
  auto load_a = [&](u32x4 (&Ad)[NA], int s) {
    if constexpr (ASRC == 1) {
      for (int i = 0; i < NA; ++i) Ad[i] = __builtin_bit_cast(u32x4, in[((s & 63) * 64 + i * 4096 + lane + (wave & 1) * 8192) & 65535]);
    } else if constexpr (ASRC == 2) {
      typedef const volatile __attribute__((address_space(3))) u32x4* lds128_t;
      const unsigned ab = base + kARing + (unsigned)((s & 7) * 4096 + (RB == 1 ? (wave & 1) * 2048 : 0) + lane * 16);
      for (int i = 0; i < NA; ++i) Ad[i] = *(lds128_t)(ab + i * 1024);
    }
  };
  auto kstep = [&](u32x4 (&Ac)[NA], u32x4 (&An)[NA], int s) {
    load_a(An, s + 1);
    const unsigned vo = lane_b + (unsigned)(((tap % 3) * 110 + tap / 3) * 16);
    tap = (tap + 2) % 9;
    read_b(B[0], vo, 0);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      u32x4(&Bc)[2] = B[cb & 1];
      u32x4(&Bn)[2] = B[(cb + 1) & 1];
      if (cb + 1 < NCB) read_b(Bn, vo, cb + 1);
      constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < RB; ++r)
          acc[r][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Ac[r * 2 + TA[m]]), __builtin_bit_cast(f16x8, Bc[TB[m]]), acc[r][cb], 0, 0, 0);
    }
  };
  int piece = 0;
  for (int s = 0; s < ksteps; s += 4) {  // four 16-deep steps = the K of two 32-deep ones
    if constexpr (STAGE > 0) {
      if (wave < 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (wave < 4) {
#pragma unroll 1
        for (int pc = wave; pc < STAGE; pc += 4) {
          const unsigned dst = ASRC == 2 && pc >= STAGE - 16 ? kARing + (unsigned)(((((s >> 1) + 2) & 3) * 8 + (pc - (STAGE - 16))) * 1024)
                                                             : kLand + (unsigned)(((piece + pc) % 48) * 1024);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(lds + dst), 16, (unsigned)(((piece * 64 + pc * 64 + lane) & 65535) * 16), 0, 0, 0);
        }
        piece += STAGE;
      }
    }
    kstep(A[0], A[1], s);
    kstep(A[1], A[0], s + 1);
    kstep(A[0], A[1], s + 2);
    kstep(A[1], A[0], s + 3);
  }
  float sum = 0.f;
  for (int r = 0; r < RB; ++r) for (int c = 0; c < NCB; ++c) for (int e = 0; e < 16; ++e) sum += acc[r][c][e];
  out[blockIdx.x * 512 + tid] = sum;
}

template <int RB, int NCB, int ASRC, int STAGE>
void run3(const char* name, const uint4* in, float* out) {
  const int ksteps = 8000, grid = 256;
  auto kern = k3<RB, NCB, ASRC, STAGE>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 163840, 0, in, out, ksteps);
  hipEventRecord(e0);
  const int reps = 8;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 163840, 0, in, out, ksteps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double flop = 2.0 * 32 * 32 * 16 * 3 * RB * NCB * (double)ksteps * grid * 8;
  printf("{\"mode\": \"%s\", \"ms\": %.3f, \"mfma_tflops\": %.0f, \"fp32_equiv_tflops\": %.0f}\n", name, ms, flop / (ms * 1e-3) / 1e12, flop / 3 / (ms * 1e-3) / 1e12);
}

int main() {
  std::vector<unsigned short> h(65536 * 8 + 64);
  srand(1);
  for (auto& v : h) { const _Float16 f = (_Float16)((rand() % 2001 - 1000) / 500.0f); __builtin_memcpy(&v, &f, 2); }
  uint4* in; float* out;
  hipMalloc(&in, h.size() * 2); hipMalloc(&out, 256 * 512 * 4);
  hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("0 registers only", in, out);
    run<1>("1 32ch x 128pos, B from LDS (4 reads / 6 MFMA)", in, out);
    run<2>("2 64ch x 64pos, B from LDS (4 reads / 12 MFMA)", in, out);
    run<3>("3 mode 1 + A from global (4 KiB / k-step / wave)", in, out);
    run<4>("4 mode 2 + A from global (8 KiB / k-step / wave)", in, out);
    run<5>("5 mode 1 + A from LDS (4 KiB / k-step / wave)", in, out);
    run<6>("6 ONE term, 64ch x 128pos, B from LDS (2 reads / 4 MFMA), A from global", in, out);
    run<7>("7 ONE term, 64ch x 128pos, B from LDS, A in registers", in, out);
    run2<4, 4, 2, 0>("8 64ch x 64pos, B and A from LDS", in, out);
    run2<4, 8, 2, 0>("9 64ch x 128pos, B and A from LDS", in, out);
    run2<2, 8, 1, 21>("10 mode 3 + staging of the bricks (21 pieces + barrier / 2 k-steps): today's kernel", in, out);
    run2<4, 4, 2, 37>("11 mode 8 + staging of bricks and A (37 pieces + barrier / 2 k-steps)", in, out);
    run2<4, 8, 2, 56>("12 mode 9 + staging of bricks and A (56 pieces + barrier / 2 k-steps)", in, out);
    run2<2, 8, 2, 37>("13 mode 5 (32ch x 128pos, A from LDS) + staging (37 pieces + barrier / 2 k-steps)", in, out);
    run3<1, 4, 0, 0>("14 32x32x16: 32ch x 128pos, B from LDS, A in registers (mode 1's tile)", in, out);
    run3<1, 4, 1, 21>("15 32x32x16: 32ch x 128pos, B from LDS, A from global, staging (mode 10's structure)", in, out);
    run3<2, 2, 2, 37>("16 32x32x16: 64ch x 64pos, A and B from LDS, staging (mode 11's structure)", in, out);
    run3<2, 4, 2, 56>("17 32x32x16: 64ch x 128pos, A and B from LDS, staging (mode 12's structure)", in, out);
    run3<2, 4, 0, 0>("18 32x32x16: 64ch x 128pos, B from LDS, A in registers", in, out);
    run2<4, 8, 2, 0, 4>("19 FOUR waves (one per SIMD) x 64ch x 128pos, B and A from LDS", in, out);
    run2<4, 8, 2, 37, 4>("20 mode 19 + staging of bricks and A (37 pieces + barrier / 2 k-steps): a 512-position tile", in, out);
  }
  return 0;
}
