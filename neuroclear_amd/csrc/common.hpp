// Shared host-side helpers for libnc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/nc_hip.h"

namespace nc {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return NC_ERR_HIP;
  }
  return NC_OK;
}

inline long cdiv(long a, long b) { return (a + b - 1) / b; }

// Kernels with more than 64 KiB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised once per (kernel, DEVICE): the
// attribute belongs to the code object loaded on that device, so a process that drives several GPUs (the reference's nn.DataParallel
// shape) needs it on each.  Thread-safe (launches come from the calling thread and from autograd's); defined in api.hip.
int raise_dyn_lds(const void* kernel, int bytes, const char* who);
template <typename K>
inline int raise_dyn_lds(K kernel, int bytes, const char* who) { return raise_dyn_lds(reinterpret_cast<const void*>(kernel), bytes, who); }

// LDS-DMA (one 16-byte / 4-byte unit per lane from a per-lane global address into LDS at lds_addr + lane * size)
// as inline assembly.  With __builtin_amdgcn_global_load_lds the compiler books the request as a FLAT access that may touch LDS:
// from then on every LDS-read wait in the kernel is lgkmcnt(0) (flat operations may return out of order), i.e. a software-pipelined
// fragment loop waits for the reads it has JUST issued, and any LDS read behind an outstanding request is preceded by vmcnt(0).
// The caller waits by hand (s_waitcnt vmcnt + barrier) where the staged data are needed.
#if defined(__HIP_DEVICE_COMPILE__)
// lds_addr: LDS byte address (nc_lds_addr(pointer into the kernel's LDS array)), wave-uniform
__device__ __forceinline__ unsigned nc_lds_addr(const void* lds_ptr) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)lds_ptr;
}
__device__ __forceinline__ void nc_dma_lds16(const void* src, unsigned lds_addr) {
  const unsigned m = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m), "v"(src) : "memory", "m0");
}
__device__ __forceinline__ void nc_dma_lds4(const void* src, unsigned lds_addr) {
  const unsigned m = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" : : "s"(m), "v"(src) : "memory", "m0");
}
#else  // (host pass: declarations only)
__device__ unsigned nc_lds_addr(const void* lds_ptr);
__device__ void nc_dma_lds16(const void* src, unsigned lds_addr);
__device__ void nc_dma_lds4(const void* src, unsigned lds_addr);
#endif


// 256 B of zeros in the code object of a translation unit: the source of padding / out-of-range lanes of LDS-DMA copies
// (no memset launch per convolution).  NC_ZERO_PAGE() once at namespace scope, nc_zero_page() = its device address.
#define NC_ZERO_PAGE()                                                                          \
  static __device__ __attribute__((used, aligned(256))) float g_nc_zero_page[64];               \
  static const float* nc_zero_page() {                                                          \
    static const float* p = nullptr;                                                            \
    if (!p) {                                                                                   \
      void* q = nullptr;                                                                        \
      if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_nc_zero_page)) == hipSuccess) p = (const float*)q; \
    }                                                                                           \
    return p;                                                                                   \
  }

// Conv geometry shared by the direct and MFMA paths.
struct ConvDims {
  int N, C, D, H, W, K, kd, kh, kw, sd, sh, sw, pd, ph, pw, Do, Ho, Wo;
};

inline bool make_dims(ConvDims& d, int N, int C, int D, int H, int W, int K, int kd, int kh, int kw, int stride,
                      int pad) {
  d.N = N; d.C = C; d.D = D; d.H = H; d.W = W; d.K = K; d.kd = kd; d.kh = kh; d.kw = kw;
  d.sd = kd > 1 ? stride : 1; d.sh = stride; d.sw = stride;
  d.pd = kd > 1 ? pad : 0; d.ph = pad; d.pw = pad;
  if (N < 1 || C < 1 || K < 1 || D < 1 || H < 1 || W < 1 || kd < 1 || kh < 1 || kw < 1 || stride < 1 || pad < 0)
    return false;
  d.Do = (D + 2 * d.pd - kd) / d.sd + 1;
  d.Ho = (H + 2 * d.ph - kh) / d.sh + 1;
  d.Wo = (W + 2 * d.pw - kw) / d.sw + 1;
  return d.Do >= 1 && d.Ho >= 1 && d.Wo >= 1 && (D + 2 * d.pd >= kd) && (H + 2 * d.ph >= kh) && (W + 2 * d.pw >= kw);
}

// ---- direct (VALU) kernels, conv_direct.hip
int conv_fwd_direct(const float* x, const float* w, const float* b, float* y, const ConvDims& d, hipStream_t s);
int conv_dgrad_direct(const float* dy, const float* w, float* dx, const ConvDims& d, hipStream_t s);
int conv_wgrad_direct(const float* x, const float* dy, float* dw, const ConvDims& d, hipStream_t s);
int bias_grad(const float* dy, float* db, int N, int K, long S, void* ws, size_t wsb, hipStream_t s);
static constexpr size_t kBiasGradWsBytes = 64 * 1024 * sizeof(double);  // K * splits <= 65536 partial sums

// ---- MFMA implicit-GEMM kernels, conv_mfma.hip
bool mfma_fwd_supported(const ConvDims& d);
// conv_c1k3.hip: 1 -> K channels, 3^3, stride 1, padding 1 forward (the U-Net's first layer)
bool c1k3_fwd_supported(const ConvDims& d);
int conv_fwd_c1k3(const float* x, const float* w, const float* bias, float* y, const ConvDims& d, hipStream_t s, float* stats_part = nullptr);
// (stats_part: the kernel also leaves partial InstanceNorm sums of its bias-free output, N == 1; c1k3_stats_finalize turns them into mean / rstd)
size_t c1k3_stats_bytes(const ConvDims& d);
int c1k3_stats_finalize(const float* stats_part, const float* bias, const ConvDims& d, float eps, float* mean, float* rstd, hipStream_t s);
bool mfma_dgrad_supported(const ConvDims& d);
bool mfma_wgrad_supported(const ConvDims& d);
size_t mfma_ws_bytes(const ConvDims& d);
size_t mfma_fwd_ws_bytes(const ConvDims& d);
int conv_fwd_mfma(const float* x, const float* w, const float* b, float* y, const ConvDims& d, void* ws, size_t wsb,
                  hipStream_t s);
int conv_dgrad_mfma(const float* dy, const float* w, float* dx, const ConvDims& d, void* ws, size_t wsb,
                    hipStream_t s);
int conv_wgrad_mfma(const float* x, const float* dy, float* dw, const ConvDims& d, void* ws, size_t wsb,
                    hipStream_t s);

// ---- generic gather-GEMM MFMA kernels, conv_gemm.hip
bool gemm_fwd_supported(const ConvDims& d);
bool gemm_dgrad_supported(const ConvDims& d);
bool gemm_wgrad_supported(const ConvDims& d);
size_t gemm_ws_bytes(const ConvDims& d);
int conv_fwd_gemm(const float* x, const float* w, const float* b, float* y, const ConvDims& d, void* ws, size_t wsb,
                  hipStream_t s);
int conv_dgrad_gemm(const float* dy, const float* w, float* dx, const ConvDims& d, void* ws, size_t wsb,
                    hipStream_t s);
int conv_wgrad_gemm(const float* x, const float* dy, float* dw, const ConvDims& d, void* ws, size_t wsb,
                    hipStream_t s);

// ---- image-staged implicit GEMM for the 2-D 4 x 4 PatchGAN layers (fwd, dgrad, wgrad), conv2d_img.hip
bool sconv_fwd_supported(const ConvDims& d);
bool sconv_dgrad_supported(const ConvDims& d);
size_t sconv_ws_bytes(const ConvDims& d);
// conv_p2d.hip: the PatchGAN's 4 x 4 stride-1 layer at Athena's batches on the split-operand arithmetic (forward, data gradient)
void p2d_set_terms(int m);
int p2d_get_terms();
bool p2d_fwd_supported(const ConvDims& d);
bool p2d_dgrad_supported(const ConvDims& d);
size_t p2d_ws_bytes(const ConvDims& d);
int conv_fwd_p2d(const float* x, const float* w, const float* bias, float* y, const ConvDims& d, void* ws, size_t wsb, hipStream_t s);
int conv_dgrad_p2d(const float* dy, const float* w, float* dx, const ConvDims& d, void* ws, size_t wsb, hipStream_t s);
// wgrad_p2d.hip: the weight gradient of the stride-1 layer on the same arithmetic
bool p2d_wgrad_supported(const ConvDims& d);
size_t p2d_wgrad_ws_bytes(const ConvDims& d);
int conv_wgrad_p2d(const float* x, const float* dy, float* dw, const ConvDims& d, void* ws, size_t wsb, hipStream_t s);
int conv_fwd_sconv(const float* x, const float* w, const float* bias, float* y, const ConvDims& d, void* ws, size_t wsb, hipStream_t s);
int conv_dgrad_sconv(const float* dy, const float* w, float* dx, const ConvDims& d, void* ws, size_t wsb, hipStream_t s);
void sconv_set_cfg(int cfg);
void sconv_set_tune(int on);
bool sconv_wgrad_supported(const ConvDims& d);
size_t sconv_wgrad_ws_bytes(const ConvDims& d);
int conv_wgrad_sconv(const float* x, const float* dy, float* dw, const ConvDims& d, void* ws, size_t wsb, hipStream_t s);

// ---- the PatchGAN's first layer, Conv2d(1 -> K <= 64, k 4, s 2, p 1) [+ LeakyReLU], as HBM streams (patchgan_edge.hip).
//      slope 1 = plain convolution; act != NULL in the gradients = g is the gradient BEHIND the LeakyReLU whose output is act
bool pg1_supported(const ConvDims& d);
size_t pg1_ws_bytes(const ConvDims& d);
int conv_fwd_pg1(const float* x, const float* w, const float* bias, float* y, const ConvDims& d, float slope, hipStream_t s);
int conv_wgrad_pg1(const float* x, const float* g, const float* act, float slope, float* dw, float* db, const ConvDims& d,
                   void* ws, size_t wsb, hipStream_t s);
int conv_dgrad_pg1(const float* g, const float* act, float slope, const float* w, float* dx, const ConvDims& d, hipStream_t s);

// ---- 1x1 convolutions on the flat voxel axis (MFMA, HBM-bound), conv_1x1.hip
bool wgrad_1x1_supported(const ConvDims& d);
size_t wgrad_1x1_ws_bytes(const ConvDims& d);
int conv_wgrad_1x1(const float* x, const float* dy, float* dw, const ConvDims& d, void* ws, size_t wsb, hipStream_t s);
bool flat_1x1_supported(const ConvDims& d);
int conv_fwd_1x1(const float* x, const float* w, const float* b, float* y, const ConvDims& d, void* ws, size_t wsb,
                 hipStream_t s);
int conv_dgrad_1x1(const float* dy, const float* w, float* dx, const ConvDims& d, void* ws, size_t wsb,
                   hipStream_t s);

// ---- many-channels -> one channel, 7^3 (VALU), conv_c1.hip
size_t conv_fwd_to1_k3_ws_bytes(int N, int D, int H, int W);
int conv_fwd_to1_k3(const float* x, const float* w, float* y, int N, int C, int D, int H, int W, void* ws, size_t wsb, hipStream_t s);
bool to1_dgrad_supported(const ConvDims& d);
int conv_dgrad_to1(const float* dy, const float* w, float* dx, const ConvDims& d, hipStream_t s);
bool to1_mfma_supported(const ConvDims& d);
size_t to1_mfma_ws_bytes(const ConvDims& d);
int conv_dgrad_to1_mfma(const float* dy, const float* w, float* dx, const ConvDims& d, void* ws, size_t wsb,
                        hipStream_t s);
// many channels -> one channel as a plain reduction (PatchGAN head), conv_c1.hip
bool k1_fwd_supported(const ConvDims& d);
bool k1_wgrad_supported(const ConvDims& d);
int conv_fwd_k1(const float* x, const float* w, const float* b, float* y, const ConvDims& d, hipStream_t s);
int conv_wgrad_k1(const float* x, const float* dy, float* dw, const ConvDims& d, hipStream_t s);
// one channel -> 64 channels weight gradient (MFMA over the tap axis), conv_c1.hip
bool c1_wgrad_supported(const ConvDims& d);
size_t c1_wgrad_ws_bytes(const ConvDims& d);
int conv_wgrad_c1(const float* x, const float* dy, float* dw, const ConvDims& d, void* ws, size_t wsb, hipStream_t s);

// ---- 16-bit (bf16 / fp16) MFMA convolutions, conv_h.hip.  Each operand comes either as the fp32 tensor (converted
//      into the workspace) or already converted to the C8 layout (xh / dyh non-null, nc_to_c8)
// conv_split.hip: fp32 3^3 convolutions as six bf16 MFMA products of a three-term split
bool s3_fwd_supported(const ConvDims& d);
bool s3_dgrad_supported(const ConvDims& d);
size_t s3_ws_bytes(const ConvDims& d);
size_t s3_tensor_bytes(int N, int C, long S);
int split3_to(const float* x, void* xs, int N, int C, long S, hipStream_t s);
int split3_into(const float* x, long xstride, void* xs, int N, int C, long S, int ctot, int c0, hipStream_t s, const unsigned* guard = nullptr);
// (S3 or H2 by the consuming layer d: conv_split.hip)
bool conv_layer_h2(const ConvDims& d);
// the explicit S3 entry points (nc_conv_*_split, operands from nc_to_s3) are three-term by definition, whatever nc_set_split_terms says
struct ForceThreeTerm { ForceThreeTerm(); ~ForceThreeTerm(); };
// RAII mark of a whole-network call (gen_nets.hip, api.hip): inside one, guard mode 1 COUNTS a flagged measured tensor instead of launching the
// three-term kernels beside the two-term ones (h2_guard_can_flip) -- ~65 near-empty launches per training step otherwise
struct NetworkScope { NetworkScope(); ~NetworkScope(); };
// RAII: the process-wide arithmetic switches (nc_set_split_terms, nc_set_h2_guard) as THIS call sees them -- sampled once when the outermost scope
// of the thread opens, so that the phases of one call (conversion / data gradient / weight gradient of conv_bwd_s3, the layers of a whole-network
// call) cannot be desynchronised by another thread moving a switch in between.  NetworkScope opens one; the per-layer entry points
// with more than one phase open their own.
struct SwitchScope { SwitchScope(); ~SwitchScope(); };
// RAII: this thread sees nc_set_split_terms(2) while it lives (kernels that exist in the two-term form only and convert their fp32 operands themselves)
struct ForceTwoTerm { ForceTwoTerm(); ~ForceTwoTerm(); int prev_terms, prev_depth; };
int frozen_terms();  // -1: no scope open on this thread
int frozen_guard();
bool h2_guard_can_flip();  // mode 2, or mode 1 outside a whole-network call
int operand_into(const ConvDims& d, const float* x, long xstride, void* xs, int N, int C, long S, int ctot, int c0, hipStream_t s);
int act_operand(const ConvDims& d, const float* x, const float* mean, const float* rstd, float slope, float* y, long ystride, void* ys, int N, int C,
                long S, int ctot, int c0, hipStream_t s);
int act_split3(const float* x, const float* mean, const float* rstd, float slope, float* y, long ystride, void* ys, int N, int C, long S,
               int ctot, int c0, hipStream_t s);
int conv_fwd_s3(const float* x, const void* xs, const float* w, const float* b, float* y, const ConvDims& d, void* ws, size_t wsb,
                hipStream_t s, void* xs_keep = nullptr, float* stats_part = nullptr);
bool epi_stats_on();  // nc_set_epi_stats / NC_EPI_STATS (conv_s3x.hip)
void epi_stats_set(int on);
int epi_stats_mode();
void s3x_w64_set(int on);  // nc_set_s3x_w64 / NC_S3X_W64 (conv_s3x.hip, k_conv_s3w)
int s3x_w64_get();
int conv_dgrad_s3(const float* dy, const void* dys, const float* w, float* dx, const ConvDims& d, void* ws, size_t wsb, hipStream_t s);
size_t s3_bwd_ws_bytes(const ConvDims& d);
int conv_bwd_s3(const float* x, const float* dy, const float* w, float* dx, float* dw, const ConvDims& d, void* ws, size_t wsb,
                hipStream_t s, int phase, const void* xs = nullptr, bool dy_guarded = false);
// api.hip: forward / backward of one layer for the whole-network training calls.  conv_fwd_keep: nc_conv_fwd; when the layer runs
// on the split-operand kernels its converted input is written to xs_keep and *kept set.  conv_bwd_keep: nc_conv_bwd with that
// tensor handed back (xs NULL: converted again).
int conv_fwd_keep(const float* x, const float* w, const float* bias, float* y, int N, int C, int D, int H, int W, int K, int ks,
                  void* ws, size_t ws_bytes, void* stream, void* xs_keep, bool* kept);
int conv_bwd_keep(const float* x, const void* xs, const float* dy, const float* w, float* dx, float* dw, int N, int C, int D, int H,
                  int W, int K, int ks, void* ws, size_t ws_bytes, void* stream);
// conv_s3x.hip: the tap-stream form of the split-operand forward / data-gradient kernel (S3 input, packed weights in wp_ws)
bool s3x_supported(int N, int Cin, int D, int H, int W, int Kout, int KS);
size_t s3x_packed_bytes(int Cin, int Kout, int KS, int NT = 3);
// NT = 2 (two-term fp16 split, three products; s3_common.hpp) from the fp32 input: ws >= s3x_h2_ws_bytes
void s3x_set_terms(int t);
int s3x_get_terms();
int conv_s3x_h2(const void* xs, const unsigned* cell_a, const unsigned* cell_b, int split_c, const float* w, const float* bias, float* y, int N,
                int Cin, int D, int H, int W, int Kout, int KS, long so, long si, int flip, unsigned* wcell, void* wp_ws, hipStream_t s,
                const unsigned* guard = nullptr, float* stats_part = nullptr);
// InstanceNorm statistics from the convolution's own epilogue (conv_s3x.hip, template parameter ST): stats_part >= s3x_stats_bytes
size_t s3x_stats_bytes(int N, int D, int H, int W, int Kout, int KS);
int s3x_stats_finalize(const float* stats_part, const float* bias, int N, int D, int H, int W, int Kout, int KS, float eps, float* mean, float* rstd,
                       hipStream_t s);
// conv_s3x.hip: Conv3d(1, 64, 7, padding 3) forward in pseudo-channel form on the two-term kernels (x fp32, measured cell, guard counted)
bool c1k7_h2_supported(const ConvDims& d);
size_t c1k7_h2_ws_bytes(const ConvDims& d);
int conv_c1k7_h2(const float* x, const float* w, const float* bias, float* y, const ConvDims& d, void* ws, size_t wsb, hipStream_t s);
size_t c1k7_h2_dgrad_ws_bytes(const ConvDims& d);
// dl_typed.hip: deep_linear_gen's layers 1 .. 5 as one position-typed 7^3 kernel (the boundary voxels' kernels and the weight-space steps)
bool dl_typed_supported(int N, int D, int H, int W);
size_t dl_typed_bytes(int N, int D, int H, int W);
const float* dl_typed_wp(const char* scratch, int N, int D, int H, int W);
float* dl_typed_dwsw(char* scratch, int N, int D, int H, int W);
int dl_typed_compose(const float* F, char* scratch, int N, int D, int H, int W, hipStream_t s);
int dl_typed_fwd_boundary(const float* act0, float* y, char* scratch, int N, int D, int H, int W, hipStream_t s);
int dl_typed_dgrad_boundary(const float* act0, const float* dy, float* g, char* scratch, int N, int D, int H, int W, hipStream_t s);
int dl_typed_p(const float* act0, const float* dy, float* Pq, char* scratch, int N, int D, int H, int W, hipStream_t s);
int conv_c1k7_h2_dgrad(const float* dy, const float* w, float* dx, const ConvDims& d, void* ws, size_t wsb, hipStream_t s);
// h2.hip: the H2 operand form (two fp16 terms of the tensor times a power of two taken from a cell)
// an H2 tensor of `elems` elements = elems * 4 bytes of units + (at this byte offset) 256 bytes of cells: [0] the cell of the channels' first
// half, [1] of the second half (a concatenation converted in two parts; equal to [0] otherwise) -- inside the elems * 6 bytes of an S3 tensor
inline size_t h2_cells_offset(size_t elems) { return (elems * 4 + 255) & ~(size_t)255; }
inline unsigned* h2_cells_of(const void* t, size_t elems) { return (unsigned*)((char*)const_cast<void*>(t) + h2_cells_offset(elems)); }
int instnorm_relu_tail_sigmoid(const float* x, const float* mean, const float* rstd, const float* w1, const float* b1, const float* w2,
                               const float* b2, float* y, int C, long S, hipStream_t s);
int instnorm_act_bwd_dbias_h2(const float* dy, const float* x, const float* mean, const float* rstd, float slope, void* dxs,
                              float* dbias, int N, int C, long S, void* ws, size_t ws_bytes, void* stream, unsigned* guard = nullptr);
// conv_split.hip: where conv_bwd_s3 keeps the range guard's words of the dY operand at the head of its workspace `ws` (behind the operand's
// S3 capacity) -- a producer that writes dY there itself (the norm backward) counts into them and conv_bwd_pre is told so (dy_guarded)
unsigned* conv_bwd_guard_words(void* ws, int N, int K, long S);
int maxpool2_h2(const void* in, void* out, int N, int C, int ctot, int D, int H, int W, hipStream_t s);
int h2_zero_cells(unsigned* cells, int n, hipStream_t s);
int h2_set_cell(unsigned* cell, float bound, hipStream_t s);
int h2_absmax(const float* x, long n, unsigned* cell, hipStream_t s, unsigned* cell2 = nullptr);
int split2h_into(const float* x, long xstride, void* xs, int N, int C, long S, int ctot, int c0, const unsigned* cell, hipStream_t s,
                 unsigned* guard = nullptr);
// ---- the RANGE GUARD of the two-term form (round 5; h2.hip).  One power of two per tensor means: an element keeps its 22+ bits only while it
// is within 2^17 of the tensor's largest magnitude (s3_common.hpp).  Outputs that mix small and large inputs do not care; outputs that see
// ONLY small ones do (a region of the volume 2^20 below an outlier elsewhere).  So wherever a cell is MEASURED (k_absmax) and the tensor is
// converted by k_split2h, that pass also counts CHUNKS (a wave's 64 voxels x 8 channels) whose largest magnitude is below 2^-17 of the cell;
// when more than 1 / kGuardShare of the non-zero chunks are, the call is FLAGGED (8 words: [kGuardLow], [kGuardAll] counts, [kGuardFlag]) and
// runs on the exact three-term kernels instead -- decided ON THE DEVICE (k_h2_guard_decide), with no host synchronisation: the operand is
// converted again into the same buffer as S3 (its capacity is the S3 tensor's), and both kernel families are launched, each leaving at
// its first instruction unless the flag says it is its turn (guard_skip).  Cells that are bounds by construction (InstanceNorm outputs)
// need no guard.  Counters: nc_h2_guard_stats.
constexpr int kGuardLow = 0, kGuardAll = 1, kGuardFlag = 2;
constexpr unsigned kGuardDrop = 17u << 23;          // float bits: 2^-17 below the cell
constexpr unsigned long long kGuardShare = 64;      // flagged when low * 64 > all
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ bool guard_skip(const unsigned* guard, int want) {  // want: 0 = a two-term kernel, 1 = a three-term one
  return guard && (int)(__builtin_nontemporal_load(guard + kGuardFlag) != 0) != want;
}
#else
__device__ bool guard_skip(const unsigned* guard, int want);
#endif
bool h2_guard_on();
int h2_guard_mode();  // 0 off; 1 (default): in-call fallback in the per-layer entry points, whole-network calls count only; 2: in-call fallback everywhere
void h2_guard_set(int on);
int h2_guard_read(unsigned long long* out4, int reset);
int h2_guard_zero(unsigned* g, hipStream_t s, int nwords = 8);
int h2_guard_decide(unsigned* ga, unsigned* gb, unsigned* prior, unsigned* flag, bool can_flip, hipStream_t s, unsigned long long total_a = 0);
int h2_to_s3_if(const void* xh, void* xs, int N, int C, long S, const unsigned* cells, const unsigned* guard, hipStream_t s);
int act_split2h(const float* x, const float* mean, const float* rstd, float slope, float* y, long ystride, void* ys, int N, int C, long S, int ctot,
                int c0, float bound, unsigned* cell, unsigned* cell2, hipStream_t s);
int act_split2h_pool(const float* x, const float* mean, const float* rstd, float slope, void* ys, void* pooled, int N, int C, int D, int H, int W,
                     int ctot, int c0, float bound, unsigned* cell, hipStream_t s);
int conv_s3x(const void* xs, const float* w, const float* bias, float* y, int N, int Cin, int D, int H, int W, int Kout, int KS, long so,
             long si, int flip, void* wp_ws, hipStream_t s, const unsigned* guard = nullptr);
bool conv_keep_supported(int N, int C, int D, int H, int W, int K, int ks);
// stats_part (nullable; two-term 3^3 layers only -- conv_layer_h2): the convolution leaves the partial InstanceNorm sums of its output there
// (s3x_stats_bytes; s3x_stats_finalize turns them into mean / rstd)
int conv_fwd_pre(const void* xs, const float* w, const float* bias, float* y, int N, int C, int D, int H, int W, int K, int ks, void* ws,
                 size_t ws_bytes, void* stream, float* stats_part = nullptr);
bool conv_bwd_pre_supported(int N, int C, int D, int H, int W, int K, int ks, bool want_dx, size_t ws_bytes);
int conv_bwd_pre(const float* x, const void* xs, const float* w, float* dx, float* dw, int N, int C, int D, int H, int W, int K, int ks,
                 void* ws, size_t ws_bytes, void* stream, bool dy_guarded = false);
// norm_act.hip: InstanceNorm + activation backward with dx written in S3 form only
bool instnorm_bwd_s3_supported(int N, int C, long S);
int instnorm_act_bwd_dbias_s3(const float* dy, const float* x, const float* mean, const float* rstd, float slope, void* dxs,
                              float* dbias, int N, int C, long S, void* ws, size_t ws_bytes, void* stream);
// convt.hip: ConvTranspose3d(k 2, s 2) forward that also (or only: y NULL) writes the S3 form of its output
bool convT_fwd_s3_supported(int N, int C, int D, int H, int W, int K);
// convt_s3.hip: the same forward on the bf16 matrix cores from an S3 input (three-term split, six products per fp32 product)
// conv_c8x.hip: the tap-stream form of the 16-bit forward / data-gradient kernel (C8 in, C8 or fp32 out; packed weights in wp_ws)
size_t c8x_packed_bytes(int Cin, int Kout, int KS);
void c8x_set_mode(int m);
int c8x_get_mode();
bool c8x_supported(int N, int Cin, int D, int H, int W, int Kout, int KS, bool fp32_out);
int conv_c8x(const void* xh, const float* w, const float* bias, float* y, void* yh, int ctot, int c0, int N, int Cin, int D, int H, int W,
             int Kout, int KS, long so, long si, int flip, int diffuse, int dt, void* wp_ws, hipStream_t s);
bool convT_s3x_supported(int N, int C, int D, int H, int W, int K);
size_t convT_s3x_ws_bytes(int C, int K);
int convT_fwd_s3x(const void* xs, const float* w, const float* bias, float* y, void* ys, int ctot, int c0, int N, int C, int D, int H, int W,
                  int K, void* ws, size_t wsb, hipStream_t s, const unsigned* h2cell = nullptr, const unsigned* xcell = nullptr);
int convT_h2_bound(const float* w, const float* bias, int C, int K, float in_bound, unsigned* cell, hipStream_t s);
int convT_fwd_split_h2(const float* x, const float* w, const float* bias, float* y, void* ys, int ys_ctot, int ys_c0, int N, int C, int D, int H,
                       int W, int K, const unsigned* out_cell, void* ws, size_t ws_bytes, hipStream_t s);
int convT_fwd_s3(const float* x, const float* w, const float* bias, float* y, void* ys, int ctot, int c0, int N, int C, int D, int H,
                 int W, int K, void* stream);
bool s3_wgrad_supported(const ConvDims& d);
// conv_split.hip: the 16x16x32 weight-gradient kernel on C8 (16-bit) operands; ws = the partial sums (c8x_wgrad_part_bytes)
bool c8x_wgrad_supported(const ConvDims& d);
size_t c8x_wgrad_part_bytes(const ConvDims& d);
int conv_wgrad_c8x(const void* xh, const void* dyh, float* dw, const ConvDims& d, int dtype, void* ws, size_t wsb, hipStream_t s);
size_t s3_wgrad_ws_bytes(const ConvDims& d);
int conv_wgrad_s3(const float* x, const void* xs, const float* dy, const void* dys, float* dw, const ConvDims& d, void* ws, size_t wsb,
                  hipStream_t s);
bool s3x_k32_supported(int N, int D, int H, int W);
bool conv_fwd_h2_k32_supported(const ConvDims& d);
int conv_fwd_h2_k32_keep(const float* x, const float* w, float* y, const ConvDims& d, void* ws, size_t wsb, hipStream_t s, void* xs_keep);
bool conv_fwd_h2_c32_supported(const ConvDims& d);
size_t conv_fwd_h2_c32_ws_bytes(const ConvDims& d);
int conv_fwd_h2_c32(const float* x, const float* w, float* y, const ConvDims& d, void* ws, size_t wsb, hipStream_t s);
bool wgrad_h2_supported(const ConvDims& d);
int conv_wgrad_h2(const float* x, const void* xs, const float* dy, const void* dys, float* dw, const ConvDims& d, void* ws, size_t wsb, hipStream_t s);
bool h_fwd_supported(const ConvDims& d);
bool h_dgrad_supported(const ConvDims& d);
bool h_wgrad_supported(const ConvDims& d);
size_t h_ws_bytes(const ConvDims& d);
int to_c8(const float* x, void* xh, int N, int C, long S, int dt, hipStream_t s);
int conv_fwd_h(const float* x, const void* xh, const float* w, const float* b, float* y, const ConvDims& d, int dt,
               void* ws, size_t wsb, hipStream_t s);
int conv_dgrad_h(const float* dy, const void* dyh, const float* w, float* dx, const ConvDims& d, int dt, void* ws,
                 size_t wsb, hipStream_t s);
int conv_wgrad_h(const float* x, const void* xh, const float* dy, const void* dyh, float* dw, const ConvDims& d, int dt,
                 void* ws, size_t wsb, hipStream_t s);
int conv_fwd_h_c8(const void* xh, const float* w, const float* b, void* yh, int ctot, int c0, const ConvDims& d, int dt,
                  void* ws, size_t wsb, hipStream_t s);
int conv_fwd_h_na1(const void* xh, const float* w, float* y, const ConvDims& d, int dt, void* ws, size_t wsb, hipStream_t s);
int conv_dgrad_h_c8(const void* dyh, const float* w, void* dxh, int ctot, int c0, const ConvDims& d, int dt, void* ws,
                    size_t wsb, hipStream_t s);
// deep_linear_gen's collapsed tail (gen_nets.hip, "the collapsed tail"): the weight-space steps, shared with the 16-bit path.  tail: dl_tail_bytes()
// of device scratch; compose fills E / Ef ([64][27] fp32: the 64 -> 1 kernel and its tap-flipped copy); q: [64][27], the caller's one-channel
// weight gradient (x := dy, dY := act1); grads turns q into dW2 .. dW5
bool dl_collapse_on();
size_t dl_tail_bytes();
const float* dl_tail_E(const char* tail);
const float* dl_tail_Ef(const char* tail);
float* dl_tail_q(char* tail);
int dl_tail_compose(const float* w2, const float* w3, const float* w4, const float* w5, char* tail, hipStream_t s);
int dl_tail_grads(const float* w2, const float* w3, const float* w4, const float* w5, char* tail, float* dw2, float* dw3, float* dw4, float* dw5,
                  hipStream_t s);
// layer 1's weight gradient from the rank structure of its dY (gen_nets.hip): P = the 32 x 64 weight gradient [64][32][125] the caller computed
float* dl_tail_P(char* tail);
const float* dl_w1_fold(char* tail, const float* w1, hipStream_t s);  // [64][32][125]: layer 1's data gradient as a forward convolution of Dsh
int dl_w1_contract(const char* tail, float* dw1, hipStream_t s);
int dl_q_from_p(char* tail, const float* w1, hipStream_t s);
const float* dl_fold_fwd64(char* tail, const float* w1, hipStream_t s);  // F[t][c][s] = sum_c' E[c'][t] W1[c'][c][s] as [64][64][125], rows >= 27 zero
int dl_combine27(const float* Z, float* y, int N, int D, int H, int W, int zch, hipStream_t s);  // y[v] = sum_t [v + t - 1 inside] Z[t][v + t - 1]  // q (tap-flipped, into the tail scratch) as a contraction of P and W1
// one-channel KS^3 layers (KS = 3, 7) in "pseudo-channel" form on the 16-bit cores (conv_h.hip)
bool c1_h_supported(int D, int H, int W, int KS);
size_t c1_h_ws_bytes(int N, int D, int H, int W, int KS);
int conv_c1_fwd_h(const float* x, const float* w, const float* bias, void* yh, int ctot, int c0, int N, int D, int H, int W, int KS,
                  int dt, void* ws, size_t wsb, hipStream_t s);
int conv_c1_dgrad_h(const void* dyh, const float* w, float* dx, int N, int D, int H, int W, int KS, void* ws, size_t wsb,
                    hipStream_t s);

// 16-bit weight rounding for the calls that follow on this host thread: 0 round-to-nearest, 1 tap-diffused (conv_h.hip)
void h_set_weight_diffusion(int on);

// weight gradient of the one-channel KS^3 layers (KS = 3, 7) on the 16-bit cores from the C8 gradient (c1_wgrad_h.hip)
bool c1_wgrad_h_supported(int N, int D, int H, int W, int KS);
size_t c1_wgrad_h_ws_bytes(int N, int D, int H, int W, int KS);
int conv_c1_wgrad_h(const float* x, const void* dyh, float* dw, int N, int D, int H, int W, int KS, void* ws, size_t wsb,
                    hipStream_t s);

// the fwd/dgrad MFMA kernel prefetches packed weights one kernel row ahead: slack behind the packed stream
static constexpr size_t kPackSlackBytes = 128 * 1024;

extern int g_force_direct;

// Live launch profiler (nc_prof_begin / nc_prof_end, api.hip): brackets one convolution launch with HIP events on its
// stream when profiling is on.  cls = op (0 fwd, 1 dgrad, 2 wgrad) | path << 4 | kernel edge << 8 | (16-bit ? 1 << 16 : 0)
struct ProfScope {
  int idx;
  hipStream_t s;
  ProfScope(int op, int path, const ConvDims& d, int lp, hipStream_t stream);
  ~ProfScope();
};

}  // namespace nc
