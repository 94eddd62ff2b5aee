// The bf16x3 dense-layer kernel (template) and its launcher: shared by gemm_bf16x3.hip (row-major tensors, every fused form) and
// gemm_bf16x3_g16.hip (the G16-layout instantiations) -- two translation units, so that the second set of instantiations does
// not lengthen the first one's five-minute compile.  Design notes: the head of gemm_bf16x3.hip.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "k4.h"
#include "pack_bodies.h"
#include "prof.h"

namespace nsdp {
extern int g_x3_dbg;                         // experiment knob (nsdp_debug_set(6, v))
extern thread_local int g_x3_side_reserve;   // (per host thread) compute units a side-stream launch leaves free (host hint 9: it runs on the weight-gradient side stream)
}  // namespace nsdp

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

using nsdp::g_x3_dbg;
using nsdp::g_x3_side_reserve;

#ifdef NSDP_X3_TIMING
// phase timers (s_memtime ticks summed over waves): 0 steps, 1 bottom wait, 2 barrier, 3 epilogue, 4 tile prologue, 5 total
__device__ unsigned long long g_x3_timers[8];
#define X3_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define X3_ADD(i, a, b) t_acc[i] += (b) - (a)
#else
#define X3_T(var)
#define X3_ADD(i, a, b)
#endif

struct X3Params {
  const float *X;
  const void *Wp;  // bf16x3 pack
  const float *bias, *residual, *mask, *out_mask;
  float *Y;
  long long M;
  int N, K;
  int relu_in, relu_out;
  int dbg;  // experiment knob (nsdp_debug_set(6, v)): bit 0 no weight DMA in the loop, bit 3 no stores, bit 9 one LDS weight
            // read per step instead of three -- wrong results, timing only
  // gathered addend (nsdp_linear_bf16x3_gather_f32, GATHER forms): Y[r] += gq[r / g_div] - gk[(r / g_rps) * g_nsrc + gidx[r]]
  // (rows of two small L2-resident tables) -- the "q_i - k_j" of a vector-attention block added by the position-encoding MLP's
  // last layer itself, so that u = q - k + pos comes out of the GEMM and the attn_pre pass (read pos, write u) disappears
  const float *gq, *gk;
  const int32_t *gidx;
  int g_div, g_rps, g_nsrc;
  float res_sign = 1.f;   // the residual enters as res_sign * residual (nsdp_linear_bf16x3_signed_f32: -1 = "minus a table")
  // added AFTER the output mask (nsdp_linear_bf16x3_addend_f32, masked-prologue forms with an out_mask): the gradient arriving
  // over the skip connection of x + f(relu(x)), which the ReLU's mask must not touch
  const float *addend = nullptr;
  // TAIL forms (nsdp_linear_bf16x3_k4tail_f32): Y = dY W2 is the gradient of h0 = relu(x4 W0^T + b0), the hidden layer of a
  // position-encoding MLP whose input (relative coordinates) needs no gradient -- so the only reader of Y is the K = 4 layer's
  // weight gradient dW0 = (Y o [h0 > 0])^T x4, db0 = its column sums.  The epilogue forms them itself: Y is never stored, the
  // ReLU mask is recomputed from the 16-byte input rows (k4.h), every wave writes ONE partial (80 floats per n tile) per row tile to t_ws
  const float *t_x4 = nullptr, *t_w0 = nullptr, *t_b0 = nullptr;
  float *t_ws = nullptr;
  // H0 forms (PRE == 3, nsdp_linear_bf16x3_h0_f32): the activation operand is the hidden layer of a position-encoding MLP,
  // h0 = relu(x4 W0^T + b0) [M, K], and is never materialised -- the operand producer recomputes it from the 16-byte coordinate
  // rows (k4.h: the very expression of the K = 4 forward kernel, so the values are the ones that kernel would have stored)
  // instead of streaming [M, K] floats from HBM.  X is unused.  h_w0 [K, 4] row-major zero-padded, h_b0 [K] or NULL.
  const float *h_x4 = nullptr, *h_w0 = nullptr, *h_b0 = nullptr;
  // ReLU bits (see below): PRE == 4 reads `bits` as the mask of X; a G16 output with relu_out writes `bits_out` when non-null
  const unsigned char *bits = nullptr;
  unsigned char *bits_out = nullptr;
};

// G16 layout (the LAY template parameter of the kernel; nsdp_linear_bf16x3_g16_f32, gemm_bf16x3_g16.hip): a tensor [M, C]
// (M % 16 == 0, C % 4 == 0) stored as [M / 16][C / 4][16 rows][4 floats] -- groups of 16 rows, channel-quad-major inside a
// group.  Lane (row li, lane group g) of this kernel's fragment convention finds the 16-byte piece (row, quad q) at
// group base + q * 256 B + li * 16 B, so the 16 rows x 64 B that a wave-wide activation load or an accumulator tile's store
// covers are ONE contiguous KiB in lane order li + 16 g (row-major: 16 runs of 64 B, 800 or 1024 B apart -- the address
// pattern was measured at 14 % (loads) + 15 % (stores) of the 200-wide launches, docs/EXPERIMENTS.md round 4).  The layout of
// the tensors BETWEEN two of these kernels: hidden layers of the attention MLPs and their gradients, which nothing else reads.
// ReLU bits (PRE == 4 / X3Params::bits): the mask of a hidden layer h = relu(.) [M, H] as ONE BIT per element instead of the fp32
// tensor itself -- [M / 16][ceil(H / 32)][64] bytes: the byte of (row group, k block kb, lane (li, g)) holds [h > 0] for row li and
// the eight channels this kernel's fragment convention gives that lane in block kb (bits 0-3: 32 kb + 4 g .. + 3, bits 4-7:
// 32 kb + 16 + 4 g .. + 3).  Written by the epilogue of the layer that produces h in G16 (bits_out), read by the masked prologue of
// its dX GEMM (PRE == 4: one byte load per row tile and k block where PRE == 1 streams two float4) and by its weight gradient:
// 28 B per row of a 200-wide layer instead of 800.
constexpr int kLayX = 1;      // X (and the PRE == 1 mask)
constexpr int kLayY = 2;      // Y: stored straight from the accumulators, one contiguous KiB per tile, no LDS staging
constexpr int kLayR = 4;      // the residual (it only initialises the accumulators: independent of Y's layout)

// two fp32 values -> the packed (lo, hi) bf16 pairs of their three split planes
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
  const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
  const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
}

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// Sums over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15) of 20 values at once, the totals in every lane; fixed order:
// pairs, quads, halves, row.  One v_add_f32 with a DPP operand per value and step -- left to the compiler this became
// v_mov 0 / v_mov_dpp / v_pk_add (2.5 instructions per step).  Written as four blocks of 20 independent instructions: the
// two wait states a DPP read needs after a VALU write of the same register are covered by the s_nop at the head of a block
// (the compiler's hazard recognizer does not look inside inline asm) and by the 19 other instructions within it.
#define NSDP_DPP4(CTRL, A, B, C, D)                                                                                              \
  asm("s_nop 1\n\t"                                                                                                              \
      "v_add_f32_dpp %0, %0, %0 " CTRL "\n\tv_add_f32_dpp %1, %1, %1 " CTRL "\n\tv_add_f32_dpp %2, %2, %2 " CTRL "\n\t"            \
      "v_add_f32_dpp %3, %3, %3 " CTRL "\n\ts_nop 1"                                                                              \
      : "+v"(A), "+v"(B), "+v"(C), "+v"(D))
#define NSDP_DPP16(CTRL, T)                                                                                                      \
  asm("s_nop 1\n\t"                                                                                                              \
      "v_add_f32_dpp %0, %0, %0 " CTRL "\n\tv_add_f32_dpp %1, %1, %1 " CTRL "\n\tv_add_f32_dpp %2, %2, %2 " CTRL "\n\t"            \
      "v_add_f32_dpp %3, %3, %3 " CTRL "\n\tv_add_f32_dpp %4, %4, %4 " CTRL "\n\tv_add_f32_dpp %5, %5, %5 " CTRL "\n\t"            \
      "v_add_f32_dpp %6, %6, %6 " CTRL "\n\tv_add_f32_dpp %7, %7, %7 " CTRL "\n\tv_add_f32_dpp %8, %8, %8 " CTRL "\n\t"            \
      "v_add_f32_dpp %9, %9, %9 " CTRL "\n\tv_add_f32_dpp %10, %10, %10 " CTRL "\n\tv_add_f32_dpp %11, %11, %11 " CTRL "\n\t"      \
      "v_add_f32_dpp %12, %12, %12 " CTRL "\n\tv_add_f32_dpp %13, %13, %13 " CTRL "\n\tv_add_f32_dpp %14, %14, %14 " CTRL "\n\t"   \
      "v_add_f32_dpp %15, %15, %15 " CTRL "\n\ts_nop 1"                                                                           \
      : "+v"(T[0]), "+v"(T[1]), "+v"(T[2]), "+v"(T[3]), "+v"(T[4]), "+v"(T[5]), "+v"(T[6]), "+v"(T[7]), "+v"(T[8]), "+v"(T[9]),   \
        "+v"(T[10]), "+v"(T[11]), "+v"(T[12]), "+v"(T[13]), "+v"(T[14]), "+v"(T[15]))
// t[0 .. 15]: the products d * x4[k] (index 4 c + k), t[16 .. 19]: the column sums.  K3: x4[3] is zero padding -- the products
// 4 c + 3 are zeros and stay out of it (16 values instead of 20)
template <bool K3>
__device__ __forceinline__ void row16_sum20(float (&t)[20]) {
  if constexpr (K3) {
    float u[16] = {t[0], t[1], t[2], t[4], t[5], t[6], t[8], t[9], t[10], t[12], t[13], t[14], t[16], t[17], t[18], t[19]};
    NSDP_DPP16("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", u);
    NSDP_DPP16("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf", u);
    NSDP_DPP16("row_half_mirror row_mask:0xf bank_mask:0xf", u);
    NSDP_DPP16("row_mirror row_mask:0xf bank_mask:0xf", u);
    t[0] = u[0]; t[1] = u[1]; t[2] = u[2]; t[4] = u[3]; t[5] = u[4]; t[6] = u[5]; t[8] = u[6]; t[9] = u[7]; t[10] = u[8];
    t[12] = u[9]; t[13] = u[10]; t[14] = u[11]; t[16] = u[12]; t[17] = u[13]; t[18] = u[14]; t[19] = u[15];
  } else {
    NSDP_DPP16("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", t);
    NSDP_DPP4("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", t[16], t[17], t[18], t[19]);
    NSDP_DPP16("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf", t);
    NSDP_DPP4("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf", t[16], t[17], t[18], t[19]);
    NSDP_DPP16("row_half_mirror row_mask:0xf bank_mask:0xf", t);
    NSDP_DPP4("row_half_mirror row_mask:0xf bank_mask:0xf", t[16], t[17], t[18], t[19]);
    NSDP_DPP16("row_mirror row_mask:0xf bank_mask:0xf", t);
    NSDP_DPP4("row_mirror row_mask:0xf bank_mask:0xf", t[16], t[17], t[18], t[19]);
  }
}
#undef NSDP_DPP16
#undef NSDP_DPP4

// hand-issued activation loads (the compiler would sink them to their first use, see decoder_fused.hip)
__device__ __forceinline__ void xload(f32x4 &dst, const float *lane_ptr) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(lane_ptr));
}
__device__ __forceinline__ void bload(unsigned &dst, const unsigned char *lane_ptr) {      // one byte of ReLU bits, zero-extended
  asm volatile("global_load_ubyte %0, %1, off" : "=v"(dst) : "v"(lane_ptr));
}


// weight fragments come back from LDS through hand-placed ds_read_b128 (the compiler sinks ordinary LDS loads
// below the MFMA block of a step, exposing their latency every step); lgkmcnt is awaited by hand, the fragment
// registers being in/out operands of the wait so that their users depend on it
template <int OFF>
__device__ __forceinline__ void lds_read(u32x4 &dst, unsigned lane_addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lane_addr), "n"(OFF));
}
__device__ __forceinline__ void lds_wait(u32x4 &a, u32x4 &b, u32x4 &c) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c));
}

// helpers of the staged epilogue (immediate offsets, counted waits; see linear_bf16x3_kernel)
typedef __attribute__((address_space(3))) f32x4 *lds_f4_ptr;
// (the epilogue's LDS traffic is plain C++: under register pressure the compiler parks the destination of a hand-issued
// ds_read in an AGPR right after the asm statement -- a copy of a register whose load is still in flight -- and later restores
// the stale copy; its own loads it waits for correctly)
template <int OFF>
__device__ __forceinline__ void lds_read_f4(f32x4 &dst, unsigned addr) {
  dst = *reinterpret_cast<lds_f4_ptr>(static_cast<uintptr_t>(addr + static_cast<unsigned>(OFF)));
}
// (stores are left to the compiler: an inline-asm ds_write / global_store is invisible to its hazard recognizer, which must
// keep the next VALU write of the DATA registers one or two wait states away from a > 64-bit store -- the hand-written form
// lost dword 0 of a chunk now and then)
template <int OFF>
__device__ __forceinline__ void lds_write_f4(unsigned addr, f32x4 v) {
  *reinterpret_cast<lds_f4_ptr>(static_cast<uintptr_t>(addr + static_cast<unsigned>(OFF))) = v;
}
template <int CNT>
__device__ __forceinline__ void lgkm_wait(f32x4 &) {}      // (the compiler waits for its own LDS loads)
__device__ __forceinline__ void store_f4(unsigned byte_off, f32x4 v, float *uniform_base) {
  *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(uniform_base) + byte_off) = v;
}

// WV waves per workgroup: 4 (one per SIMD, MT up to 4 row tiles: 512 registers per lane) or 8 (two per SIMD, MT <= 2:
// 256 registers per lane -- the second wave of a SIMD issues MFMAs while the first splits, stores or waits)
// XREG: raw activations through registers even without a mask (frees the 32 KiB X staging: at 13 n tiles two 4-wave
// workgroups then fit into one CU's LDS)
// WRES (weights RESIDENT): all KBM k blocks of the three weight planes are DMA'd into LDS ONCE per workgroup (N, K <= 128:
// 4 x 8 x 3 KiB = 96 KiB) and stay there -- no weight DMA and no workgroup barrier inside the k loop, so the eight waves
// drift apart and one wave's epilogue stores sit under the other waves' MFMA steps (the streaming form re-fetches the
// planes L2 -> LDS for every 256-row tile: as many bytes as the HBM traffic, and its per-k-block barrier keeps all waves
// in the same phase).
template <int MT, int NT, int PRE, int WV, bool XREG = false, int KBM = 2, int GATHER = 0, int TAIL = 0, int LAY = 0>
__global__ __launch_bounds__(WV * 64, XREG ? 2 : 1) void linear_bf16x3_kernel(X3Params p) {
  constexpr bool kXG = (LAY & kLayX) != 0, kYG = (LAY & kLayY) != 0, kRG = (LAY & kLayR) != 0;
  static_assert(!kYG || (!GATHER && !TAIL), "G16 output: plain epilogue forms");
  static_assert(!kXG || PRE != 3, "G16 input: there is no X in the H0 forms");
  static_assert(PRE != 4 || kXG, "ReLU bits belong to G16 operands");
  constexpr bool kMasked = PRE == 1 || PRE == 4;      // the masked prologues keep the raw activations in registers
  static_assert(!GATHER || PRE == 0 || PRE == 3, "the gathered addend belongs to the plain-prologue forms");
  static_assert(!TAIL || (PRE == 0 && !GATHER), "the K = 4 tail belongs to the plain-prologue forms");
  constexpr bool WRES = KBM > 2;
  // H0: the operand is recomputed from 16-byte coordinate rows (see X3Params::h_x4): no activation DMA, no staging, no raw registers
  constexpr bool kH0 = PRE == 3;
  // PRE != 1: the raw fp32 activations go global -> LDS by DMA as well (wave-private 8 KiB pieces, two k blocks
  // deep): no registers in flight, issued a whole k block earlier.  PRE == 1 (activation + mask) would not fit
  // in LDS next to the weights and keeps the register path.
  constexpr bool kXLds = !kMasked && !XREG && !kH0;
  // LDS: [weight buffer 0][epilogue extension][weight buffer 1] (streaming form) + the activation staging.
  // STAGED EPILOGUE (streaming form): the output tile of a wave goes to HBM through LDS -- accumulators (lane = row li, four
  // columns) are written row-major into a per-wave piece of the weight buffer that the tile's last k block has just released
  // plus the extension, read back as consecutive 16-byte chunks and stored so that one store instruction covers KiB-sized runs
  // of the output rows instead of 16 rows x 64 B.  Ablation with contiguous (wrong) store addresses: -15 % on the 200- and
  // 256-wide launches; WRITE_SIZE was 1.14 x the output bytes because every other row's 64-byte segment straddled two
  // memory blocks (800-byte rows).  The extension is what the CU's LDS has left (two workgroups per CU: 80 KiB each).
  constexpr int kWTile = NT * 3 * 64;                                        // u32x4 per weight buffer
  constexpr int kXTile = kXLds ? 2 * WV * MT * 2 * 64 : 1;                   // u32x4 of activation staging
  constexpr bool kTwoPerCu = XREG || (NT <= 8 && WV == 4);                   // (launch_x3: these run two workgroups per CU)
  constexpr int kCap = (kTwoPerCu ? 80 : 160) * 64;                          // LDS budget in u32x4
  constexpr int kBiasLds = NT * 4;                                           // u32x4: the bias vector, staged once per workgroup
  constexpr int kH0Rows = kH0 ? ((NT * 16 + 31) / 32) * 32 : 0;              // rows of the K = 4 layer's table (whole k blocks)
  constexpr int kH0XRows = kH0 ? 2 * WV * MT * 16 : 0;                       // coordinate rows of this tile and the next, per wave
  constexpr int kH0Lds = kH0Rows + kH0Rows / 4 + kH0XRows;                   // u32x4: the K = 4 layer pair-wise (k4.h), its biases, the rows
  constexpr int kExtFree = kCap - KBM * kWTile - kXTile - kBiasLds - kH0Lds;
  constexpr bool kStage = !WRES && kExtFree >= 0 && !TAIL && !kYG;                   // (resident weights fill the LDS: direct epilogue; the K = 4 tail stores no tile)
  constexpr int kExtWant = WV * NT * 64 - kWTile;                            // whole 16-row tiles for every wave
  constexpr int kExt = !kStage ? 0 : (kExtWant < 0 ? 0 : (kExtFree < 0 ? 0 : (kExtWant < kExtFree ? kExtWant : kExtFree)));
  constexpr int kPerWave = (kWTile + kExt) / WV;                             // u32x4 of epilogue staging per wave
  // 16-column tiles per pass (1 KiB per tile), at most 8: a pass keeps ~4 float4 per tile in flight (addend / mask chunks, bias,
  // read-back) next to the accumulators -- the forms with room for a whole 13- or 16-tile row in one pass (the H0 forms: no
  // activation staging) spilled ~120 registers per lane and tile there, 3 GB of scratch traffic per 1.8 M-row launch
  // (profiles/r6_pmc_attribution.md: WRITE_SIZE 2.05 x the output)
  constexpr int kTppFit = kPerWave / 64 < NT ? kPerWave / 64 : NT;
  constexpr int kTppCap = (NT == 13 && (WV == 8 || XREG)) ? 5 : 8;      // (the 256-register 13-tile forms: 13 = 5 + 4 + 4; the 8-wave H0 forms also hold the K = 4 producer's state)
  constexpr int kTppMax = kTppFit < kTppCap ? kTppFit : kTppCap;
  constexpr int kPasses = kStage ? (NT + kTppMax - 1) / kTppMax : 1;
  constexpr int kTpp = (NT + kPasses - 1) / kPasses;
  static_assert(!kStage || kTppMax >= 1, "epilogue staging: no room for a 16 x 16 tile per wave");
  __shared__ __attribute__((aligned(16))) u32x4 wlds[KBM * kWTile + kExt];
  __shared__ __attribute__((aligned(16))) u32x4 xbuf[kXLds ? 2 : 1][kXLds ? WV : 1][kXLds ? MT * 2 * 64 : 1];
  __shared__ __attribute__((aligned(16))) float bias_lds[(kStage || TAIL) ? NT * 16 : 4];      // (zeros without a bias: read unconditionally)
  // K = 4 tail: the K = 4 layer's weight rows [n][4] and (in bias_lds) its bias, staged once per workgroup; the 16-byte input rows
  // of the wave's current tile, DMA'd at the tile's start (one KiB per wave: lane l holds row min(l, MT * 16 - 1))
  __shared__ __attribute__((aligned(16))) f32x4 t_w0lds[TAIL ? NT * 16 : 1];
  __shared__ __attribute__((aligned(16))) f32x4 t_x4lds[TAIL ? WV * 64 : 1];
  __shared__ __attribute__((aligned(16))) f32x4 h_w0lds[kH0 ? kH0Rows : 1];
  __shared__ __attribute__((aligned(16))) float h_b0lds[kH0 ? kH0Rows : 4];
  __shared__ __attribute__((aligned(16))) f32x4 h_x4lds[kH0 ? 2 : 1][kH0 ? WV : 1][kH0 ? MT * 16 : 1];
  static_assert(!TAIL || !WRES, "the K = 4 tail needs 1 KiB of LDS per wave and 272 B per n tile next to the weight buffers");
  static_assert(!TAIL || MT * 16 <= 64, "the K = 4 tail stages one input row per lane");
  // buffer b of the weight ring: the extension sits between buffers 0 and 1, so that whichever of the two is free forms one
  // contiguous region with it
  auto wbuf_at = [&](int b) -> u32x4 * { return wlds + b * kWTile + (b >= 1 ? kExt : 0); };
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int K = p.K, N = p.N;
  const int KB = (K + 31) >> 5;                // >= 2 (host contract)
  const int ntiles = (N + 15) >> 4;            // n tiles present in the pack (<= NT)
  constexpr long long kRowsWg = static_cast<long long>(WV) * MT * 16;
  const long long wg_tiles = (p.M + kRowsWg - 1) / kRowsWg;
  const long long stride = gridDim.x;
  long long tile = blockIdx.x;                 // persistent workgroup: tile, tile + grid, ...

  // per-lane activation rows of a tile, clamped into the tensor (rows >= M are computed and never stored)
  const float *xa[MT], *xn[MT];
  const float *ma[MT], *mn[MT];      // (PRE == 4: byte pointers into the ReLU bits, carried as float pointers)
  auto set_rows = [&](long long t, const float **x, const float **m) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      long long r = (t * WV + wave) * (MT * 16) + mt * 16 + li;
      r = r < p.M ? r : (p.M - 1);
      // (G16: the row's 16-byte piece of quad 0; quad q is 64 floats further -- see xissue)
      const long long ro = kXG ? (r & ~15LL) * K + (r & 15) * 4 : r * K;
      x[mt] = p.X + ro;
      m[mt] = PRE == 1 ? p.mask + ro : nullptr;
      if constexpr (PRE == 4)      // byte (row group, kb = 0, lane (li, g)); k block kb is 64 bytes further
        m[mt] = reinterpret_cast<const float *>(p.bits + (r >> 4) * (static_cast<long long>((K + 31) >> 5) * 64) + (r & 15) * 4 + g);
    }
  };
  set_rows(tile, xa, ma);
  set_rows(tile + stride, xn, mn);
  // H0: the wave's coordinate rows of the current tile and of the next one (whose first k block is produced inside this tile's
  // last) sit in LDS -- DMA'd a tile ahead, no registers in flight; h_xs: the rows the producer of the k block being split
  // works on (lane li: row 16 mt + li), h_par: which of the two row buffers holds the current tile
  f32x4 h_xs[kH0 ? MT : 1];
  int h_kb = 0;
  unsigned h_par = 0;
  auto h_rows = [&](long long t, unsigned buf) {      // lane l < MT * 16: row l of the wave's tile t
    if (lane < MT * 16) {
      long long r = (t * WV + wave) * (MT * 16) + lane;
      r = r < p.M ? r : (p.M - 1);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.h_x4 + r * 4), (lds_ptr_t)(&h_x4lds[buf][wave][0]), 16, 0, 0);
    }
  };
  auto h_take = [&](unsigned buf) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) h_xs[mt] = h_x4lds[buf][wave][mt * 16 + li];
  };
  if constexpr (kH0) {
    h_rows(tile, 0u);
    h_rows(tile + stride, 1u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    h_take(0u);
  }

  const char *wlane = static_cast<const char *>(p.Wp) + lane * 16;
  auto stage = [&](int kb, int buf) {   // DMA one k block of weight pieces (1 KiB each), spread over the 4 waves
    const int pieces = ntiles * 3;
    for (int q = wave; q < pieces; q += WV)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wlane + ((static_cast<long long>(kb) * pieces + q) << 10)),
                                       (lds_ptr_t)(wbuf_at(buf) + q * 64), 16, 0, 0);
  };
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_ptr_t)(&wlds[lane])));
  const unsigned ldsw = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_ptr_t)(&wlds[0])));
  const unsigned ldsb = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_ptr_t)(&bias_lds[0])));
  constexpr unsigned kBufBytes = NT * 3 * 1024;
  constexpr unsigned kExtBytes = kExt * 16;

  // raw activations of one k block: [mt][half] = 4 consecutive k each (k = 32 kb + 16 half + 4 g ..)
  f32x4 raw[kXLds ? 1 : MT][2], rawm[kXLds ? 1 : MT][2];
  unsigned rawb[PRE == 4 ? MT : 1];      // (PRE == 4) the k block's ReLU bits of this lane's row: bits 0-3 half 0, bits 4-7 half 1
  auto xissue = [&](const float *const *x, const float *const *m, int kb, unsigned xb) {
    if constexpr (kH0) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        int ko = kb * 32 + 16 * hf + 4 * g;     // the four lane groups of a row read 64 contiguous bytes per instruction
        ko = ko < K ? ko : (K - 4);     // past the row end: re-read in-row data (the packed weights are zero there)
        if constexpr (kXG) ko *= 16;    // (floats between consecutive channel quads: 16 rows x 4)
        if constexpr (kXLds) {
          const float *src = x[mt] + ko;
          // (ablation knob 1024, timing only, wrong results: every DMA instruction reads ONE contiguous KiB of the wave's rows
          // instead of 16 rows x 64 B -- what the address pattern costs: 14 % of the 200-wide launch.  Eight rows x 128 B per
          // instruction, built and measured in round 4, bought nothing: at an 800-byte row pitch a 128-byte run straddles two
          // cache lines, so an instruction still touches 16 lines)
          if (p.dbg & 1024) src = x[0] - li * K + (((kb * MT * 2 + mt * 2 + hf) * 256) % 6144) + lane * 4;
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(&xbuf[xb][wave][(mt * 2 + hf) * 64]), 16, 0, 0);
        } else {
          xload(raw[mt][hf], x[mt] + ko);
          if constexpr (PRE == 1) xload(rawm[mt][hf], m[mt] + ko);
          if constexpr (PRE == 4) {
            if (hf == 0) bload(rawb[mt], reinterpret_cast<const unsigned char *>(m[mt]) + kb * 64);
          }
        }
      }
  };
  auto xwait = [&]() {   // all outstanding vector memory operations (DMA included)
    if constexpr (kXLds || kH0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {   // the raw registers become data-dependent on the wait
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[mt][0]), "+v"(raw[mt][1]));
        if constexpr (PRE == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rawm[mt][0]), "+v"(rawm[mt][1]));
        if constexpr (PRE == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rawb[mt]));
      }
    }
  };
  struct Planes {
    u32x4 h[MT], m[MT], l[MT];
  };
  auto convert_pair = [&](Planes &pl, int mt, int pr, unsigned xb) {   // pr = 0..3: values 2 pr, 2 pr + 1 of the lane's 8
    if constexpr (kH0) {
      // k = 32 kb + 16 (pr / 2) + 4 g + 2 (pr % 2), + 1: two rows of the K = 4 layer's table (zeros beyond K: h0 = relu(0) = 0 there,
      // against zero weights).  The row tiles of a pair index share these reads (the pair order below is pr-major)
      const int k0 = h_kb * 32 + 16 * (pr >> 1) + 4 * g + 2 * (pr & 1);      // (even: one pair of the table)
      const f32x4 wa = h_w0lds[k0], wb = h_w0lds[k0 + 1];
      const f32x2 bb = *reinterpret_cast<const f32x2 *>(&h_b0lds[k0]);
      const float4 xq = make_float4(h_xs[mt][0], h_xs[mt][1], h_xs[mt][2], h_xs[mt][3]);
      const f32x2 pre = nsdp::k4_preact_pair(xq, f32x2{wa[0], wa[1]}, f32x2{wa[2], wa[3]}, f32x2{wb[0], wb[1]}, f32x2{wb[2], wb[3]}, bb);
      unsigned h, m, l;
      split_pair(fmaxf(pre[0], 0.f), fmaxf(pre[1], 0.f), h, m, l);
      pl.h[mt][pr] = h; pl.m[mt][pr] = m; pl.l[mt][pr] = l;
      return;
    }
    f32x4 v;
    if constexpr (kXLds) {
      v = __builtin_bit_cast(f32x4, xbuf[xb][wave][(mt * 2 + (pr >> 1)) * 64 + lane]);
    } else {
      v = raw[mt][pr >> 1];
      if constexpr (PRE == 1) {
        const f32x4 mk = rawm[mt][pr >> 1];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = mk[c] > 0.f ? v[c] : 0.f;
      }
      if constexpr (PRE == 4) {      // (only the pair's two values are used below)
        const unsigned nib = rawb[mt] >> (4 * (pr >> 1));
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = ((nib >> c) & 1u) ? v[c] : 0.f;
      }
    }
    if (PRE == 2) {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = __builtin_amdgcn_fmed3f(v[c], 0.f, __builtin_inff());   // max(v, 0), one VALU op
    }
    unsigned h, m, l;
    split_pair(v[2 * (pr & 1)], v[2 * (pr & 1) + 1], h, m, l);
    pl.h[mt][pr] = h; pl.m[mt][pr] = m; pl.l[mt][pr] = l;
  };

  // pair i of a k block's split -> (row tile, value pair): row-tile-major, H0 value-pair-major (see convert_pair)
  auto pair_mt = [](int i) { return kH0 ? i % MT : i >> 2; };
  auto pair_pr = [](int i) { return kH0 ? i / MT : i & 3; };
  Planes cur, nxt;
  if constexpr (kH0) {
    for (int c = threadIdx.x; c < kH0Rows; c += WV * 64) h_b0lds[c] = (p.h_b0 && c < K) ? p.h_b0[c] : 0.f;
    for (int j = threadIdx.x; j < kH0Rows / 2; j += WV * 64) nsdp::k4_pair_table(p.h_w0, K, j, h_w0lds[2 * j], h_w0lds[2 * j + 1]);
    __syncthreads();      // (the prologue below already splits the first k block)
  }
  if constexpr (kStage) {
    for (int c = threadIdx.x; c < NT * 16; c += WV * 64) bias_lds[c] = (p.bias && c < p.N) ? p.bias[c] : 0.f;      // visible after the prologue's barrier
  }
  if constexpr (TAIL) {
    for (int c = threadIdx.x; c < NT * 16; c += WV * 64) {
      bias_lds[c] = (p.t_b0 && c < p.N) ? p.t_b0[c] : 0.f;
      t_w0lds[c] = c < p.N ? *reinterpret_cast<const f32x4 *>(p.t_w0 + static_cast<long long>(c) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // stagger the persistent workgroups by eighths of a tile time (~640 cycles per k block): all of them run the same
  // program on the same amount of work, and without it their epilogue store bursts hit HBM at the same moments
  // (measured: -7 % time on the 1.8 M-row layers).  Only worth it when a workgroup has several tiles to go.
  if (wg_tiles >= 4 * stride) {
    // (two workgroups per CU: the second half of the grid is shifted by half a tile against the first)
    int phase = static_cast<int>(blockIdx.x & 7);
    if (WV == 4 && gridDim.x > 256 && blockIdx.x >= gridDim.x / 2) phase = (phase + 4) & 7;
    for (int i = 0; i < phase * KB; ++i) __builtin_amdgcn_s_sleep(10);
  }
  // prologue (once per workgroup): block 0 of the first tile, split; block 1 in flight
  if constexpr (WRES) {
    for (int kb = 0; kb < KB; ++kb) stage(kb, kb);
  } else {
    stage(0, 0);
  }
  xissue(xa, ma, 0, 0u);
  if constexpr (kXLds) xissue(xa, ma, 1, 1u);
  xwait();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) convert_pair(cur, mt, pr, 0u);
  if constexpr (!kXLds) {
    xissue(xa, ma, 1, 0u);
    xwait();
  }
  __syncthreads();

  // the split of the next k block's activations is spread over the first kConvSteps n-tile steps of a block
  constexpr int kConvSteps = NT > 4 ? 4 : NT - 1;
  constexpr int kConvFirst = kXLds ? NT - kConvSteps : 0;     // first n-tile step that carries split work
  constexpr int kPairs = MT * 4;
  constexpr int kPerStep = (kPairs + kConvSteps - 1) / kConvSteps;
  constexpr int kValuPerMfma = (kPerStep * ((PRE == 1 || PRE == 4) ? 13 : PRE == 2 ? 10 : PRE == 3 ? 21 : 9) + 6 * MT - 1) / (6 * MT);

#ifdef NSDP_X3_TIMING
  unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_begin = __builtin_readcyclecounter();
#endif
  unsigned gs = 0;   // running k block count: weight buffer parity
  for (;;) {
    X3_T(t_tile0);
    const long long row0 = (tile * WV + wave) * (MT * 16);
    const bool next_tile = tile + stride < wg_tiles;
    if constexpr (TAIL) {
      // this tile's 16-byte input rows of the K = 4 layer: global -> LDS by DMA, no registers; older than every load the k loop
      // issues and waits for, so it has landed when the epilogue reads it (vector memory operations return in order)
      long long r = row0 + (lane < MT * 16 ? lane : MT * 16 - 1);
      r = r < p.M ? r : p.M - 1;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.t_x4 + r * 4), (lds_ptr_t)(&t_x4lds[wave * 64]), 16, 0, 0);
    }
    // (opaque per-tile copies of the lane coordinates: everything the prologue / epilogue derives from them is
    // tile-invariant, and LICM would otherwise keep ~60 such values live across the whole k loop)
    int li_t = li, g_t = g;
    asm volatile("" : "+v"(li_t), "+v"(g_t));
    f32x4 acc[MT][NT];
    // TRANSPOSED product D = W X^T: lane (li, g) of accumulator (mt, nt) holds row row0 + 16 mt + li, columns
    // 16 nt + 4 g .. + 3 -- four consecutive floats of Y, so residual / bias / out_mask / Y move as float4
    // GATHER: Y[r] += gq[r / g_div] - gk[(r / g_rps) * g_nsrc + gidx[r]], added in the EPILOGUE (an accumulator that started
    // from q - k, a few units, would round each of the ~40 small MFMA addends of the position encoding at the difference's
    // ulp: measured 7x the rms error of the separate pass).  Only the two row offsets (floats) of this lane's rows are
    // fetched here -- the index load is the head of a dependent chain -- and ride through the k loop.
    // GATHER == 2: ONE table that already holds the difference (p.gk = q - k per shape and source, p.gq unused): half the loads
    unsigned gqo[GATHER == 1 ? MT : 1], gko[GATHER ? MT : 1];
    if constexpr (GATHER) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        long long row = row0 + mt * 16 + li_t;
        row = row < p.M ? row : (p.M - 1);
        const unsigned r32 = static_cast<unsigned>(row);                       // (host contract: M, table elements < 2^31)
        if constexpr (GATHER == 1) gqo[mt] = (r32 / static_cast<unsigned>(p.g_div)) * static_cast<unsigned>(N);
        gko[mt] = ((r32 / static_cast<unsigned>(p.g_rps)) * static_cast<unsigned>(p.g_nsrc) + static_cast<unsigned>(p.gidx[row])) *
                  static_cast<unsigned>(N);
      }
    }
    if (!GATHER && p.residual) {  // residual add fused as the accumulator's initial value
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        long long row = row0 + mt * 16 + li_t;
        row = row < p.M ? row : (p.M - 1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          int col = nt * 16 + 4 * g_t;
          col = col + 4 <= N ? col : (N - 4);
          const long long ro = kRG ? (row & ~15LL) * N + (row & 15) * 4 + col * 16 : row * N + col;
          const float4 v = *reinterpret_cast<const float4 *>(p.residual + ro);
          acc[mt][nt] = f32x4{v.x, v.y, v.z, v.w};
          if constexpr (PRE == 0) acc[mt][nt] *= p.res_sign;      // (signed residuals come without masks / input ReLU)
        }
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    X3_T(t_tile1);
    X3_ADD(4, t_tile0, t_tile1);
    for (int kb = 0; kb < KB; ++kb, ++gs) {
      X3_T(t_k0);
      const unsigned buf = gs & 1u;                        // X staging parity (and the weight buffer of the streaming form)
      const bool more = kb + 1 < KB || next_tile;          // a k block follows (this tile's, or the next tile's first)
      if constexpr (!WRES) {
        if (more && !(p.dbg & 1)) stage(kb + 1 < KB ? kb + 1 : 0, buf ^ 1u);
      }
      bool x_issued = false;
      if constexpr (kXLds) {   // activations two k blocks ahead into the X buffer whose block was split last iteration
        if (kb + 2 < KB) { xissue(xa, ma, kb + 2, buf); x_issued = true; }
        else if (next_tile) { xissue(xn, mn, kb + 2 - KB, buf); x_issued = true; }
      }
      // vmcnt retires in order: "all but the MT*2 youngest" = everything except the activation pieces just issued
      auto xwait_older = [&]() {
        if (x_issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MT * 2) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      };
      if constexpr (kH0) {      // the block split during this one: this tile's next, or the next tile's first
        const bool wrap = kb + 1 >= KB;
        h_kb = wrap ? 0 : kb + 1;
        if (wrap) h_take(h_par ^ 1u);      // (the rows landed with the first k block's wait)
      }
      const unsigned wsel = WRES ? static_cast<unsigned>(kb) : buf;
      const unsigned wl_addr = lds0 + wsel * kBufBytes + (wsel >= 1u ? kExtBytes : 0u);
      u32x4 wh, wm, wl;
      lds_read<0>(wh, wl_addr); lds_read<1024>(wm, wl_addr); lds_read<2048>(wl, wl_addr);
      lds_wait(wh, wm, wl);
      static_for<0, NT>([&](auto I) {
        constexpr int nt = decltype(I)::value;
        u32x4 nh, nm, nl;
        if constexpr (nt + 1 < NT) {
          lds_read<(nt + 1) * 3072>(nh, wl_addr);
          if (!(p.dbg & 512)) {      // (ablation knob: one LDS read per step instead of three -- wrong results, timing only)
            lds_read<(nt + 1) * 3072 + 1024>(nm, wl_addr);
            lds_read<(nt + 1) * 3072 + 2048>(nl, wl_addr);
          } else {
            nm = nh; nl = nh;
          }
        }
        if constexpr (!kXLds && !kH0 && nt == kConvFirst + kConvSteps) {   // the raw registers are free again: activations two k blocks ahead
          if (kb + 2 < KB) xissue(xa, ma, kb + 2, 0u);
          else if (next_tile) xissue(xn, mn, kb + 2 - KB, 0u);
        }
        // ---- one scheduling region: 24 MFMAs + this step's share of the activation split (VALU) ----
        // (unconditional -- after the last block it splits stale data that nobody uses: a branch would put the
        // VALU work into its own basic block, where it cannot be interleaved with the MFMAs)
        // LDS path: the split runs in the LAST steps of the block, behind a counted wait -- the DMA of that data was
        // issued at the top of the previous block and has had 1 2/3 blocks to land
        if constexpr (kXLds && nt == kConvFirst) xwait_older();
        if constexpr (nt >= kConvFirst && nt < kConvFirst + kConvSteps) {
#pragma unroll
          for (int i = 0; i < kPerStep; ++i) {
            constexpr int base = (nt - kConvFirst) * kPerStep;
            if (base + i < kPairs) convert_pair(nxt, pair_mt(base + i), pair_pr(base + i), buf ^ 1u);
          }
        }
        // smallest products first; the MT accumulators of a product are independent
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = mfma_bf16(wh, cur.l[mt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = mfma_bf16(wl, cur.h[mt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = mfma_bf16(wm, cur.m[mt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = mfma_bf16(wh, cur.m[mt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = mfma_bf16(wm, cur.h[mt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = mfma_bf16(wh, cur.h[mt], acc[mt][nt]);
        if constexpr (nt >= kConvFirst && nt < kConvFirst + kConvSteps) {
          // a wave issues in order: the split only overlaps the matrix pipe if its VALU ops sit BETWEEN MFMAs
#pragma unroll
          for (int i = 0; i < 6 * MT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, kValuPerMfma, 0);
          }
        }
        // MFMAs are pure values to the compiler; pin them (and the split's results) to this step
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+a"(acc[mt][nt]));
        if constexpr (nt >= kConvFirst && nt < kConvFirst + kConvSteps) {
#pragma unroll
          for (int i = 0; i < kPerStep; ++i) {
            constexpr int base = (nt - kConvFirst) * kPerStep;
            if (base + i < kPairs) {
              const int mt = pair_mt(base + i), pr = pair_pr(base + i);
              asm volatile("" : "+v"(nxt.h[mt][pr]), "+v"(nxt.m[mt][pr]), "+v"(nxt.l[mt][pr]));
            }
          }
        }
        if constexpr (nt + 1 < NT) {
          lds_wait(nh, nm, nl);
          wh = nh; wm = nm; wl = nl;
        }
      });
      X3_T(t_k1);
      if constexpr (kXLds) xwait_older();   // next block's weights have landed (the activations after next may still fly)
      else xwait();                         // next block's weights (DMA) and the raw registers of the block after next
      X3_T(t_k2);
      // raw barrier: __syncthreads() carries a fence that drains vmcnt(0) whenever an LDS-DMA is pending -- exactly
      // the activation prefetch this loop wants to keep in flight across the barrier.  Every wave has waited for
      // its own share of the next weight block above, so after the barrier the whole block is in LDS.
      if constexpr (!WRES) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();    // every wave is done reading wbuf[buf]
      }
      asm volatile("" ::: "memory");
      X3_T(t_k3);
      X3_ADD(0, t_k0, t_k1); X3_ADD(1, t_k1, t_k2); X3_ADD(2, t_k2, t_k3);
      cur = nxt;
    }

    X3_T(t_e0);
    // epilogue: lane (li, g) of (mt, nt) holds Y[row0 + 16 mt + li][16 nt + 4 g .. + 3]
    if (row0 < p.M && !(p.dbg & 8)) {
      const bool full_rows = row0 + MT * 16 <= p.M;
      int li_e = li, g_e = g;
      asm volatile("" : "+v"(li_e), "+v"(g_e));
      // all bias fragments up front: one L2 round trip instead of one per n tile (each tile below is its own basic
      // block, so the loads would otherwise be waited for one by one -- that was ~half of the epilogue time)
      float4 bias4[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bias4[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias && (!kStage || (p.dbg & (8192 | 16384 | 32768)))) {      // (the staged form takes the bias from LDS)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int col = nt * 16 + 4 * g_e;
          bias4[nt] = *reinterpret_cast<const float4 *>(p.bias + (col + 4 <= N ? col : (N - 4)));
        }
      }
      // GATHER: the addend fragments of n tile nt (this lane's rows, its four columns)
      auto gload = [&](int nt, f32x4 *ga, f32x4 *gb) {
        if constexpr (GATHER) {
          int col = nt * 16 + 4 * g_e;
          col = col + 4 <= N ? col : (N - 4);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            if constexpr (GATHER == 1) ga[mt] = *reinterpret_cast<const f32x4 *>(p.gq + gqo[mt] + col);
            gb[mt] = *reinterpret_cast<const f32x4 *>(p.gk + gko[mt] + col);
          }
        }
      };
      auto otile = [&](int nt, auto has_omask, auto guarded, const f32x4 *ga = nullptr, const f32x4 *gb = nullptr) {
        const int col = nt * 16 + 4 * g_e;
        const bool cv = col + 4 <= N;
        const int colc = cv ? col : (N - 4);
        const float4 bv = bias4[nt];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const long long row = row0 + mt * 16 + li_e;
          const bool rv = !decltype(guarded)::value || row < p.M;
          const long long rowc = rv ? row : (p.M - 1);
          float4 v = make_float4(acc[mt][nt][0] + bv.x, acc[mt][nt][1] + bv.y, acc[mt][nt][2] + bv.z, acc[mt][nt][3] + bv.w);
          if constexpr (GATHER) {
            const f32x4 d = GATHER == 2 ? gb[mt] : ga[mt] - gb[mt];
            v.x += d[0]; v.y += d[1]; v.z += d[2]; v.w += d[3];
          }
          if (p.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          if (decltype(has_omask)::value) {
            const float4 om = *reinterpret_cast<const float4 *>(p.out_mask + rowc * N + colc);
            v.x = om.x > 0.f ? v.x : 0.f; v.y = om.y > 0.f ? v.y : 0.f; v.z = om.z > 0.f ? v.z : 0.f; v.w = om.w > 0.f ? v.w : 0.f;
          }
          if constexpr (decltype(has_omask)::value == 2) {      // (has_omask = 2: out_mask, then the skip-connection addend)
            const float4 ad = *reinterpret_cast<const float4 *>(p.addend + rowc * N + colc);
            v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
          }
          const f32x4 vv = {v.x, v.y, v.z, v.w};
          if (decltype(guarded)::value) {
            if (cv && rv) {
              if (p.dbg & 64) __builtin_nontemporal_store(vv, reinterpret_cast<f32x4 *>(p.Y + rowc * N + colc));
              else *reinterpret_cast<f32x4 *>(p.Y + rowc * N + colc) = vv;
            }
          } else {
            if (p.dbg & 64) __builtin_nontemporal_store(vv, reinterpret_cast<f32x4 *>(p.Y + row * N + col));
            else if (p.dbg & 2048)      // (ablation, timing only: one contiguous KiB per store instruction)
              *reinterpret_cast<f32x4 *>(p.Y + row0 * N + (((mt * NT + nt) * 256) % 6144) + (li_e + 16 * g_e) * 4) = vv;
            else *reinterpret_cast<f32x4 *>(p.Y + row * N + col) = vv;
          }
        }
      };
      // ---- staged form: accumulators -> LDS (row-major, 16 rows x tiles-of-this-pass) -> consecutive 16-byte chunks -> HBM ----
      auto staged = [&](auto has_omask) {
        const unsigned buf_last = (gs - 1u) & 1u;               // the weight buffer this tile's last k block read: free now
        const unsigned ebase = ldsw + (buf_last ? kBufBytes : 0u) + static_cast<unsigned>(wave) * (kPerWave * 16u);
        // (everything below derives from the opaque per-tile copies li_e / g_e: LICM would otherwise carry the chunk -> (row,
        // column) arithmetic of every read-back instruction across the whole k loop, in registers the 8-wave forms do not have)
        const unsigned lane_e = static_cast<unsigned>(li_e + 16 * g_e);
        const unsigned rd = ebase + lane_e * 16u;                                  // chunk i of a pass: + 1024 i
        float *ytile = p.Y + row0 * N;                                             // (wave-uniform: SGPR base of the stores)
        const float *mtile = decltype(has_omask)::value ? p.out_mask + row0 * N : nullptr;
        const float *atile = decltype(has_omask)::value == 2 ? p.addend + row0 * N : nullptr;
        const int rows_left = p.M - row0 < MT * 16 ? static_cast<int>(p.M - row0) : MT * 16;
        static_for<0, MT>([&](auto MI) {
          constexpr int mt = decltype(MI)::value;
          static_for<0, kPasses>([&](auto PI) {
            constexpr int pass = decltype(PI)::value;
            constexpr int nt0 = pass * kTpp;
            constexpr int ntn = (NT - nt0) < kTpp ? (NT - nt0) : kTpp;        // tiles of this pass = KiB staged per 16 rows
            constexpr unsigned pitch = ntn * 64u;                              // bytes per staged row
            const unsigned wr = ebase + static_cast<unsigned>(li_e) * pitch + static_cast<unsigned>(g_e) * 16u;
            const int valid = (N - nt0 * 16) * 4 < static_cast<int>(pitch) ? (N - nt0 * 16) * 4 : static_cast<int>(pitch);   // bytes of a staged row that exist
            // chunk i of the read-back: where it goes in the output, whether it exists -- and, with an output mask, the mask
            // chunk from the SAME offset of the mask tensor: loaded here in KiB-sized row runs, in flight during the LDS round
            // trip (the direct epilogue fetched it as 16 rows x 64 B per tile, one wait each)
            unsigned off[ntn];
            bool live[ntn];
            f32x4 om[decltype(has_omask)::value ? ntn : 1];
            f32x4 ad[decltype(has_omask)::value == 2 ? ntn : 1];
            // GATHER == 2 (one table, a row per output row): the addend in the READ-BACK layout -- row runs of the table, 2-3
            // segments per instruction (in the fragment layout, 16 rows x 64 B per instruction, this gather cost the 13-tile form
            // +330 us on 1.8 M rows, here +200).  The table row of chunk i's output row comes from the lane that owns that row in
            // the fragment layout (ds_bpermute).  GATHER == 1 (per-point queries: the q rows are broadcasts) stays in the fragment
            // layout below: measured 0 / +55 us on 320 000 x 256 x 256 against the read-back form.
            f32x4 gka[GATHER == 2 ? ntn : 1];
            static_for<0, ntn>([&](auto CI) {
              constexpr int i = decltype(CI)::value;
              const unsigned f = static_cast<unsigned>(i) * 1024u + lane_e * 16u;
              const unsigned r = f / pitch, cb = f - r * pitch;
              off[i] = ((static_cast<unsigned>(mt) * 16u + r) * static_cast<unsigned>(N) + static_cast<unsigned>(nt0) * 16u) * 4u + cb;
              live[i] = static_cast<int>(cb) < valid && static_cast<int>(mt * 16 + r) < rows_left;
              if constexpr (decltype(has_omask)::value) {
                om[i] = live[i] ? *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(mtile) + off[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
              }
              if constexpr (decltype(has_omask)::value == 2) {
                ad[i] = live[i] ? *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(atile) + off[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
              }
              if constexpr (GATHER == 2) {
                const unsigned cbytes = static_cast<unsigned>(nt0) * 64u + cb;       // byte offset of the chunk within a table row
                const unsigned ko = static_cast<unsigned>(__builtin_amdgcn_ds_bpermute(static_cast<int>(r * 4u), static_cast<int>(gko[mt])));
                gka[i] = live[i] ? *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(p.gk + ko) + cbytes) : f32x4{0.f, 0.f, 0.f, 0.f};
              }
            });
            f32x4 gav[GATHER == 1 ? ntn : 1], gbv[GATHER == 1 ? ntn : 1];      // two tables, fragment layout: in flight over the bias reads
            if constexpr (GATHER == 1) {
              static_for<0, ntn>([&](auto TI) {
                constexpr int t = decltype(TI)::value;
                int col = (nt0 + t) * 16 + 4 * g_e;
                col = col + 4 <= N ? col : (N - 4);
                gav[t] = *reinterpret_cast<const f32x4 *>(p.gq + gqo[mt] + col);
                gbv[t] = *reinterpret_cast<const f32x4 *>(p.gk + gko[mt] + col);
              });
            }
            f32x4 bvv[ntn];
            static_for<0, ntn>([&](auto TI) {
              constexpr int t = decltype(TI)::value;
              lds_read_f4<(nt0 + t) * 64>(bvv[t], ldsb + static_cast<unsigned>(16 * g_e));
            });
            static_for<0, ntn>([&](auto TI) {
              constexpr int t = decltype(TI)::value;
              constexpr int nt = nt0 + t;
              lgkm_wait<ntn - 1 - t>(bvv[t]);
              f32x4 v = acc[mt][nt] + bvv[t];
              if constexpr (GATHER == 1) v += gav[t] - gbv[t];
              if (p.relu_out) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
              }
              lds_write_f4<t * 64>(wr, v);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (one wave: its LDS operations complete in order)
            f32x4 back[ntn];
            static_for<0, ntn>([&](auto CI) {
              constexpr int i = decltype(CI)::value;
              lds_read_f4<i * 1024>(back[i], rd);
            });
            static_for<0, ntn>([&](auto CI) {
              constexpr int i = decltype(CI)::value;
              lgkm_wait<ntn - 1 - i>(back[i]);
              f32x4 v = back[i];
              if constexpr (decltype(has_omask)::value) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = om[i][c] > 0.f ? v[c] : 0.f;
              }
              if constexpr (decltype(has_omask)::value == 2) v += ad[i];
              if constexpr (GATHER == 2) v += gka[i];
              if (live[i]) store_f4(off[i], v, ytile);
            });
          });
        });
      };
      // ---- K = 4 tail: nothing is stored but one partial per wave tile.  In the accumulator layout lane (li, g) holds row li and the
      // columns 16 nt + 4 g .. + 3 of every n tile: d = y * [pre-activation of the K = 4 layer > 0] (recomputed from the lane's own
      // 16-byte input row, k4.h), the 16 products d * x4 and the 4 column sums are added up over the wave's row tiles in the lane,
      // then over the 16 lanes of the row group by four DPP steps (fixed order); lane li = 0 of each g writes 5 float4.
      auto direct_tail = [&]() __attribute__((always_inline)) {      // (as a CALL it would spill the accumulators)
        f32x4 xr[MT];
        bool rvm[MT];
        const unsigned x4a = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_ptr_t)(&t_x4lds[wave * 64]))) + static_cast<unsigned>(li_e) * 16u;
        const unsigned w0a = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_ptr_t)(&t_w0lds[0]))) + static_cast<unsigned>(g_e) * 64u;
        static_for<0, MT>([&](auto MI) {
          constexpr int mt = decltype(MI)::value;
          rvm[mt] = row0 + mt * 16 + li_e < p.M;
          lds_read_f4<mt * 256>(xr[mt], x4a);
        });
        float *wsp = p.t_ws + (tile * WV + wave) * static_cast<long long>(NT * 80) + g_e * 20;
        static_for<0, NT>([&](auto NI) {
          constexpr int nt = decltype(NI)::value;
          const int col = nt * 16 + 4 * g_e;
          const bool cv = col < N;                                      // (N % 4 == 0: a column quad exists or does not)
          f32x4 w0[4], b0;
          static_for<0, 4>([&](auto CI) {
            constexpr int c = decltype(CI)::value;
            lds_read_f4<(nt * 16 + c) * 16>(w0[c], w0a);      // rows 16 nt + 4 g + c (zeros beyond N)
          });
          lds_read_f4<nt * 64>(b0, ldsb + static_cast<unsigned>(16 * g_e));
          float ta[20];
#pragma unroll
          for (int v = 0; v < 20; ++v) ta[v] = 0.f;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const float4 xq = make_float4(xr[mt][0], xr[mt][1], xr[mt][2], xr[mt][3]);
            const bool ok = rvm[mt] && cv;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float pre = nsdp::k4_preact_n(xq, make_float4(w0[c][0], w0[c][1], w0[c][2], w0[c][3]), b0[c], c);      // (column 16 nt + 4 g + c)
              const float d = (ok && pre > 0.f) ? acc[mt][nt][c] : 0.f;
              ta[16 + c] += d;
#pragma unroll
              for (int k = 0; k < (TAIL == 2 ? 3 : 4); ++k) ta[4 * c + k] += d * xr[mt][k];
            }
          }
          row16_sum20<TAIL == 2>(ta);
          if (li_e == 0) {
#pragma unroll
            for (int j = 0; j < 5; ++j)
              *reinterpret_cast<f32x4 *>(wsp + nt * 80 + j * 4) = f32x4{ta[4 * j], ta[4 * j + 1], ta[4 * j + 2], ta[4 * j + 3]};
          }
        });
      };
      // ---- G16 output: the accumulator tile (mt, nt) -- lane (li, g): row li, columns 16 nt + 4 g .. + 3 -- IS the contiguous KiB
      // [quads 4 nt .. + 3][16 rows][4 floats] of the row group, in lane order li + 16 g: one store instruction per tile straight
      // from the registers, consecutive tiles consecutive KiB.  No out_mask / addend (host contract).
      auto g16_out = [&]() __attribute__((always_inline)) {      // (as a CALL it would spill the accumulators)
        float *ytile = p.Y + row0 * N + (li_e + 16 * g_e) * 4;
        // ReLU bits of the output (X3Params::bits_out): n tiles 2 j and 2 j + 1 are the two halves of k block j of the NEXT
        // kernel's fragment convention -- the lane owns both nibbles of byte (row group, j, li * 4 + g)
        unsigned char *btile = p.bits_out ? p.bits_out + (row0 >> 4) * (static_cast<long long>((N + 31) >> 5) * 64) + li_e * 4 + g_e : nullptr;
        static_for<0, MT>([&](auto MI) __attribute__((always_inline)) {
          constexpr int mt = decltype(MI)::value;
          if (row0 + mt * 16 < p.M) {                                // (M % 16 == 0: a row tile exists or does not)
            unsigned byte = 0;
            static_for<0, NT>([&](auto NI) __attribute__((always_inline)) {
              constexpr int nt = decltype(NI)::value;
              const float4 bv = bias4[nt];
              f32x4 v = {acc[mt][nt][0] + bv.x, acc[mt][nt][1] + bv.y, acc[mt][nt][2] + bv.z, acc[mt][nt][3] + bv.w};
              if (p.relu_out) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
              }
              const bool cv = nt * 16 + 4 * g_e + 4 <= N;
              if (cv) *reinterpret_cast<f32x4 *>(ytile + mt * 16 * N + nt * 256) = v;
              if (btile) {
                const unsigned nib = cv ? ((v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u)) : 0u;
                byte = (nt & 1) ? (byte | (nib << 4)) : nib;
                if ((nt & 1) || nt == NT - 1) {
                  if (nt * 16 < ((N + 31) >> 5) * 32) btile[mt * (((N + 31) >> 5) * 64) + (nt >> 1) * 64] = static_cast<unsigned char>(byte);
                }
              }
            });
          }
        });
      };
      auto epilogue = [&](auto has_omask) {
        const int full_tiles = N >> 4;  // tiles whose 16 columns are all valid
        if constexpr (TAIL) {
          direct_tail();
          return;
        }
        if constexpr (kYG) {
          g16_out();
          return;
        }
        if constexpr (kStage) {
          staged(has_omask);
          return;
        }
        if constexpr (GATHER) {      // the next n tile's addend loads fly while this one is stored
          f32x4 ga[2][MT], gb[2][MT];
          gload(0, ga[0], gb[0]);
          static_for<0, NT>([&](auto I) {
            constexpr int nt = decltype(I)::value;
            if (nt * 16 < N) {
              if constexpr (nt + 1 < NT) {
                if ((nt + 1) * 16 < N) gload(nt + 1, ga[(nt + 1) & 1], gb[(nt + 1) & 1]);
              }
              otile(nt, has_omask, std::true_type{}, ga[nt & 1], gb[nt & 1]);
            }
          });
          return;
        }
        if (full_rows) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if (nt < full_tiles) otile(nt, has_omask, std::false_type{});
            else if (nt * 16 < N) otile(nt, has_omask, std::true_type{});
          }
        } else {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            if (nt * 16 < N) otile(nt, has_omask, std::true_type{});
        }
      };
      if (p.out_mask) {
        if constexpr (kMasked) {
          if (p.addend) epilogue(std::integral_constant<int, 2>{});
          else epilogue(std::true_type{});
        } else {
          epilogue(std::true_type{});
        }
      } else {
        epilogue(std::false_type{});
      }
    }

    X3_T(t_e1);
    X3_ADD(3, t_e0, t_e1);
    if (!next_tile) break;
    if constexpr (kStage) {
      // the next tile's first k block stages weights into the buffer the epilogue pieces live in: every wave must have
      // read its piece back
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    tile += stride;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { xa[mt] = xn[mt]; ma[mt] = mn[mt]; }
    set_rows(tile + stride, xn, mn);
    if constexpr (kH0) {      // h_xs already holds the new tile's rows; the buffer of the tile just finished takes the one after
      h_rows(tile + stride, h_par);
      h_par ^= 1u;
    }
  }
#ifdef NSDP_X3_TIMING
  t_acc[5] = __builtin_readcyclecounter() - t_begin;
  if (lane == 0)
    for (int i = 0; i < 6; ++i) atomicAdd(&g_x3_timers[i], t_acc[i]);
#endif
}

template <int MT, int NT, int PRE, int WV, bool XREG = false, int KBM = 2, int GATHER = 0, int LAY = 0>
void launch_x3_pre(const X3Params &p, hipStream_t st, int wgs_per_cu = 1) {
  const long long rows_per_wg = static_cast<long long>(WV) * MT * 16;
  const long long wg_tiles = (p.M + rows_per_wg - 1) / rows_per_wg;
  // persistent workgroups, one per CU: the next tile's first k blocks are prefetched under the current tile's
  // last MFMAs and epilogue
  // (experiment knob, read once: NSDP_X3_RESERVE_CUS = compute units left free for the other stream's kernels)
  static const int reserve_env = getenv("NSDP_X3_RESERVE_CUS") ? atoi(getenv("NSDP_X3_RESERVE_CUS")) : 0;
  // (host hint 9, set around launches that run on a side stream beside the critical chain: see nsdp_debug_set)
  const int reserve = g_x3_side_reserve > 0 && g_x3_side_reserve < nsdp::num_cus() ? g_x3_side_reserve : reserve_env;
  const long long slots = static_cast<long long>(nsdp::num_cus() - reserve) * wgs_per_cu;
  const unsigned grid = static_cast<unsigned>(wg_tiles < slots ? wg_tiles : slots);
  NSDP_TRACE("linear_bf16x3<%d,%d,%d,%d,%d>x%d%s%s%s", MT, NT, PRE, WV, static_cast<int>(XREG), wgs_per_cu, KBM > 2 ? " wres" : "",
             GATHER == 2 ? " gather1" : GATHER ? " gather" : "",
             LAY == 0 ? "" : LAY == 1 ? " g16:x" : LAY == 2 ? " g16:y" : LAY == 3 ? " g16:xy" : LAY == 4 ? " g16:r" : LAY == 5 ? " g16:xr" : LAY == 6 ? " g16:yr" : " g16:xyr");
  hipLaunchKernelGGL((linear_bf16x3_kernel<MT, NT, PRE, WV, XREG, KBM, GATHER, 0, LAY>), dim3(grid), dim3(WV * 64), 0, st, p);
}

}  // namespace
