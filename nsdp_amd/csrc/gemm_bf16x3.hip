// fp32 dense layers on the CDNA4 bf16 matrix pipe by error-compensated splitting ("bf16x3"):
//
//     x = h + m + l,   h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)        (round to nearest, residuals exact)
//     x * w  ~=  h*wh + h*wm + m*wh + m*wm + h*wl + l*wh                          (6 of the 9 products)
//
// The dropped products (m*wl, l*wm, l*wl) are <= 2^-23 |x||w|, the split itself loses <= 2^-24 |x| -- the result
// is accurate to fp32 rounding level (tests: error vs. an fp64 reference is the same as the exact-fp32 MFMA
// kernel's), products are exact in the matrix pipe and accumulate in fp32.  v_mfma_f32_16x16x32_bf16 has 16x
// the rate of v_mfma_f32_16x16x4_f32, so 6 products cost 6/16 of the exact path: a 2.67x higher compute
// roofline (2.5 PF / 6 = 417 TFLOP/s of fp32-equivalent work), which moves the tall-skinny layers of the
// TDNet path (M ~ 10^6, N, K <= 256) from the MFMA bound to the HBM bound.
//
// nsdp_linear_bf16x3_f32: Y[M,N] = post( pre(X)[M,K] W[N,K]^T + b ) (+ residual), same contract as nsdp_linear_f32.
//   * one wave owns 64 rows (MT = 4 row tiles) and all N columns: accumulators 4 x NT x 4 <= 256 AGPRs.
//   * W comes pre-split (nsdp_pack_weight_bf16x3: [k block of 32][n tile][plane h,m,l][lane][8 bf16] = 1 KiB
//     per wave-wide operand), is DMA'd global -> LDS once per workgroup and k block (global_load_lds, two
//     buffers, one barrier per k block) and read back with conflict-free ds_read_b128: per-wave register
//     loads of W would need 62 B/clk/CU of L1 bandwidth at this MFMA rate.
//   * X is read once from HBM, 32 B per lane and k block (k-permuted fragment convention: lane group g
//     supplies k = 32 kb + {4 g .. + 3, 16 + 4 g .. + 3} to A and B alike, so row-major rows need no transposition
//     and the four lane groups of a row read 64 contiguous bytes per load instruction), one
//     k block ahead, and split on the VALU (v_cvt_pk_bf16_f32 / v_pk_add_f32: 4.5 ops per value) during
//     the first MFMA steps of the previous k block.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "k4.h"
#include "pack_bodies.h"
#include "prof.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

int g_x3_dbg = 0;
thread_local int g_x3_side_reserve = 0;      // (per host thread) compute units a side-stream launch leaves free (host hint 9: it runs on the weight-gradient side stream)

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
};

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
template <int MT, int NT, int PRE, int WV, bool XREG = false, int KBM = 2, int GATHER = 0, int TAIL = 0>
__global__ __launch_bounds__(WV * 64, XREG ? 2 : 1) void linear_bf16x3_kernel(X3Params p) {
  static_assert(!GATHER || PRE == 0 || PRE == 3, "the gathered addend belongs to the plain-prologue forms");
  static_assert(!TAIL || (PRE == 0 && !GATHER), "the K = 4 tail belongs to the plain-prologue forms");
  constexpr bool WRES = KBM > 2;
  // H0: the operand is recomputed from 16-byte coordinate rows (see X3Params::h_x4): no activation DMA, no staging, no raw registers
  constexpr bool kH0 = PRE == 3;
  // PRE != 1: the raw fp32 activations go global -> LDS by DMA as well (wave-private 8 KiB pieces, two k blocks
  // deep): no registers in flight, issued a whole k block earlier.  PRE == 1 (activation + mask) would not fit
  // in LDS next to the weights and keeps the register path.
  constexpr bool kXLds = PRE != 1 && !XREG && !kH0;
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
  constexpr bool kStage = !WRES && kExtFree >= 0 && !TAIL;                   // (resident weights fill the LDS: direct epilogue; the K = 4 tail stores no tile)
  constexpr int kExtWant = WV * NT * 64 - kWTile;                            // whole 16-row tiles for every wave
  constexpr int kExt = !kStage ? 0 : (kExtWant < 0 ? 0 : (kExtFree < 0 ? 0 : (kExtWant < kExtFree ? kExtWant : kExtFree)));
  constexpr int kPerWave = (kWTile + kExt) / WV;                             // u32x4 of epilogue staging per wave
  constexpr int kTppMax = kPerWave / 64 < NT ? kPerWave / 64 : NT;           // 16-column tiles per pass (1 KiB per tile)
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
  const float *ma[MT], *mn[MT];
  auto set_rows = [&](long long t, const float **x, const float **m) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      long long r = (t * WV + wave) * (MT * 16) + mt * 16 + li;
      r = r < p.M ? r : (p.M - 1);
      x[mt] = p.X + r * K;
      m[mt] = PRE == 1 ? p.mask + r * K : nullptr;
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
  auto xissue = [&](const float *const *x, const float *const *m, int kb, unsigned xb) {
    if constexpr (kH0) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        int ko = kb * 32 + 16 * hf + 4 * g;     // the four lane groups of a row read 64 contiguous bytes per instruction
        ko = ko < K ? ko : (K - 4);     // past the row end: re-read in-row data (the packed weights are zero there)
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
  constexpr int kValuPerMfma = (kPerStep * (PRE == 1 ? 13 : PRE == 2 ? 10 : PRE == 3 ? 21 : 9) + 6 * MT - 1) / (6 * MT);

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
          const float4 v = *reinterpret_cast<const float4 *>(p.residual + row * N + col);
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
      auto epilogue = [&](auto has_omask) {
        const int full_tiles = N >> 4;  // tiles whose 16 columns are all valid
        if constexpr (TAIL) {
          direct_tail();
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
        if constexpr (PRE == 1) {
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

// K = 4 tail, reduction of the per-wave-tile partials [n tile][g][20] (16 products c * 4 + k, then 4 column sums).  Stage 1:
// workgroup b sums the tiles b, b + grid, ... float4-wise (the layout is the same for every tile) -> part[b]; fixed order.
__global__ __launch_bounds__(512) void x3_tail_sum_tiles_kernel(const f32x4 *__restrict__ ws, long long tiles, int quads,
                                                                f32x4 *__restrict__ part) {
  const int q = threadIdx.x;
  if (q >= quads) return;
  f32x4 a[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  long long t = blockIdx.x;
  const long long g = gridDim.x;
  for (; t + 3 * g < tiles; t += 4 * g) {
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] += ws[(t + u * g) * quads + q];
  }
  for (; t < tiles; t += g) a[0] += ws[t * quads + q];
  part[static_cast<long long>(blockIdx.x) * quads + q] = (a[0] + a[1]) + (a[2] + a[3]);
}
// Stage 2: partials [S][n tiles][4][20] -> dW [N][k_out] (the layer's own input width: 3 or 4), db [N]
__global__ __launch_bounds__(256) void x3_tail_reduce_kernel(const float *__restrict__ part, int S, int N, int ntiles, int k_out,
                                                             float *__restrict__ dW, float *__restrict__ db, int accumulate) {
  const int e = blockIdx.x * 256 + threadIdx.x;      // e < 4 N: dW[n][k], n = e / 4; else db[e - 4 N]
  if (e >= 5 * N) return;
  const int n = e < 4 * N ? (e >> 2) : (e - 4 * N);
  const int v = e < 4 * N ? 4 * (n & 3) + (e & 3) : 16 + (n & 3);
  const long long stride = static_cast<long long>(ntiles) * 80;
  const float *src = part + (n >> 2) * 20 + v;      // ((n / 16) * 4 + (n % 16) / 4) * 20
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // 8 loads in flight per lane
  int i = 0;
  for (; i + 8 <= S; i += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] += src[(i + u) * stride];
  }
  for (; i < S; ++i) acc[0] += src[i * stride];
  const float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  if (e < 4 * N) {
    const int k = e & 3;
    if (k < k_out) dW[n * k_out + k] = accumulate ? dW[n * k_out + k] + s : s;
  } else if (db) {
    db[n] = accumulate ? db[n] + s : s;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// "Anti-phase" form (experiment, nsdp_debug_set(6, 128)): ONE 8-wave workgroup per CU whose two 4-wave groups run
// half a tile period apart, so that the epilogue (stores) and tile prologue of one group sit under the MFMA steps of
// the other -- while both still share ONE weight stream: the workgroup cycles through the k blocks 0 .. KB-1 forever,
// one block per barrier-delimited slot, and a group that starts its tile at slot s consumes the blocks in the rotated
// order s mod KB, s+1 mod KB, ... (a dot product does not care).  A tile takes KB compute slots + E slots in which the
// group only stores / idles (E = 1 or 2, same parity as KB, so that the half period is a whole number of slots).
// The stores are posted and never waited for by the group that issued them until its next counted vmcnt wait (gfx9 retires
// vector memory operations in order): weight pieces are staged by the group that is past the first slot of its tile.
template <int NT, int PRE>
__global__ __launch_bounds__(512, 1) void linear_bf16x3_ap_kernel(X3Params p, int E) {
  static_assert(PRE != 1, "the masked prologue keeps the register path");
  constexpr int MT = 2, WV = 8;
  __shared__ __attribute__((aligned(16))) u32x4 wbuf[2][NT * 3 * 64];
  __shared__ __attribute__((aligned(16))) u32x4 xbuf[2][WV][MT * 2 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2, wq = wave & 3;
  const int li = lane & 15, g = lane >> 4;
  const int K = p.K, N = p.N;
  const int KB = (K + 31) >> 5;
  const int ntiles = (N + 15) >> 4;
  const int P = KB + E, D = grp ? (P >> 1) : 0, Do = grp ? 0 : (P >> 1);
  constexpr long long kRowsGt = 4 * MT * 16;                    // rows of one group tile
  const long long gtiles = (p.M + kRowsGt - 1) / kRowsGt;
  const long long gstride = 2LL * gridDim.x;
  auto count = [&](long long first) -> int { return first < gtiles ? static_cast<int>((gtiles - first + gstride - 1) / gstride) : 0; };
  const int ntl0 = count(2LL * blockIdx.x);
  const int ntl = count(2LL * blockIdx.x + grp), ntlo = count(2LL * blockIdx.x + (grp ^ 1));
  const int S_total = (P >> 1) + ntl0 * P;                      // every wave runs exactly this many slots (barriers)
  long long gt = 2LL * blockIdx.x + grp;

  // activation addresses = wave-uniform tile base (SGPRs) + a 32-bit lane offset: row (clamped into the tensor: rows >= M
  // are computed and never stored) times the row pitch
  auto tile_base = [&](long long t) -> const char * {
    return reinterpret_cast<const char *>(p.X + (t < gtiles ? t : gtiles - 1) * kRowsGt * K);
  };
  auto tile_rows = [&](long long t) -> int {
    const long long left = p.M - (t < gtiles ? t : gtiles - 1) * kRowsGt;
    return left < kRowsGt ? static_cast<int>(left) : static_cast<int>(kRowsGt);
  };

  const char *wbase = static_cast<const char *>(p.Wp);      // uniform base + 32-bit lane offset: no per-lane pointer to keep
  const unsigned lane16 = lane * 16u;
  const int pieces = ntiles * 3;
  auto stage = [&](int kb, unsigned buf, int first, int step) {
    for (int q = first; q < pieces; q += step)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wbase + ((static_cast<long long>(kb) * pieces + q) << 10) + lane16),
                                       (lds_ptr_t)(&wbuf[buf][q * 64]), 16, 0, 0);
  };
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_ptr_t)(&wbuf[0][lane])));
  constexpr unsigned kBufBytes = NT * 3 * 1024;

  auto xissue = [&](const char *base, int rows, int kb, unsigned xb) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      int r = wq * (MT * 16) + mt * 16 + li;
      r = r < rows ? r : rows - 1;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        int ko = kb * 32 + 16 * hf + 4 * g;
        ko = ko < K ? ko : (K - 4);
        const unsigned off = static_cast<unsigned>(r * K + ko) * 4u;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + off), (lds_ptr_t)(&xbuf[xb][wave][(mt * 2 + hf) * 64]), 16, 0, 0);
      }
    }
  };
  struct Planes {
    u32x4 h[MT], m[MT], l[MT];
  };
  auto convert_pair = [&](Planes &pl, int mt, int pr, unsigned xb) {
    f32x4 v = __builtin_bit_cast(f32x4, xbuf[xb][wave][(mt * 2 + (pr >> 1)) * 64 + lane]);
    if (PRE == 2) {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = __builtin_amdgcn_fmed3f(v[c], 0.f, __builtin_inff());
    }
    unsigned h, m, l;
    split_pair(v[2 * (pr & 1)], v[2 * (pr & 1) + 1], h, m, l);
    pl.h[mt][pr] = h; pl.m[mt][pr] = m; pl.l[mt][pr] = l;
  };
  auto wrap = [&](int v) { return v >= KB ? v - KB : v; };

  // global slot state (identical in every wave) and the other group's position in its period
  int s = 0;
  int kwn = KB > 1 ? 1 : 0;      // (s + 1) mod KB: the weight block staged during slot s
  int ol = -Do;                  // other group's local slot index (negative: not started)
  int oi = 0;                    // ... modulo P once started
  auto other_comp_i = [&]() -> int { return (ol >= 0 && ol < ntlo * P && oi < KB) ? oi : -1; };
  auto advance = [&]() {
    ++s;
    if (++kwn == KB) kwn = 0;
    ++ol;
    if (ol > 0) { ++oi; if (oi == P) oi = 0; }
  };
  // a slot in which this group does not compute: `staging` only when no group computes at all (then all 8 waves stage)
  auto passive_slot = [&]() {
    const bool other_computes = other_comp_i() >= 0;
    if (!other_computes && s + 1 < S_total && !(p.dbg & 1)) {
      stage(kwn, (s + 1) & 1u, wave, WV);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    advance();
  };

  // prologue: weight block 0 by everybody; this group's first tile: blocks ks, ks+1 in flight, the first one split
  Planes cur, nxt;
  stage(0, 0u, wave, WV);
  int ks = D % KB;               // rotated start of the tile
  const char *xa = tile_base(gt), *xn = tile_base(gt + gstride);
  int ra = tile_rows(gt), rn = tile_rows(gt + gstride);
  if (ntl > 0) {
    xissue(xa, ra, ks, 0u);
    xissue(xa, ra, wrap(ks + 1), 1u);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (ntl > 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) convert_pair(cur, mt, pr, 0u);
  }
  __syncthreads();
  if (blockIdx.x & 7) {           // stagger the workgroups so that their store bursts do not coincide
    for (int i = 0; i < static_cast<int>(blockIdx.x & 7) * KB; ++i) __builtin_amdgcn_s_sleep(10);
  }
  for (int i = 0; i < D; ++i) passive_slot();

  constexpr int kConvSteps = NT > 4 ? 4 : NT - 1;
  constexpr int kConvFirst = NT - kConvSteps;
  constexpr int kPairs = MT * 4;
  constexpr int kPerStep = (kPairs + kConvSteps - 1) / kConvSteps;
  constexpr int kValuPerMfma = (kPerStep * (PRE == 2 ? 10 : 9) + 6 * MT - 1) / (6 * MT);
  unsigned cs = 0;               // this wave's compute-slot count: X buffer parity

  for (int n = 0; n < ntl; ++n) {
    const long long row0 = (gt * 4 + wq) * (MT * 16);
    const bool next_tile = n + 1 < ntl;
    const int ks_next = (ks + P) % KB;
    int li_t = li, g_t = g;
    asm volatile("" : "+v"(li_t), "+v"(g_t));
    f32x4 acc[MT][NT];
    if (p.residual) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        long long row = row0 + mt * 16 + li_t;
        row = row < p.M ? row : (p.M - 1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          int col = nt * 16 + 4 * g_t;
          col = col + 4 <= N ? col : (N - 4);
          const float4 v = *reinterpret_cast<const float4 *>(p.residual + row * N + col);
          acc[mt][nt] = f32x4{v.x, v.y, v.z, v.w};
        }
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int i = 0; i < KB; ++i, ++cs) {
      const unsigned wb = s & 1u, xb = cs & 1u;
      // who stages the next weight block: the group(s) past the first slot of their tile; the first-slot group only if alone
      const int oc = other_comp_i();
      const bool me_stage = i >= 1 || oc < 1;
      const bool other_stage = oc >= 1 || (oc == 0 && i < 1);
      if (me_stage && s + 1 < S_total && !(p.dbg & 1)) {
        if (other_stage) stage(kwn, wb ^ 1u, wave, WV);
        else stage(kwn, wb ^ 1u, wq, 4);
      }
      bool x_issued = false;
      if (i + 2 < KB) { xissue(xa, ra, wrap(ks + i + 2), xb); x_issued = true; }
      else if (next_tile) { xissue(xn, rn, wrap(ks_next + (i + 2 - KB)), xb); x_issued = true; }
      auto xwait_older = [&]() {
        if (x_issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MT * 2) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      };
      const unsigned wl_addr = lds0 + wb * kBufBytes;
      u32x4 wh, wm, wl;
      lds_read<0>(wh, wl_addr); lds_read<1024>(wm, wl_addr); lds_read<2048>(wl, wl_addr);
      lds_wait(wh, wm, wl);
      static_for<0, NT>([&](auto I) {
        constexpr int nt = decltype(I)::value;
        u32x4 nh, nm, nl;
        if constexpr (nt + 1 < NT) {
          lds_read<(nt + 1) * 3072>(nh, wl_addr); lds_read<(nt + 1) * 3072 + 1024>(nm, wl_addr);
          lds_read<(nt + 1) * 3072 + 2048>(nl, wl_addr);
        }
        if constexpr (nt == kConvFirst) xwait_older();
        if constexpr (nt >= kConvFirst && nt < kConvFirst + kConvSteps) {
#pragma unroll
          for (int q = 0; q < kPerStep; ++q) {
            constexpr int base = (nt - kConvFirst) * kPerStep;
            if (base + q < kPairs) convert_pair(nxt, (base + q) >> 2, (base + q) & 3, xb ^ 1u);
          }
        }
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
#pragma unroll
          for (int q = 0; q < 6 * MT; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, kValuPerMfma, 0);
          }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+a"(acc[mt][nt]));
        if constexpr (nt >= kConvFirst && nt < kConvFirst + kConvSteps) {
#pragma unroll
          for (int q = 0; q < kPerStep; ++q) {
            constexpr int base = (nt - kConvFirst) * kPerStep;
            if (base + q < kPairs) {
              const int mt = (base + q) >> 2, pr = (base + q) & 3;
              asm volatile("" : "+v"(nxt.h[mt][pr]), "+v"(nxt.m[mt][pr]), "+v"(nxt.l[mt][pr]));
            }
          }
        }
        if constexpr (nt + 1 < NT) {
          lds_wait(nh, nm, nl);
          wh = nh; wm = nm; wl = nl;
        }
      });
      xwait_older();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      cur = nxt;
      advance();
    }

    // epilogue: posted stores, nothing waits for them here
    if (row0 < p.M && !(p.dbg & 8)) {
      const bool full_rows = row0 + MT * 16 <= p.M;
      int li_e = li, g_e = g;
      asm volatile("" : "+v"(li_e), "+v"(g_e));
      float4 bias4[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bias4[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int col = nt * 16 + 4 * g_e;
          bias4[nt] = *reinterpret_cast<const float4 *>(p.bias + (col + 4 <= N ? col : (N - 4)));
        }
      }
      auto otile = [&](int nt, auto has_omask, auto guarded) {
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
          if (p.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          if (decltype(has_omask)::value) {
            const float4 om = *reinterpret_cast<const float4 *>(p.out_mask + rowc * N + colc);
            v.x = om.x > 0.f ? v.x : 0.f; v.y = om.y > 0.f ? v.y : 0.f; v.z = om.z > 0.f ? v.z : 0.f; v.w = om.w > 0.f ? v.w : 0.f;
          }
          const f32x4 vv = {v.x, v.y, v.z, v.w};
          if (decltype(guarded)::value) {
            if (cv && rv) *reinterpret_cast<f32x4 *>(p.Y + rowc * N + colc) = vv;
          } else {
            *reinterpret_cast<f32x4 *>(p.Y + row * N + col) = vv;
          }
        }
      };
      auto epilogue = [&](auto has_omask) {
        const int full_tiles = N >> 4;
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
      if (p.out_mask) epilogue(std::true_type{});
      else epilogue(std::false_type{});
    }
    for (int e = 0; e < E; ++e) passive_slot();
    gt += gstride;
    ks = ks_next;
    xa = xn; ra = rn;
    xn = tile_base(gt + gstride); rn = tile_rows(gt + gstride);
  }
  while (s < S_total) passive_slot();
}

template <int NT, int PRE>
void launch_x3_ap(const X3Params &p, hipStream_t st) {
  const int KB = (p.K + 31) >> 5;
  int E = (KB & 1) ? 1 : 2;
  if ((p.dbg >> 8) & 3) E = ((p.dbg >> 8) & 3) + ((((p.dbg >> 8) & 3) ^ KB) & 1);       // override, parity fixed up
  const long long gtiles = (p.M + 127) / 128;
  const long long wgs = (gtiles + 1) / 2;
  const unsigned grid = static_cast<unsigned>(wgs < nsdp::num_cus() ? wgs : nsdp::num_cus());
  NSDP_TRACE("linear_bf16x3_ap<%d,%d> E=%d", NT, PRE, E);
  hipLaunchKernelGGL((linear_bf16x3_ap_kernel<NT, PRE>), dim3(grid), dim3(512), 0, st, p, E);
}

// bf16x3 packs.  Wp  [ceil(K/32)][ceil(N/16)][3 planes][64 lanes][8 bf16]:
//   element j of lane 16 g + li of (kb, tn) = plane_p( W[16 tn + li][32 kb + kperm(g, j)] ),
//   kperm(g, j) = 16 (j / 4) + 4 g + j % 4: the k permutation of the activation loads (above)
// WpT [ceil(N/32)][ceil(K/16)][3][64][8]: the pack of W^T (operand of dX = dY W):
//   element j of lane 16 g + li of (nb, tk) = plane_p( W[32 nb + kperm(g, j)][16 tk + li] )
// zero outside [N,K].  One thread per (block, tile, lane) of each output.
__global__ __launch_bounds__(256) void pack_bf16x3_kernel(const float *__restrict__ W, int N, int K,
                                                          u32x4 *__restrict__ Wp, u32x4 *__restrict__ WpT) {
  nsdp::pack::x3_body(W, N, K, Wp, WpT, static_cast<long long>(blockIdx.x) * 256 + threadIdx.x);
}

template <int MT, int NT, int PRE, int WV, bool XREG = false, int KBM = 2, int GATHER = 0>
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
  NSDP_TRACE("linear_bf16x3<%d,%d,%d,%d,%d>x%d%s%s", MT, NT, PRE, WV, static_cast<int>(XREG), wgs_per_cu, KBM > 2 ? " wres" : "",
             GATHER == 2 ? " gather1" : GATHER ? " gather" : "");
  hipLaunchKernelGGL((linear_bf16x3_kernel<MT, NT, PRE, WV, XREG, KBM, GATHER>), dim3(grid), dim3(WV * 64), 0, st, p);
}

// (the hand-issued loads of this file must never be spilled while in flight: every variant is built spill-free)
template <int NT, int GATHER = 0>
int launch_x3(const X3Params &p, hipStream_t st) {
  const int pre = p.mask ? 1 : (p.relu_in ? 2 : 0);
  nsdp::prof::Scope scope(nsdp::prof::kLinearX3, st, 2.0 * p.M * p.N * p.K,
                          4.0 * (static_cast<double>(p.M) * (p.K + p.N) + static_cast<double>(p.N) * p.K));
  // measured per class: up to 13 n tiles two waves per SIMD with 2 row tiles each win (1.36 -> 1.20 ms on the
  // 1.8 M x 200 x 200 layers), at 16 n tiles one wave per SIMD with 3; the masked prologue keeps its raw activation
  // and mask registers in flight and uses the one-wave, spill-free variants throughout.
  // Up to 8 n tiles the two waves per SIMD come from TWO 4-wave workgroups per CU (2 x 80 KiB of LDS, exactly the
  // CU's 160 KiB): they share no barrier, so one's epilogue stores overlap the other's MFMA steps (2-10 % faster than
  // one 8-wave workgroup, bit-identical results).  13 n tiles would need 2 x 110 KiB.
  constexpr int MT1 = NT >= 16 ? 2 : NT >= 13 ? 3 : 4;
  const bool two_waves = NT <= 13 && !(g_x3_dbg & 32);
  // (the masked prologue as well, up to 8 n tiles: 10-25 % over one 4-wave workgroup with more row tiles)
  // N, K <= 128 (<= 8 n tiles, <= 4 k blocks), unmasked: the weight planes stay resident in LDS for the workgroup's lifetime
  // (96 KiB + 64 KiB of activation staging = the CU's 160 KiB): one 8-wave workgroup per CU, no barrier in the k loop.
  // nsdp_debug_set(6, 256) switches back to the streaming two-workgroup form (A/B).
  if (pre != 1 && NT <= 8 && p.K <= 128 && !(g_x3_dbg & 256)) {
    if (pre == 0) launch_x3_pre<2, (NT <= 8 ? NT : 8), 0, 8, false, 4, GATHER>(p, st);
    else launch_x3_pre<2, (NT <= 8 ? NT : 8), 2, 8, false, 4>(p, st);
  } else if (pre == 1 && NT <= 8 && two_waves) launch_x3_pre<2, (NT <= 8 ? NT : 8), 1, 4, true>(p, st, 2);
  else if (pre == 1) launch_x3_pre<MT1, NT, 1, 4>(p, st);
  else if (NT <= 8 && two_waves) {
    if (pre == 0) launch_x3_pre<2, (NT <= 8 ? NT : 8), 0, 4, false, 2, GATHER>(p, st, 2);
    else launch_x3_pre<2, (NT <= 8 ? NT : 8), 2, 4>(p, st, 2);
  } else if (NT == 13 && two_waves && p.M <= (1 << 19)) {
    // 13 n tiles, up to ~0.5 M rows: the same two-workgroups-per-CU form, made to fit (2 x 78 KiB) by taking the raw
    // activations through registers instead of the 32 KiB LDS staging: 8-19 % faster there, on par at 1.8 M rows
    if (pre == 0) launch_x3_pre<2, 13, 0, 4, true, 2, GATHER>(p, st, 2);
    else launch_x3_pre<2, 13, 2, 4, true>(p, st, 2);
  } else if (two_waves) {
    if ((g_x3_dbg & 128) && !GATHER && p.res_sign == 1.f && !p.addend) {
      if (pre == 0) launch_x3_ap<13, 0>(p, st);
      else launch_x3_ap<13, 2>(p, st);
    } else if (pre == 0) launch_x3_pre<2, (NT > 8 ? NT : 13), 0, 8, false, 2, GATHER>(p, st);
    else launch_x3_pre<2, (NT > 8 ? NT : 13), 2, 8>(p, st);
  } else {
    constexpr int MT0 = NT >= 16 ? 3 : 4;
    // Tile quantisation (16 n tiles): one persistent workgroup per CU walks row blocks of 4 waves x MT0 x 16 = 192 rows;
    // 51 200 rows (the 100-anchor attention blocks of a 32-shape batch) are 267 blocks on 256 CUs -- two rounds, the
    // second one 4 % full.  Two row tiles per wave (128-row blocks: 400 blocks, two shorter rounds) win whenever
    // rounds x rows per block is smaller by more than what the narrower wave tile costs per row (~8 %).
    bool narrow = false;
    if constexpr (NT >= 16) {
      const long long cus = nsdp::num_cus();
      const long long r3 = ((p.M + 191) / 192 + cus - 1) / cus * 192, r2 = ((p.M + 127) / 128 + cus - 1) / cus * 128;
      narrow = !(g_x3_dbg & 4096) && r2 * 108 < r3 * 100;
    }
    if (narrow) {
      if (pre == 0) launch_x3_pre<2, NT, 0, 4, false, 2, GATHER>(p, st);
      else launch_x3_pre<2, NT, 2, 4>(p, st);
    } else if (pre == 0) launch_x3_pre<MT0, NT, 0, 4, false, 2, GATHER>(p, st);
    else launch_x3_pre<MT0, NT, 2, 4>(p, st);
  }
  return nsdp::launch_status("linear_bf16x3_kernel");
}

// H0 forms (the hidden layer of a position-encoding MLP recomputed in the operand producer): the plain-prologue classes of
// launch_x3 without their activation staging.  13 n tiles always take the 8-wave form -- two 4-wave workgroups per CU have no
// room for the K = 4 layer's table next to 2 x 78 KiB of weight planes.
template <int NT, int GATHER>
int launch_x3_h0(const X3Params &p, hipStream_t st) {
  static_assert(NT == 8 || NT == 13 || NT == 16, "H0 forms: 8, 13 or 16 n tiles");
  nsdp::prof::Scope scope(nsdp::prof::kLinearX3, st, 2.0 * p.M * p.N * p.K,
                          4.0 * (static_cast<double>(p.M) * (4 + p.N) + static_cast<double>(p.N) * p.K));
  if constexpr (NT == 8) {
    launch_x3_pre<2, 8, 3, 8, false, 4, GATHER>(p, st);      // (K <= 128: resident weight planes)
  } else if constexpr (NT == 13) {
    launch_x3_pre<2, 13, 3, 8, false, 2, GATHER>(p, st);
  } else {
    const long long cus = nsdp::num_cus();
    const long long r3 = ((p.M + 191) / 192 + cus - 1) / cus * 192, r2 = ((p.M + 127) / 128 + cus - 1) / cus * 128;
    if (!(g_x3_dbg & 4096) && r2 * 108 < r3 * 100) launch_x3_pre<2, 16, 3, 4, false, 2, GATHER>(p, st);
    else launch_x3_pre<3, 16, 3, 4, false, 2, GATHER>(p, st);
  }
  return nsdp::launch_status("linear_bf16x3_kernel (h0)");
}

// K = 4 tail: the forms the position-encoding MLPs of the TDNet step take at scale (their dX GEMMs have square weights)
constexpr int kTailGrid = 256;      // stage-1 partials
template <int MT, int NT, int WV, int KBM = 2>
struct TailForm {
  static constexpr int kTailFloats = NT * 80;      // a wave tile's partial: [n tile][g][20]
  static long long wave_tiles(long long M) { return (M + MT * 16 - 1) / (MT * 16); }
  static size_t ws_floats(long long M) { return static_cast<size_t>(wave_tiles(M) + kTailGrid) * kTailFloats; }
  static void launch(X3Params p, int k_out, float *dW0, float *db0, int accumulate, hipStream_t st) {
    const long long rows_per_wg = static_cast<long long>(WV) * MT * 16;
    const long long wg_tiles = (p.M + rows_per_wg - 1) / rows_per_wg;
    const long long slots = nsdp::num_cus() - (g_x3_side_reserve > 0 && g_x3_side_reserve < nsdp::num_cus() ? g_x3_side_reserve : 0);
    const unsigned grid = static_cast<unsigned>(wg_tiles < slots ? wg_tiles : slots);
    NSDP_TRACE("linear_bf16x3<%d,%d,0,%d,0> k4tail", MT, NT, WV);
    // (k_out = 3: the fourth input column is zero padding -- its products stay out of the reduction)
    if (k_out == 3) hipLaunchKernelGGL((linear_bf16x3_kernel<MT, NT, 0, WV, false, KBM, 0, 2>), dim3(grid), dim3(WV * 64), 0, st, p);
    else hipLaunchKernelGGL((linear_bf16x3_kernel<MT, NT, 0, WV, false, KBM, 0, 1>), dim3(grid), dim3(WV * 64), 0, st, p);
    const long long tiles = wave_tiles(p.M);
    const int quads = kTailFloats / 4;
    const int S = static_cast<int>(tiles < kTailGrid ? tiles : kTailGrid);
    float *part = p.t_ws + tiles * kTailFloats;
    hipLaunchKernelGGL(x3_tail_sum_tiles_kernel, dim3(S), dim3(512), 0, st, reinterpret_cast<const f32x4 *>(p.t_ws), tiles, quads,
                       reinterpret_cast<f32x4 *>(part));
    hipLaunchKernelGGL(x3_tail_reduce_kernel, dim3((5 * p.N + 255) / 256), dim3(256), 0, st, part, S, p.N, NT, k_out, dW0, db0,
                       accumulate);
  }
};
inline int tail_tiles(int N) { const int nt = (N + 15) / 16; return nt > 8 && nt <= 13 ? 13 : nt > 13 && nt <= 16 ? 16 : 0; }

}  // namespace

namespace nsdp {
#ifdef NSDP_X3_TIMING
extern "C" void nsdp_debug_x3_timers(unsigned long long *out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_x3_timers), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_x3_timers), z, sizeof(z)); }
}
#endif
void debug_set_x3(int value) { g_x3_dbg = value; }
void debug_set_x3_reserve(int value) { g_x3_side_reserve = value; }
}  // namespace nsdp

extern "C" {

long long nsdp_packed_weight_bf16x3_bytes(int N, int K, int transposed) {
  const long long blocks = transposed ? static_cast<long long>((N + 31) / 32) * ((K + 15) / 16)
                                      : static_cast<long long>((K + 31) / 32) * ((N + 15) / 16);
  return blocks * 3 * 1024;
}

int nsdp_pack_weight_bf16x3(const float *W, int N, int K, void *Wp, void *WpT, void *stream) {
  if (N <= 0 || K <= 0) return 0;
  NSDP_REQUIRE(W && (Wp || WpT), "pack_weight_bf16x3: null pointer");
  const long long b0 = Wp ? nsdp_packed_weight_bf16x3_bytes(N, K, 0) / 3072 : 0;
  const long long b1 = WpT ? nsdp_packed_weight_bf16x3_bytes(N, K, 1) / 3072 : 0;
  const long long threads = (b0 > b1 ? b0 : b1) * 64;
  hipStream_t st = nsdp::as_stream(stream);
  hipLaunchKernelGGL(pack_bf16x3_kernel, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, st, W, N, K,
                     static_cast<u32x4 *>(Wp), static_cast<u32x4 *>(WpT));
  return nsdp::launch_status("pack_bf16x3_kernel");
}

int nsdp_linear_bf16x3_f32(const float *X, const void *Wp, const float *bias, const float *residual,
                           const float *mask, const float *out_mask, float *Y, long long M, int N, int K,
                           int relu_in, int relu_out, void *stream) {
  if (M <= 0 || N <= 0) return 0;
  NSDP_REQUIRE(X && Wp && Y, "linear_bf16x3: null pointer");
  NSDP_REQUIRE(K > 32 && K % 4 == 0, "linear_bf16x3: K=%d must be a multiple of 4 and > 32 (two k blocks)", K);
  NSDP_REQUIRE(N <= 256 && N % 4 == 0, "linear_bf16x3: N=%d must be a multiple of 4 and <= 256", N);
  NSDP_REQUIRE(((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Wp) | reinterpret_cast<uintptr_t>(Y) |
                 reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual) |
                 reinterpret_cast<uintptr_t>(mask) | reinterpret_cast<uintptr_t>(out_mask)) & 15) == 0,
               "linear_bf16x3: all operands must be 16-byte aligned");
  X3Params p{X, Wp, bias, residual, mask, out_mask, Y, M, N, K, relu_in, relu_out, g_x3_dbg};
  hipStream_t st = nsdp::as_stream(stream);
  const int nt = (N + 15) / 16;
  if (nt <= 4) return launch_x3<4>(p, st);
  if (nt <= 8) return launch_x3<8>(p, st);
  if (nt <= 13) return launch_x3<13>(p, st);
  return launch_x3<16>(p, st);
}

int nsdp_linear_bf16x3_gather_f32(const float *X, const void *Wp, const float *bias, const float *gq, int g_div, const float *gk,
                                  const int32_t *gidx, int g_rows_per_shape, int g_nsrc, float *Y, long long M, int N, int K,
                                  int relu_in, int relu_out, void *stream) {
  if (M <= 0 || N <= 0) return 0;
  NSDP_REQUIRE(X && Wp && Y && gk && gidx, "linear_bf16x3_gather: null pointer");
  NSDP_REQUIRE(K > 32 && K % 4 == 0, "linear_bf16x3_gather: K=%d must be a multiple of 4 and > 32 (two k blocks)", K);
  NSDP_REQUIRE(N <= 256 && N % 4 == 0, "linear_bf16x3_gather: N=%d must be a multiple of 4 and <= 256", N);
  NSDP_REQUIRE(M < (1LL << 31) && g_div > 0 && g_rows_per_shape > 0 && g_nsrc > 0, "linear_bf16x3_gather: bad row maps");
  NSDP_REQUIRE(!relu_in && !relu_out, "linear_bf16x3_gather: no fused ReLU on either side");
  {   // the kernel addresses both tables with 32-bit element offsets
    const long long q_rows = gq ? (M + g_div - 1) / g_div : 0, k_rows = ((M + g_rows_per_shape - 1) / g_rows_per_shape) * g_nsrc;
    NSDP_REQUIRE(q_rows * N < (1LL << 31) && k_rows * N < (1LL << 31), "linear_bf16x3_gather: tables beyond 2^31 elements");
  }
  NSDP_REQUIRE(((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Wp) | reinterpret_cast<uintptr_t>(Y) |
                 reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(gq) | reinterpret_cast<uintptr_t>(gk)) & 15) == 0,
               "linear_bf16x3_gather: all operands must be 16-byte aligned");
  X3Params p{X, Wp, bias, nullptr, nullptr, nullptr, Y, M, N, K, relu_in, relu_out, g_x3_dbg, gq, gk, gidx, g_div, g_rows_per_shape, g_nsrc};
  hipStream_t st = nsdp::as_stream(stream);
  const int nt = (N + 15) / 16;
  if (!gq) {      // one table holding the difference already (queries per shape): the look-ahead form
    if (nt <= 4) return launch_x3<4, 2>(p, st);
    if (nt <= 8) return launch_x3<8, 2>(p, st);
    if (nt <= 13) return launch_x3<13, 2>(p, st);
    return launch_x3<16, 2>(p, st);
  }
  if (nt <= 4) return launch_x3<4, 1>(p, st);
  if (nt <= 8) return launch_x3<8, 1>(p, st);
  if (nt <= 13) return launch_x3<13, 1>(p, st);
  return launch_x3<16, 1>(p, st);
}

int nsdp_linear_bf16x3_signed_f32(const float *X, const void *Wp, const float *bias, const float *residual, float residual_sign,
                                  float *Y, long long M, int N, int K, int relu_in, int relu_out, void *stream) {
  if (M <= 0 || N <= 0) return 0;
  NSDP_REQUIRE(X && Wp && Y && residual, "linear_bf16x3_signed: null pointer");
  NSDP_REQUIRE(residual_sign == 1.f || residual_sign == -1.f, "linear_bf16x3_signed: the residual's sign is +1 or -1");
  NSDP_REQUIRE(!relu_in, "linear_bf16x3_signed: no input ReLU (the sign lives in the plain-prologue kernels)");
  NSDP_REQUIRE(K > 32 && K % 4 == 0, "linear_bf16x3_signed: K=%d must be a multiple of 4 and > 32 (two k blocks)", K);
  NSDP_REQUIRE(N <= 256 && N % 4 == 0, "linear_bf16x3_signed: N=%d must be a multiple of 4 and <= 256", N);
  NSDP_REQUIRE(((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Wp) | reinterpret_cast<uintptr_t>(Y) |
                 reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0,
               "linear_bf16x3_signed: all operands must be 16-byte aligned");
  X3Params p{X, Wp, bias, residual, nullptr, nullptr, Y, M, N, K, relu_in, relu_out, g_x3_dbg};
  p.res_sign = residual_sign;
  hipStream_t st = nsdp::as_stream(stream);
  const int nt = (N + 15) / 16;
  if (nt <= 4) return launch_x3<4>(p, st);
  if (nt <= 8) return launch_x3<8>(p, st);
  if (nt <= 13) return launch_x3<13>(p, st);
  return launch_x3<16>(p, st);
}

// Second layer of a position-encoding MLP Linear(3 or 4, K) -> ReLU -> Linear(K, N) straight from the coordinates:
// Y[M,N] = relu(X4 W0^T + b0) W^T + bias (+ the gathered addend of nsdp_linear_bf16x3_gather_f32 when gk != NULL; gq == NULL:
// its one-table form).  The hidden tensor [M, K] is neither written nor read: the operand producer recomputes it (k4.h, the
// expression of nsdp_linear_k4 / its weight gradient's mask) -- values bit-identical to the two-launch form.  X4 [M,4]
// zero-padded rows, W0 [K,4] row-major zero-padded, b0 [K] or NULL, Wp the bf16x3 pack of W [N,K].
int nsdp_linear_bf16x3_h0_supported(long long M, int N, int K) {
  return M > 0 && M < (1LL << 31) && N > 64 && N <= 256 && N % 4 == 0 && K > 32 && K % 4 == 0 && K <= ((N + 15) / 16) * 16;
}
int nsdp_linear_bf16x3_h0_f32(const float *X4, const float *W0, const float *b0, const void *Wp, const float *bias,
                              const float *gq, int g_div, const float *gk, const int32_t *gidx, int g_rows_per_shape, int g_nsrc,
                              float *Y, long long M, int N, int K, void *stream) {
  if (M <= 0 || N <= 0) return 0;
  NSDP_REQUIRE(X4 && W0 && Wp && Y, "linear_bf16x3_h0: null pointer");
  NSDP_REQUIRE(nsdp_linear_bf16x3_h0_supported(M, N, K), "linear_bf16x3_h0: unsupported shape M=%lld N=%d K=%d", M, N, K);
  NSDP_REQUIRE(((reinterpret_cast<uintptr_t>(X4) | reinterpret_cast<uintptr_t>(W0) | reinterpret_cast<uintptr_t>(Wp) |
                 reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(gq) |
                 reinterpret_cast<uintptr_t>(gk)) & 15) == 0,
               "linear_bf16x3_h0: all operands must be 16-byte aligned");
  X3Params p{nullptr, Wp, bias, nullptr, nullptr, nullptr, Y, M, N, K, 0, 0, g_x3_dbg};
  p.h_x4 = X4; p.h_w0 = W0; p.h_b0 = b0;
  hipStream_t st = nsdp::as_stream(stream);
  const int nt = (N + 15) / 16;
  if (!gk) {
    if (nt <= 8) return launch_x3_h0<8, 0>(p, st);
    if (nt <= 13) return launch_x3_h0<13, 0>(p, st);
    return launch_x3_h0<16, 0>(p, st);
  }
  NSDP_REQUIRE(gidx && g_div > 0 && g_rows_per_shape > 0 && g_nsrc > 0, "linear_bf16x3_h0: bad row maps");
  {   // the kernel addresses both tables with 32-bit element offsets
    const long long q_rows = gq ? (M + g_div - 1) / g_div : 0, k_rows = ((M + g_rows_per_shape - 1) / g_rows_per_shape) * g_nsrc;
    NSDP_REQUIRE(q_rows * N < (1LL << 31) && k_rows * N < (1LL << 31), "linear_bf16x3_h0: tables beyond 2^31 elements");
  }
  p.gq = gq; p.gk = gk; p.gidx = gidx; p.g_div = g_div; p.g_rps = g_rows_per_shape; p.g_nsrc = g_nsrc;
  if (!gq) {
    if (nt <= 8) return launch_x3_h0<8, 2>(p, st);
    if (nt <= 13) return launch_x3_h0<13, 2>(p, st);
    return launch_x3_h0<16, 2>(p, st);
  }
  if (nt <= 8) return launch_x3_h0<8, 1>(p, st);
  if (nt <= 13) return launch_x3_h0<13, 1>(p, st);
  return launch_x3_h0<16, 1>(p, st);
}

// Can the dX GEMM dY [M,K] x W2 [K,N] of a position-encoding MLP's second layer take the first (K = 4) layer's weight gradient
// along (nsdp_linear_bf16x3_k4tail_f32)?  N = width of the hidden layer.
int nsdp_linear_bf16x3_k4tail_ok(long long M, int N, int K) {
  return M >= 65536 && M < (1LL << 31) && K > 32 && K % 4 == 0 && N % 4 == 0 && N <= 256 && tail_tiles(N) != 0;
}
size_t nsdp_linear_bf16x3_k4tail_workspace_bytes(long long M, int N) {
  if (M <= 0) return 0;
  return sizeof(float) * (tail_tiles(N) == 13 ? TailForm<2, 13, 8>::ws_floats(M) : TailForm<3, 16, 4>::ws_floats(M));
}
// dW0 [N, k_out], db0 [N] of h0 = relu(X4 W0^T + b0) from the gradient dY [M, K] of the NEXT layer's output: the dX GEMM
// Y = dY W2 (WpT = bf16x3 pack of W2^T, as for nsdp_linear_bf16x3_f32) with the masked reduction Y^T X4 in its epilogue; Y is
// never written.  X4 [M,4] (zero-padded coordinates), W0 [N,4] row-major zero-padded, b0 [N] or NULL.  accumulate: add to
// dW0 / db0.  Deterministic (fixed summation order).
int nsdp_linear_bf16x3_k4tail_f32(const float *dY, const void *WpT, const float *X4, const float *W0, const float *b0,
                                  float *dW0, float *db0, long long M, int N, int K, int k_out, int accumulate, float *ws,
                                  size_t ws_bytes, void *stream) {
  if (M <= 0 || N <= 0) return 0;
  NSDP_REQUIRE(dY && WpT && X4 && W0 && dW0 && ws, "linear_bf16x3_k4tail: null pointer");
  NSDP_REQUIRE(nsdp_linear_bf16x3_k4tail_ok(M, N, K), "linear_bf16x3_k4tail: unsupported shape M=%lld N=%d K=%d", M, N, K);
  NSDP_REQUIRE(k_out == 3 || k_out == 4, "linear_bf16x3_k4tail: k_out=%d (the layer's input width) must be 3 or 4", k_out);
  NSDP_REQUIRE(ws_bytes >= nsdp_linear_bf16x3_k4tail_workspace_bytes(M, N), "linear_bf16x3_k4tail: workspace too small");
  NSDP_REQUIRE(((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(WpT) | reinterpret_cast<uintptr_t>(X4) |
                 reinterpret_cast<uintptr_t>(W0) | reinterpret_cast<uintptr_t>(b0) | reinterpret_cast<uintptr_t>(ws)) & 15) == 0,
               "linear_bf16x3_k4tail: all operands must be 16-byte aligned");
  X3Params p{dY, WpT, nullptr, nullptr, nullptr, nullptr, nullptr, M, N, K, 0, 0, g_x3_dbg};
  p.t_x4 = X4; p.t_w0 = W0; p.t_b0 = b0; p.t_ws = ws;
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kLinearX3, st, 2.0 * M * N * K, 4.0 * (static_cast<double>(M) * (K + 4) + static_cast<double>(N) * K));
  if (tail_tiles(N) == 13) TailForm<2, 13, 8>::launch(p, k_out, dW0, db0, accumulate, st);
  else TailForm<3, 16, 4>::launch(p, k_out, dW0, db0, accumulate, st);
  return nsdp::launch_status("linear_bf16x3_kernel (k4 tail)");
}

int nsdp_linear_bf16x3_addend_f32(const float *X, const void *Wp, const float *bias, const float *residual, const float *mask,
                                  const float *out_mask, const float *addend, float *Y, long long M, int N, int K,
                                  int relu_out, void *stream) {
  if (M <= 0 || N <= 0) return 0;
  NSDP_REQUIRE(X && Wp && Y && mask && out_mask && addend, "linear_bf16x3_addend: null pointer (mask, out_mask and addend are required)");
  NSDP_REQUIRE(K > 32 && K % 4 == 0, "linear_bf16x3_addend: K=%d must be a multiple of 4 and > 32 (two k blocks)", K);
  NSDP_REQUIRE(N <= 256 && N % 4 == 0, "linear_bf16x3_addend: N=%d must be a multiple of 4 and <= 256", N);
  NSDP_REQUIRE(((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Wp) | reinterpret_cast<uintptr_t>(Y) |
                 reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(mask) |
                 reinterpret_cast<uintptr_t>(out_mask) | reinterpret_cast<uintptr_t>(addend)) & 15) == 0,
               "linear_bf16x3_addend: all operands must be 16-byte aligned");
  X3Params p{X, Wp, bias, residual, mask, out_mask, Y, M, N, K, 0, relu_out, g_x3_dbg};
  p.addend = addend;
  hipStream_t st = nsdp::as_stream(stream);
  const int nt = (N + 15) / 16;
  if (nt <= 4) return launch_x3<4>(p, st);
  if (nt <= 8) return launch_x3<8>(p, st);
  if (nt <= 13) return launch_x3<13>(p, st);
  return launch_x3<16>(p, st);
}

}  // extern "C"
