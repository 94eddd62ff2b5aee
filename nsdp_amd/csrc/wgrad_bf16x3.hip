// Weight gradient of a dense layer on the bf16 matrix pipe with the error-compensated 3-way split of
// gemm_bf16x3.hip (fp32 rounding-level accuracy, 6 bf16 MFMA products per fp32 product):
//
//     dW[N,K] = pre(dY)[M,N]^T pre(X)[M,K],   db[N] = colsum(pre(dY))
//
// Rows are the MFMA reduction dimension and BOTH operands are activations, so both must be split at run time.
// One persistent workgroup (4 waves, one per SIMD) walks a contiguous range of 32-row blocks:
//
//   produce  all 256 lanes cooperatively load the [32 rows x (N + K) columns] slab of the next block (a lane
//            owns 8 consecutive rows of ONE column per slot: consecutive lanes = consecutive columns, 256 B
//            per load instruction), apply the ReLU mask / ReLU, split into three bf16 planes on the VALU and
//            write them to LDS directly in MFMA fragment order (one ds_write_b128 per plane and slot) --
//            every value is loaded and split exactly once per workgroup;
//   consume  wave (a, b) owns the [n tiles of half a] x [k tiles of half b] quarter of dW (<= 7 x 7 tiles =
//            196 accumulator registers), reads its A (dY) and B (X) fragments back with conflict-free
//            ds_read_b128 and issues 6 MFMAs per tile pair and block.
//
// Two LDS plane buffers (2 x (NT + KT) x 3 KiB <= 156 KiB) and one barrier per block; the next block's
// global loads are issued one block ahead and its split is interleaved with the first MFMA steps.  Partial
// results go to a workspace as fragment-ordered float4 (one partial dW per workgroup) and are summed by a
// second kernel in fixed order (deterministic, no atomics).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "k4.h"
#include "prof.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

typedef __attribute__((address_space(3))) void *lds_ptr_t;

#ifdef WG3_TIMING
// phase timers (s_memtime ticks summed over waves): 0 producer steps, 1 plain steps, 2 bottom wait, 3 barrier, 4 total
__device__ unsigned long long g_wg3_timers[8];
#define WG3_T(var) var = __builtin_readcyclecounter()
#define WG3_ADD(i, a, b) t_acc[i] += (b) - (a)
#else
#define WG3_T(var)
#define WG3_ADD(i, a, b)
#endif

struct WgX3Params {
  const float *dY, *X, *mask;
  int relu_x;
  float *ws;            // [S][NT*KT*256 + NT*16] partials (fragment order, then db)
  long long M;
  int N, K;
  long long blocks_per_wg;   // 32-row blocks per workgroup
  int want_db;
  int kparts;                // grid.y: column ranges of X of KTB*16 columns each (K > 208: LDS holds N + K/2 columns)
  // H0 forms: X [M, 4] holds the 16-byte coordinate rows of a position-encoding MLP and the operand is its hidden layer
  // relu(X W0^T + b0) [M, K], recomputed by the producer (k4.h) instead of read; h_w0 [K, 4] row-major zero-padded, h_b0 [K] or NULL
  const float *h_w0 = nullptr, *h_b0 = nullptr;
  // BITS forms: the mask of a G16 dY as ReLU bits ([M / 16][ceil(N / 32)][64] bytes, x3_kernel.h) instead of an fp32 tensor
  const unsigned char *bits = nullptr;
};

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

__device__ __forceinline__ const float *sgpr_ptr(const float *q) {   // pin a wave-uniform pointer into SGPRs
  const unsigned long long u = reinterpret_cast<unsigned long long>(q);
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(u));
  const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(u >> 32));
  return reinterpret_cast<const float *>((static_cast<unsigned long long>(hi) << 32) | lo);
}

// hand-placed memory operations (the compiler sinks ordinary loads to their first use; see decoder_fused.hip)
__device__ __forceinline__ void gload(float &dst, const float *uniform_base, unsigned byte_off) {
  asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(uniform_base));
}
template <int OFF>
__device__ __forceinline__ void lds_read(u32x4 &dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void lds_wait3(u32x4 &a, u32x4 &b, u32x4 &c) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c));
}

// NTA / KTB: 16-column tiles of dY / X.  Wave (a, b) = (wave >> 1, wave & 1) owns n tiles [a TA, a TA + TA) and
// k tiles [b TB, b TB + TB).
// TAIL: M is not a multiple of 32 (the last block of the matrix is partial).  Without it the row clamps and row
// masks are compiled out: ~30 % of the VALU work of a producer step, and the producer steps are VALU-bound.
// (the column-wise form of rounds 1-2 -- lanes loading 8 rows of one column and writing fragment-ordered planes -- was
// retired in round 3: the row-wise kernel below is 1.09-1.17x faster on every shape, same error against fp64.)
// Row-wise form: the producer loads float4s of ROWS (1 KiB contiguous per wave instruction), splits them and writes the
// three planes as ROW-MAJOR bf16 images; the consumer fetches its operands with ds_read_b64_tr_b16 (lane i of a 16-lane
// group gets column i of the 4 x 16 block the group addresses -- half an operand of dY^T / X^T).  A lane group contracts
// over rows 4 g .. + 3 and 16 + 4 g .. + 3 (both operands agree), the image pitch is 32 (mod 64) bytes: one read's rows tile
// the LDS banks.
constexpr int tr_pitch_bytes(int cols) {
  int pitch = cols * 2;
  while (pitch % 64 != 32) pitch += 16;
  return pitch;
}
__device__ __forceinline__ void gload4(f32x4 &dst, const float *uniform_base, unsigned byte_off) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(uniform_base));
}
__device__ __forceinline__ void gload_i32(int &dst, const float *uniform_base, unsigned byte_off) {
  asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(uniform_base));
}
template <int OFF>
__device__ __forceinline__ void tr_read(unsigned long long &dst, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
struct Frag3 {
  unsigned long long h[2], m[2], l[2];
};
__device__ __forceinline__ void frag_wait_h(Frag3 &f) {      // (one-hot operand: only the h plane exists)
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.h[0]), "+v"(f.h[1]));
}
__device__ __forceinline__ void frag_wait(Frag3 &f) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.h[0]), "+v"(f.h[1]), "+v"(f.m[0]), "+v"(f.m[1]), "+v"(f.l[0]), "+v"(f.l[1]));
}
__device__ __forceinline__ u32x4 frag_vec(const unsigned long long (&q)[2]) {
  return u32x4{static_cast<unsigned>(q[0]), static_cast<unsigned>(q[0] >> 32), static_cast<unsigned>(q[1]),
               static_cast<unsigned>(q[1] >> 32)};
}

// ONEHOT ("scatter as a GEMM", nsdp_scatter_rows_onehot_f32): the dY operand is the one-hot matrix of a row -> table-row
// index list (p.dY points at int32 indices), generated by the producer straight into the h plane image -- 1.0 is exact in
// bf16, so table[a][c] = sum_{r: idx[r] = a} X[r][c] needs the three products 1 x {h, m, l} only and comes out in a fixed
// summation order (no atomics).  grid.y = shapes, each with its own p.M rows, index list and table.
// H0 (nsdp_linear_wgrad_bf16x3_h0_f32): the X operand is the hidden layer h0 = relu(x4 W0^T + b0) of a position-encoding MLP,
// recomputed from the 16-byte coordinate rows: a B slot loads its row's float4 instead of four floats of an [M, K] tensor and
// evaluates the K = 4 layer for its four columns (table rows from LDS: the 4 KiB the two plane buffers leave free hold 204).
// Producer slot -> (image row, float4 column) of a G16 operand.  64 consecutive slots (one wave instruction) cover the 16 rows of
// four consecutive channel quads of one row group -- one contiguous KiB of the tensor -- in the order rows 0-7 x 4 quads, then rows
// 8-15 x 4 quads: the 32 lanes an 8-byte LDS write resolves together then hit 8 rows x 4 quads of the plane image, whose pitch
// (32 mod 64 bytes) puts rows r and r + 8 into the same banks (all 16 rows of a quad per half-wave: 2-way conflicts, measured
// +2-5 % on the launch)
__device__ __forceinline__ void g16_slot(int s, int c4s, int &row, int &c4) {
  const int chunk = s >> 6, l = s & 63, chunks_per_group = c4s >> 2;
  const int grp = chunk / chunks_per_group;
  row = grp * 16 + ((l >> 5) << 3) + (l & 7);
  c4 = (chunk - grp * chunks_per_group) * 4 + ((l >> 3) & 3);
}

// LAY (nsdp_linear_wgrad_bf16x3_g16_f32): bit 0 = dY (and the mask), bit 1 = X stored in the G16 layout of gemm_bf16x3_g16.hip
// ([M / 16][C / 4][16 rows][4 floats]).  Only the producer's slot -> (row, float4 column) map and its addresses change: a slot of a
// G16 operand walks the rows of one channel quad first (one contiguous KiB per wave instruction), the plane images, the MFMA order
// and the bias sums are the row-major kernel's -- dW and db come out bit-identical.
template <int NTA, int KTB, bool MASK, bool TAIL, bool ONEHOT = false, bool H0 = false, int LAY = 0, bool BITS = false>
__global__ __launch_bounds__(256) void wgrad_bf16x3_rows_kernel(WgX3Params p) {
  static_assert(!BITS || (MASK && (LAY & 1) && !TAIL), "ReLU bits: the mask of a G16 dY, whole 32-row blocks");
  static_assert(!ONEHOT || (!MASK && NTA == 8), "one-hot operand: 128 table rows, no mask");
  static_assert(!H0 || (!MASK && !ONEHOT), "H0 operand: plain dY");
  static_assert(!LAY || (!ONEHOT && !H0), "G16 operands: the plain / masked forms");
  constexpr bool kAG = (LAY & 1) != 0, kBG = (LAY & 2) != 0;
  constexpr int kH0Rows = !H0 ? 0 : (KTB == 13 ? 204 : KTB * 16);
  constexpr int TA = (NTA + 1) / 2, TB = (KTB + 1) / 2;
  constexpr int kColsA = NTA * 16, kColsB = KTB * 16, kC4A = NTA * 4, kC4B = KTB * 4;
  constexpr int kPitchA = tr_pitch_bytes(kColsA), kPitchB = tr_pitch_bytes(kColsB);
  constexpr int kPlaneA = 32 * kPitchA, kPlaneB = 32 * kPitchB;
  constexpr unsigned kBufBytes = 3 * (kPlaneA + kPlaneB);            // [A h, m, l][B h, m, l], row-major images
  constexpr int RA = ONEHOT ? 1 : (32 * kC4A + 255) / 256, RB = (32 * kC4B + 255) / 256;   // float4 slots per lane
  __shared__ __attribute__((aligned(16))) unsigned char planes[2 * kBufBytes];
  __shared__ __attribute__((aligned(16))) f32x4 h_w0lds[H0 ? kH0Rows : 1];
  __shared__ __attribute__((aligned(16))) float h_b0lds[H0 ? kH0Rows : 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wave >> 1, wb = wave & 1;
  const int i = lane & 15, g = lane >> 4;
  const int N = p.N, K = p.K;
  const int k_off = ONEHOT ? 0 : static_cast<int>(blockIdx.y) * kColsB;
  const int Kpart = K - k_off < kColsB ? K - k_off : kColsB;
  // (one-hot: blockIdx.y is the shape; its index list and rows start blockIdx.y * M entries / rows in)
  const long long shape_rows = ONEHOT ? static_cast<long long>(blockIdx.y) * p.M : 0;
  const float *const dYp = sgpr_ptr(p.dY + shape_rows), *const Xp = sgpr_ptr(p.X + shape_rows * K), *const Mp = sgpr_ptr(p.mask);
  const long long Mrows = p.M;
  const long long mb0 = static_cast<long long>(blockIdx.x) * p.blocks_per_wg;
  const long long mb_all = (p.M + 31) >> 5;
  long long mb1 = mb0 + p.blocks_per_wg;
  mb1 = mb1 < mb_all ? mb1 : mb_all;
  const int nblk = static_cast<int>(mb1 - mb0);

  // ---- producer slots: s = tid + 256 r -> row s / c4s, float4 column s % c4s (surplus slots alias slot s - 256) ----
  unsigned offA[RA], offB[RB];       // byte offset of the slot's float4 in dY / X for the block loaded next
  unsigned offM[BITS ? RA : 1];      // (BITS) byte offset of the slot's mask byte; its nibble is mshift[r] bits up
  int mshift[BITS ? RA : 1];
  unsigned ldsA[RA], ldsB[RB];       // byte offset inside a plane image
  int rowA[RA], rowB[RB];            // image row (TAIL)
  int colB[H0 ? RB : 1];             // (H0) first of the slot's four columns, relative to k_off
  const unsigned strideA = ONEHOT ? 4u : static_cast<unsigned>(N) * 4u, strideB = H0 ? 16u : static_cast<unsigned>(K) * 4u;
  if constexpr (H0) {      // this column range's rows of the K = 4 layer's table (host contract: Kpart <= kH0Rows)
    // (pair-wise, k4.h; k_off is a multiple of 16 and Kpart of 4: a pair never straddles the range)
    for (int c = tid; c < kH0Rows; c += 256) h_b0lds[c] = (p.h_b0 && c < Kpart) ? p.h_b0[k_off + c] : 0.f;
    for (int j = tid; j < kH0Rows / 2; j += 256)
      nsdp::k4_pair_table(p.h_w0 + static_cast<long long>(k_off) * 4, Kpart, j, h_w0lds[2 * j], h_w0lds[2 * j + 1]);
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < RA; ++r) {
    if constexpr (ONEHOT) {      // thread = (image row tid / 8, 16-column segment tid % 8); the row's index is one dword
      rowA[r] = tid >> 3;
      offA[r] = static_cast<unsigned>((mb0 * 32 + (tid >> 3)) * 4);
      ldsA[r] = static_cast<unsigned>((tid >> 3) * kPitchA + (tid & 7) * 32);
    } else {
      int s = tid + 256 * r;
      s = s < 32 * kC4A ? s : s - 256;
      int row = s / kC4A, c4 = s % kC4A;
      if constexpr (kAG) g16_slot(s, kC4A, row, c4);
      const int col = 4 * c4 + 4 <= N ? 4 * c4 : N - 4;      // padding columns re-read the last real ones (never reduced)
      rowA[r] = row;
      offA[r] = kAG ? static_cast<unsigned>(((mb0 * 32 + (row & 16)) * N + col * 16 + (row & 15) * 4) * 4)
                    : static_cast<unsigned>(((mb0 * 32 + row) * N + col) * 4);
      ldsA[r] = static_cast<unsigned>(row * kPitchA + c4 * 8);
      if constexpr (BITS) {      // quad c4 = k block c4 / 8, half (c4 % 8) / 4, lane group c4 % 4 of the GEMM's fragment convention
        const int kbn = (N + 31) >> 5, cc = col >> 2;
        offM[r] = static_cast<unsigned>((mb0 * 2 + (row >> 4)) * kbn * 64 + (cc >> 3) * 64 + (row & 15) * 4 + (cc & 3));
        mshift[r] = ((cc >> 2) & 1) * 4;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    int s = tid + 256 * r;
    s = s < 32 * kC4B ? s : s - 256;
    int row = s / kC4B, c4 = s % kC4B;
    if constexpr (kBG) g16_slot(s, kC4B, row, c4);
    const int col = k_off + (4 * c4 + 4 <= Kpart ? 4 * c4 : Kpart - 4);
    rowB[r] = row;
    if constexpr (H0) {
      colB[r] = col - k_off;
      offB[r] = static_cast<unsigned>((mb0 * 32 + row) * 16);
    } else if constexpr (kBG) {
      offB[r] = static_cast<unsigned>(((mb0 * 32 + (row & 16)) * K + col * 16 + (row & 15) * 4) * 4);
    } else {
      offB[r] = static_cast<unsigned>(((mb0 * 32 + row) * K + col) * 4);
    }
    ldsB[r] = static_cast<unsigned>(3 * kPlaneA + row * kPitchB + c4 * 8);
  }
  const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_ptr_t)(&planes[0])));

  f32x4 rawA[RA], rawM[(MASK && !BITS) ? RA : 1], rawB[RB];
  unsigned rawMb[BITS ? RA : 1];
  const unsigned char *const Bp = p.bits;
  int rawI = -1;                     // (one-hot) table row of image row tid / 8 of the block loaded next
  f32x4 dbsum[RA];
#pragma unroll
  for (int r = 0; r < RA; ++r) dbsum[r] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue_a = [&](int r, long long mb) {
    unsigned off = offA[r];
    if (TAIL && mb * 32 + 32 > Mrows) {           // rows past M: re-read row M - 1 (zeroed in the split)
      const int last_row = static_cast<int>(Mrows - 1 - mb * 32);
      if constexpr (kAG) {                        // (M % 16 == 0: the second row group is missing -- re-read the first)
        if (rowA[r] > last_row) off -= 16u * strideA;
      } else {
        if (rowA[r] > last_row) off -= static_cast<unsigned>(rowA[r] - last_row) * strideA;
      }
    }
    if constexpr (ONEHOT) {
      gload_i32(rawI, dYp, off);
    } else {
      gload4(rawA[r], dYp, off);
      if constexpr (BITS) asm volatile("global_load_ubyte %0, %1, %2" : "=v"(rawMb[r]) : "v"(offM[r]), "s"(Bp));
      else if (MASK) gload4(rawM[r], Mp, off);
    }
  };
  auto issue_b = [&](int r, long long mb) {
    unsigned off = offB[r];
    if (TAIL && mb * 32 + 32 > Mrows) {
      const int last_row = static_cast<int>(Mrows - 1 - mb * 32);
      if constexpr (kBG) {
        if (rowB[r] > last_row) off -= 16u * strideB;
      } else {
        if (rowB[r] > last_row) off -= static_cast<unsigned>(rowB[r] - last_row) * strideB;
      }
    }
    gload4(rawB[r], Xp, off);
  };
  auto issue = [&](long long mb) {
#pragma unroll
    for (int r = 0; r < RA; ++r) issue_a(r, mb);
#pragma unroll
    for (int r = 0; r < RB; ++r) issue_b(r, mb);
  };
  auto advance = [&]() {
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      offA[r] += 32u * strideA;
      if constexpr (BITS) offM[r] += 2u * static_cast<unsigned>((N + 31) >> 5) * 64u;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) offB[r] += 32u * strideB;
  };
  auto gwait = [&]() {
    if constexpr (ONEHOT) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rawI));
#pragma unroll
    for (int r = 0; r < (ONEHOT ? 0 : RA); ++r) {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(rawA[r]));
      if constexpr (BITS) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rawMb[r]));
      else if (MASK) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rawM[r]));
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rawB[r]));
  };
  auto write3 = [&](const f32x4 &v, unsigned byte_off, int plane_bytes) {
    unsigned h0, m0, l0, h1, m1, l1;
    split_pair(v[0], v[1], h0, m0, l0);
    split_pair(v[2], v[3], h1, m1, l1);
    unsigned char *dst = &planes[0] + byte_off;
    *reinterpret_cast<unsigned long long *>(dst) = (static_cast<unsigned long long>(h1) << 32) | h0;
    *reinterpret_cast<unsigned long long *>(dst + plane_bytes) = (static_cast<unsigned long long>(m1) << 32) | m0;
    *reinterpret_cast<unsigned long long *>(dst + 2 * plane_bytes) = (static_cast<unsigned long long>(l1) << 32) | l0;
  };
  auto produce_a = [&](int r, unsigned buf_off, int rows_left, bool real) {
    if constexpr (ONEHOT) {
      // 16 columns of one image row: a single bf16 1.0 where the row's table index falls into them (rows past M: none)
      const int rel = ((TAIL && rowA[r] >= rows_left) ? -1 : rawI) - 16 * (tid & 7);
      u32x4 lo, hi;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        lo[w] = rel == 2 * w ? 0x00003f80u : (rel == 2 * w + 1 ? 0x3f800000u : 0u);
        hi[w] = rel == 8 + 2 * w ? 0x00003f80u : (rel == 9 + 2 * w ? 0x3f800000u : 0u);
      }
      unsigned char *dst = &planes[0] + buf_off + ldsA[r];
      *reinterpret_cast<u32x4 *>(dst) = lo;
      *reinterpret_cast<u32x4 *>(dst + 16) = hi;
      return;
    }
    f32x4 v = rawA[r];
    if constexpr (BITS) {
      const unsigned nib = rawMb[r] >> mshift[r];
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = ((nib >> c) & 1u) ? v[c] : 0.f;
    } else if (MASK) {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = rawM[r][c] > 0.f ? v[c] : 0.f;
    }
    if (TAIL) {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = rowA[r] < rows_left ? v[c] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) dbsum[r][c] += real ? v[c] : 0.f;     // (the split after the last block works on stale registers)
    write3(v, buf_off + ldsA[r], kPlaneA);
  };
  const float relu_floor = p.relu_x ? 0.f : -__builtin_inff();
  auto produce_b = [&](int r, unsigned buf_off) {
    f32x4 v;
    if constexpr (H0) {
      const float4 xq = make_float4(rawB[r][0], rawB[r][1], rawB[r][2], rawB[r][3]);
      const f32x4 bb = *reinterpret_cast<const f32x4 *>(&h_b0lds[colB[r]]);
#pragma unroll
      for (int c = 0; c < 4; c += 2) {      // (the slot's first column is a multiple of 4: two pairs of the table)
        const f32x4 wa = h_w0lds[colB[r] + c], wb = h_w0lds[colB[r] + c + 1];
        const f32x2 pre = nsdp::k4_preact_pair(xq, f32x2{wa[0], wa[1]}, f32x2{wa[2], wa[3]}, f32x2{wb[0], wb[1]}, f32x2{wb[2], wb[3]},
                                               f32x2{bb[c], bb[c + 1]});
        v[c] = fmaxf(pre[0], 0.f);
        v[c + 1] = fmaxf(pre[1], 0.f);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = __builtin_amdgcn_fmed3f(rawB[r][c], relu_floor, __builtin_inff());
    }
    write3(v, buf_off + ldsB[r], kPlaneB);
  };
  auto rows_in = [&](long long mb) {
    const long long left = p.M - mb * 32;
    return static_cast<int>(left < 32 ? left : 32);
  };

  f32x4 acc[TA][TB];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(mb0);
  gwait();
  {
    const int rl = rows_in(mb0);
#pragma unroll
    for (int r = 0; r < RA; ++r) produce_a(r, 0u, rl, true);
#pragma unroll
    for (int r = 0; r < RB; ++r) produce_b(r, 0u);
  }
  advance();
  if (nblk > 1) issue(mb0 + 1);
  advance();
  gwait();
  __syncthreads();

  // operand addresses: row 4 g + i / 4 (+ 16 for the second half), columns 16 t + 4 (i % 4) of the wave's first tile
  const unsigned fragA = lds_base + static_cast<unsigned>((4 * g + (i >> 2)) * kPitchA + 8 * (i & 3) + wa * TA * 32);
  const unsigned fragB = lds_base + static_cast<unsigned>(3 * kPlaneA + (4 * g + (i >> 2)) * kPitchB + 8 * (i & 3) + wb * TB * 32);
  auto read_a = [&](auto T, Frag3 &f, unsigned base) {
    constexpr int t = decltype(T)::value;
    tr_read<t * 32>(f.h[0], base); tr_read<t * 32 + 16 * kPitchA>(f.h[1], base);
    if constexpr (!ONEHOT) {
      tr_read<t * 32 + kPlaneA>(f.m[0], base); tr_read<t * 32 + kPlaneA + 16 * kPitchA>(f.m[1], base);
      tr_read<t * 32 + 2 * kPlaneA>(f.l[0], base); tr_read<t * 32 + 2 * kPlaneA + 16 * kPitchA>(f.l[1], base);
    }
  };
  auto read_b = [&](auto T, Frag3 &f, unsigned base) {
    constexpr int t = decltype(T)::value;
    tr_read<t * 32>(f.h[0], base); tr_read<t * 32 + 16 * kPitchB>(f.h[1], base);
    tr_read<t * 32 + kPlaneB>(f.m[0], base); tr_read<t * 32 + kPlaneB + 16 * kPitchB>(f.m[1], base);
    tr_read<t * 32 + 2 * kPlaneB>(f.l[0], base); tr_read<t * 32 + 2 * kPlaneB + 16 * kPitchB>(f.l[1], base);
  };
  constexpr int kSlots = RA + RB;
  constexpr int kAH = 4;
  constexpr int kPasses = (TA + kAH - 1) / kAH;
  constexpr int kSteps = kPasses * TB;
  constexpr int kPerStep = (kSlots + kSteps - 1) / kSteps;             // producer slots per consumer step

  for (int ib = 0; ib < nblk; ++ib) {
    const unsigned buf = static_cast<unsigned>(ib & 1) * kBufBytes, nbuf = kBufBytes - buf;
    const int rl_next = rows_in(mb0 + ib + 1 < mb_all ? mb0 + ib + 1 : mb_all - 1);
    static_for<0, kPasses>([&](auto P) {
      constexpr int pass = decltype(P)::value;
      constexpr int kShort = TA % kAH;
      constexpr int a0 = (kShort == 0) ? pass * kAH : (pass == 0 ? 0 : kShort + (pass - 1) * kAH);
      constexpr int na = (kShort != 0 && pass == 0) ? kShort : kAH;
      static_assert(a0 + na <= TA, "pass geometry");
      Frag3 fa[na];
      static_for<0, na>([&](auto I) {
        constexpr int a = decltype(I)::value;
        read_a(std::integral_constant<int, a0 + a>{}, fa[a], fragA + buf);
      });
      Frag3 fb;
      read_b(std::integral_constant<int, 0>{}, fb, fragB + buf);
      u32x4 ah[na], am[na], al[na];
      static_for<0, na>([&](auto I) {
        constexpr int a = decltype(I)::value;
        if constexpr (ONEHOT) {
          frag_wait_h(fa[a]);
          ah[a] = frag_vec(fa[a].h); am[a] = ah[a]; al[a] = ah[a];      // (m, l: unused)
        } else {
          frag_wait(fa[a]);
          ah[a] = frag_vec(fa[a].h); am[a] = frag_vec(fa[a].m); al[a] = frag_vec(fa[a].l);
        }
      });
      frag_wait(fb);
      u32x4 bh = frag_vec(fb.h), bm = frag_vec(fb.m), bl = frag_vec(fb.l);
      static_for<0, TB>([&](auto I) {
        constexpr int b = decltype(I)::value;
        constexpr int st = pass * TB + b;
        Frag3 nb;
        if constexpr (b + 1 < TB) read_b(std::integral_constant<int, b + 1>{}, nb, fragB + buf);
        // the next block's split: kPerStep producer slots per step, in the same basic block as the step's MFMAs
        static_for<0, kPerStep>([&](auto Q) {
          constexpr int slot = st * kPerStep + decltype(Q)::value;
          if constexpr (slot < RA) produce_a(slot, nbuf, rl_next, ib + 1 < nblk);
          else if constexpr (slot < kSlots) produce_b(slot - RA, nbuf);
        });
        if constexpr (!ONEHOT) {
#pragma unroll
          for (int a = 0; a < na; ++a) acc[a0 + a][b] = mfma_bf16(al[a], bh, acc[a0 + a][b]);
        }
#pragma unroll
        for (int a = 0; a < na; ++a) acc[a0 + a][b] = mfma_bf16(ah[a], bl, acc[a0 + a][b]);
        if constexpr (!ONEHOT) {
#pragma unroll
          for (int a = 0; a < na; ++a) acc[a0 + a][b] = mfma_bf16(am[a], bm, acc[a0 + a][b]);
#pragma unroll
          for (int a = 0; a < na; ++a) acc[a0 + a][b] = mfma_bf16(am[a], bh, acc[a0 + a][b]);
        }
#pragma unroll
        for (int a = 0; a < na; ++a) acc[a0 + a][b] = mfma_bf16(ah[a], bm, acc[a0 + a][b]);
#pragma unroll
        for (int a = 0; a < na; ++a) acc[a0 + a][b] = mfma_bf16(ah[a], bh, acc[a0 + a][b]);
        if constexpr (st * kPerStep < kSlots) {
#pragma unroll
          for (int q = 0; q < (ONEHOT ? 3 : 6) * na; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2 * kPerStep, 0);
          }
        }
#pragma unroll
        for (int a = 0; a < na; ++a) asm volatile("" : "+a"(acc[a0 + a][b]));
        // a slot's raw registers are free as soon as it is split: the loads of the block after next follow
        static_for<0, kPerStep>([&](auto Q) {
          constexpr int slot = st * kPerStep + decltype(Q)::value;
          if constexpr (slot < RA) {
            if (ib + 2 < nblk) issue_a(slot, mb0 + ib + 2);
          } else if constexpr (slot < kSlots) {
            if (ib + 2 < nblk) issue_b(slot - RA, mb0 + ib + 2);
          }
        });
        if constexpr (st == kSteps - 1) advance();
        if constexpr (b + 1 < TB) {
          frag_wait(nb);
          bh = frag_vec(nb.h); bm = frag_vec(nb.m); bl = frag_vec(nb.l);
        }
      });
    });
    gwait();
    __syncthreads();
  }

  float *wsp = p.ws + (static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) *
                         (static_cast<size_t>(NTA) * KTB * 256 + NTA * 16);
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      const int ta = wa * TA + a, tb = wb * TB + b;
      if (ta < NTA && tb < KTB) {
        const f32x4 v = acc[a][b];
        *reinterpret_cast<float4 *>(wsp + (static_cast<size_t>(ta) * KTB + tb) * 256 + lane * 4) =
            make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  if (p.want_db && blockIdx.y == 0) {
    // every slot (row, float4 column) has exactly one owner: park the per-slot sums in LDS ([32 rows][kColsA]) and add a
    // column's 32 rows in fixed order (deterministic)
    float *dbl = reinterpret_cast<float *>(&planes[0]);       // all plane reads are behind the last barrier
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const int s = tid + 256 * r;
      int srow = s / kC4A, sc4 = s % kC4A;
      if constexpr (kAG) g16_slot(s, kC4A, srow, sc4);
      if (s < 32 * kC4A) *reinterpret_cast<f32x4 *>(dbl + srow * kColsA + 4 * sc4) = dbsum[r];
    }
    __syncthreads();
    for (int c = tid; c < kColsA; c += 256) {
      float t = 0.f;
      for (int r = 0; r < 32; ++r) t += dbl[r * kColsA + c];
      wsp[static_cast<size_t>(NTA) * KTB * 256 + c] = t;
    }
  }
}

// dW[n][k] = sum over the S partials of element (n, k): accumulator (tile n/16, tile k/16), lane 16 (n%16)/4 + k%16,
// register (n%16)%4  (C/D layout of the MFMA: row = 4 (lane >> 4) + reg, column = lane & 15).
// 32 elements x 8 partial ranges per workgroup: a thread sums one eighth of the partials of one element (S/8
// independent loads in flight instead of a chain of S), the eight sub-sums are combined through LDS in fixed order.
__global__ __launch_bounds__(256) void wgrad_bf16x3_reduce_kernel(const float *__restrict__ ws, int S, int NTA, int KTB,
                                                                  int N, int K, float *__restrict__ dW,
                                                                  float *__restrict__ db, int accumulate) {
  __shared__ float part[8][32];
  const int el = threadIdx.x & 31, q = threadIdx.x >> 5;
  const long long e = static_cast<long long>(blockIdx.x) * 32 + el;
  const long long nw = static_cast<long long>(N) * K;
  const size_t stride = static_cast<size_t>(NTA) * KTB * 256 + NTA * 16;
  size_t idx = 0;
  bool valid = true;
  if (e < nw) {
    const int n = static_cast<int>(e / K), kg = static_cast<int>(e % K);
    const int part = kg / (KTB * 16), k = kg - part * KTB * 16;       // column range (grid.y of the main kernel)
    const int nl = n & 15, kl = k & 15;
    idx = static_cast<size_t>(part) * S * stride +
          (static_cast<size_t>(n >> 4) * KTB + (k >> 4)) * 256 + (16 * (nl >> 2) + kl) * 4 + (nl & 3);
  } else if (e < nw + N && db) {
    idx = static_cast<size_t>(NTA) * KTB * 256 + (e - nw);
  } else {
    valid = false;
  }
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (valid) {
    const int per = (S + 7) / 8;
    int s = q * per;
    const int s_end = s + per < S ? s + per : S;
    for (; s + 4 <= s_end; s += 4) {
      s0 += ws[idx + (s + 0) * stride]; s1 += ws[idx + (s + 1) * stride];
      s2 += ws[idx + (s + 2) * stride]; s3 += ws[idx + (s + 3) * stride];
    }
    for (; s < s_end; ++s) s0 += ws[idx + s * stride];
  }
  part[q][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (q == 0 && valid) {
    const float t = ((part[0][el] + part[1][el]) + (part[2][el] + part[3][el])) +
                    ((part[4][el] + part[5][el]) + (part[6][el] + part[7][el]));
    if (e < nw) dW[e] = accumulate ? dW[e] + t : t;
    else db[e - nw] = accumulate ? db[e - nw] + t : t;
  }
}

// The same sums for up to kReduceBatch layers in ONE launch (descriptors by value in the kernel arguments, grid.y = layer,
// grid.x covers the largest layer): element order, partial ranges and the combination are those of the kernel above, so a
// gradient does not depend on which of the two reduced it.
constexpr int kReduceBatch = 48;
struct ReduceBatch {
  NsdpWgradReduceDesc d[kReduceBatch];
};
__global__ __launch_bounds__(256) void wgrad_bf16x3_reduce_batched_kernel(ReduceBatch b) {
  __shared__ float part[8][32];
  const NsdpWgradReduceDesc &e = b.d[blockIdx.y];
  const int N = e.N, K = e.K, S = e.S, NTA = e.nta, KTB = e.ktb;
  const long long nw = static_cast<long long>(N) * K;
  const long long ne = nw + (e.db ? N : 0);
  if (static_cast<long long>(blockIdx.x) * 32 >= ne) return;      // (uniform per workgroup)
  const float *__restrict__ ws = e.ws;
  const int el = threadIdx.x & 31, q = threadIdx.x >> 5;
  const long long el_g = static_cast<long long>(blockIdx.x) * 32 + el;
  const size_t stride = static_cast<size_t>(NTA) * KTB * 256 + NTA * 16;
  size_t idx = 0;
  bool valid = true;
  if (el_g < nw) {
    const int n = static_cast<int>(el_g / K), kg = static_cast<int>(el_g % K);
    const int kpart = kg / (KTB * 16), k = kg - kpart * KTB * 16;
    const int nl = n & 15, kl = k & 15;
    idx = static_cast<size_t>(kpart) * S * stride +
          (static_cast<size_t>(n >> 4) * KTB + (k >> 4)) * 256 + (16 * (nl >> 2) + kl) * 4 + (nl & 3);
  } else if (el_g < nw + N && e.db) {
    idx = static_cast<size_t>(NTA) * KTB * 256 + (el_g - nw);
  } else {
    valid = false;
  }
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (valid) {
    const int per = (S + 7) / 8;
    int s = q * per;
    const int s_end = s + per < S ? s + per : S;
    for (; s + 4 <= s_end; s += 4) {
      s0 += ws[idx + (s + 0) * stride]; s1 += ws[idx + (s + 1) * stride];
      s2 += ws[idx + (s + 2) * stride]; s3 += ws[idx + (s + 3) * stride];
    }
    for (; s < s_end; ++s) s0 += ws[idx + s * stride];
  }
  part[q][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (q == 0 && valid) {
    const float t = ((part[0][el] + part[1][el]) + (part[2][el] + part[3][el])) +
                    ((part[4][el] + part[5][el]) + (part[6][el] + part[7][el]));
    if (el_g < nw) e.dW[el_g] = e.accumulate ? e.dW[el_g] + t : t;
    else e.db[el_g - nw] = e.accumulate ? e.db[el_g - nw] + t : t;
  }
}

thread_local int g_wg3_reserve = 0;      // host hint 9: per host thread (autograd runs one backward thread per device)

struct X3Plan {
  int nta, ktb, kparts, grid;
  long long blocks_per_wg;
  size_t ws_floats;
};

X3Plan plan_x3(long long M, int N, int K) {
  X3Plan pl;
  // instantiated tile classes: 8 (<= 128 columns), 13 (<= 208), 16 (<= 256, dY only).  Two plane buffers of
  // (nta + ktb) x 3 KiB must fit in 160 KiB: wider problems split the columns of X over grid.y
  pl.nta = N <= 128 ? 8 : N <= 208 ? 13 : 16;
  pl.ktb = K <= 128 ? 8 : 13;
  pl.kparts = 1;
  if (K > 208 || pl.nta + pl.ktb > 26) {
    pl.ktb = 8;
    pl.kparts = (K + 127) / 128;
  }
  const long long mb_all = (M + 31) / 32;
  // g_wg3_reserve (host hint, nsdp_debug_set(9, n)): compute units left free because this launch shares the chip with another
  // stream's kernels -- see hip_linear._wgrad_deferred.  NSDP_WGRAD_MAX_BLOCKS (experiment knob, read once): cap on the 32-row
  // blocks per workgroup (> 0: more, shorter workgroups instead of one persistent workgroup per CU; measured slower:
  // 42.4 / 44.2 / 47.6 ms at 64 / 16 / 8 against 41.8)
  const int reserve = g_wg3_reserve;
  static const int max_blocks = getenv("NSDP_WGRAD_MAX_BLOCKS") ? atoi(getenv("NSDP_WGRAD_MAX_BLOCKS")) : 0;
  long long grid = (nsdp::num_cus() - reserve) / pl.kparts;
  if (grid < 1) grid = 1;
  if (grid > mb_all / 4) grid = mb_all / 4 > 0 ? mb_all / 4 : 1;     // at least 4 blocks per workgroup
  pl.blocks_per_wg = (mb_all + grid - 1) / grid;
  if (max_blocks > 0 && pl.blocks_per_wg > max_blocks) pl.blocks_per_wg = max_blocks;
  pl.grid = static_cast<int>((mb_all + pl.blocks_per_wg - 1) / pl.blocks_per_wg);
  pl.ws_floats = static_cast<size_t>(pl.grid) * pl.kparts * (static_cast<size_t>(pl.nta) * pl.ktb * 256 + pl.nta * 16);
  return pl;
}

template <int NTA, int KTB>
void launch_wg(const WgX3Params &p, int grid, hipStream_t st) {
  const dim3 g(grid, p.kparts);
  const bool tail = (p.M & 31) != 0;
  if constexpr ((NTA == 8 && KTB == 8) || (NTA == 13 && KTB == 13) || (NTA == 16 && KTB == 8)) {
    if (p.h_w0) {
      NSDP_TRACE("wgrad_bf16x3<%d,%d,h0,%s>", NTA, KTB, tail ? "tail" : "notail");
      if (tail) hipLaunchKernelGGL((wgrad_bf16x3_rows_kernel<NTA, KTB, false, true, false, true>), g, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((wgrad_bf16x3_rows_kernel<NTA, KTB, false, false, false, true>), g, dim3(256), 0, st, p);
      return;
    }
  }
  NSDP_TRACE("wgrad_bf16x3<%d,%d,%s,%s>", NTA, KTB, p.mask ? "mask" : "plain", tail ? "tail" : "notail");
  if (p.mask) {
    if (tail) hipLaunchKernelGGL((wgrad_bf16x3_rows_kernel<NTA, KTB, true, true>), g, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((wgrad_bf16x3_rows_kernel<NTA, KTB, true, false>), g, dim3(256), 0, st, p);
  } else {
    if (tail) hipLaunchKernelGGL((wgrad_bf16x3_rows_kernel<NTA, KTB, false, true>), g, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((wgrad_bf16x3_rows_kernel<NTA, KTB, false, false>), g, dim3(256), 0, st, p);
  }
}

// G16 operands (LAY of the kernel): the square tile classes of the attention MLPs' hidden layers, whole 32-row blocks
template <int NTA, int KTB>
bool launch_wg_g16(const WgX3Params &p, int layout, int grid, hipStream_t st) {
  const dim3 g(grid, p.kparts);
  NSDP_TRACE("wgrad_bf16x3<%d,%d,%s,notail> g16:%s", NTA, KTB, p.bits ? "bits" : p.mask ? "mask" : "plain", layout == 1 ? "dy" : layout == 2 ? "x" : "dy,x");
  if (layout == 1 && p.bits) hipLaunchKernelGGL((wgrad_bf16x3_rows_kernel<NTA, KTB, true, false, false, false, 1, true>), g, dim3(256), 0, st, p);
  else if (layout == 1 && p.mask) hipLaunchKernelGGL((wgrad_bf16x3_rows_kernel<NTA, KTB, true, false, false, false, 1>), g, dim3(256), 0, st, p);
  else if (layout == 1) hipLaunchKernelGGL((wgrad_bf16x3_rows_kernel<NTA, KTB, false, false, false, false, 1>), g, dim3(256), 0, st, p);
  else if (layout == 2 && !p.mask) hipLaunchKernelGGL((wgrad_bf16x3_rows_kernel<NTA, KTB, false, false, false, false, 2>), g, dim3(256), 0, st, p);
  else return false;
  return true;
}

// table[b][n][k] = sum over the S partials of shape b (fragment order, see wgrad_bf16x3_reduce_kernel), fixed order
__global__ __launch_bounds__(256) void onehot_tables_reduce_kernel(const float *__restrict__ ws, int S, int NTA, int KTB, int N,
                                                                   int K, float *__restrict__ table) {
  const long long e = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (e >= static_cast<long long>(N) * K) return;
  const int n = static_cast<int>(e / K), k = static_cast<int>(e % K);
  const int nl = n & 15, kl = k & 15;
  const size_t stride = static_cast<size_t>(NTA) * KTB * 256 + NTA * 16;
  const float *w = ws + static_cast<size_t>(blockIdx.y) * S * stride +
                   (static_cast<size_t>(n >> 4) * KTB + (k >> 4)) * 256 + (16 * (nl >> 2) + kl) * 4 + (nl & 3);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int s = 0;
  for (; s + 4 <= S; s += 4) {
    a0 += w[(s + 0) * stride]; a1 += w[(s + 1) * stride]; a2 += w[(s + 2) * stride]; a3 += w[(s + 3) * stride];
  }
  for (; s < S; ++s) a0 += w[s * stride];
  table[static_cast<long long>(blockIdx.y) * N * K + e] = (a0 + a1) + (a2 + a3);
}

struct OneHotPlan {
  int ktb, grid;
  long long blocks_per_wg;
  size_t ws_floats;
};

OneHotPlan plan_onehot(int B, long long rows, int d) {
  OneHotPlan pl;
  pl.ktb = d <= 128 ? 8 : 13;
  const long long mb_all = (rows + 31) / 32;
  long long grid = (nsdp::num_cus() + B - 1) / B;      // workgroups per shape: one per CU in total (126 KiB of LDS each)
  if (grid > mb_all / 4) grid = mb_all / 4 > 0 ? mb_all / 4 : 1;
  pl.blocks_per_wg = (mb_all + grid - 1) / grid;
  pl.grid = static_cast<int>((mb_all + pl.blocks_per_wg - 1) / pl.blocks_per_wg);
  pl.ws_floats = static_cast<size_t>(B) * pl.grid * (static_cast<size_t>(8) * pl.ktb * 256 + 8 * 16);
  return pl;
}

}  // namespace

namespace nsdp {
void debug_set_wg3(int value) { g_wg3_reserve = value < 0 ? 0 : value; }      // knob 9: compute units to leave free (host hint)
}  // namespace nsdp

extern "C" {

int nsdp_linear_wgrad_bf16x3_supported(long long M, int N, int K) {
  return M >= 1024 && N > 16 && K > 16 && N <= 256 && K <= 256 &&
         static_cast<double>(M) * (N > K ? N : K) * 4.0 < 4.0e9;      // 32-bit byte offsets inside a tensor
}

size_t nsdp_linear_wgrad_bf16x3_workspace_bytes(long long M, int N, int K) {
  if (M <= 0) return 0;
  return plan_x3(M, N, K).ws_floats * sizeof(float);
}

int nsdp_linear_wgrad_bf16x3_partials_f32(const float *dY, const float *X, const float *mask, int relu_x, float *dW,
                                          float *db, long long M, int N, int K, int accumulate, float *workspace,
                                          size_t workspace_bytes, NsdpWgradReduceDesc *desc_out, void *stream) {
  NSDP_REQUIRE(nsdp_linear_wgrad_bf16x3_supported(M, N, K),
               "linear_wgrad_bf16x3_partials: shape M=%lld N=%d K=%d outside the kernel's range", M, N, K);
  NSDP_REQUIRE(dY && X && dW && workspace && desc_out, "linear_wgrad_bf16x3_partials: null pointer");
  NSDP_REQUIRE(workspace_bytes >= nsdp_linear_wgrad_bf16x3_workspace_bytes(M, N, K),
               "linear_wgrad_bf16x3_partials: workspace too small");
  const X3Plan pl = plan_x3(M, N, K);
  WgX3Params p{dY, X, mask, relu_x, workspace, M, N, K, pl.blocks_per_wg, db != nullptr, pl.kparts};
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kWgradX3, st, 2.0 * M * N * K, 4.0 * (static_cast<double>(M) * (K + N)));
  if (pl.nta == 8 && pl.ktb == 8) launch_wg<8, 8>(p, pl.grid, st);
  else if (pl.nta == 8) launch_wg<8, 13>(p, pl.grid, st);
  else if (pl.nta == 16) launch_wg<16, 8>(p, pl.grid, st);
  else if (pl.ktb == 8) launch_wg<13, 8>(p, pl.grid, st);
  else launch_wg<13, 13>(p, pl.grid, st);
  *desc_out = NsdpWgradReduceDesc{workspace, dW, db, pl.grid, pl.nta, pl.ktb, N, K, accumulate ? 1 : 0, 0};
  return nsdp::launch_status("wgrad_bf16x3_kernel");
}

// dW [N,K], db [N] of the SECOND layer of a position-encoding MLP Linear(3 or 4, K) -> ReLU -> Linear(K, N) from the gradient
// dY [M,N] of its output and the 16-byte coordinate rows X4 [M,4]: dW = dY^T relu(X4 W0^T + b0), the hidden tensor recomputed by
// the operand producer (bit for bit the values of the two-launch forward).  W0 [K,4] zero-padded, b0 [K] or NULL.  Workspace:
// nsdp_linear_wgrad_bf16x3_workspace_bytes(M, N, K).  desc_out != NULL: only the partial sums are formed and the reduction is
// described for nsdp_wgrad_bf16x3_reduce_batched (as nsdp_linear_wgrad_bf16x3_partials_f32); NULL: reduced here.
int nsdp_linear_wgrad_bf16x3_h0_supported(long long M, int N, int K) {
  if (!nsdp_linear_wgrad_bf16x3_supported(M, N, K) || K % 4 || N % 4 || M >= (1LL << 27)) return 0;
  const X3Plan pl = plan_x3(M, N, K);
  const bool form = (pl.nta == 8 && pl.ktb == 8) || (pl.nta == 13 && pl.ktb == 13) || (pl.nta == 16 && pl.ktb == 8);
  return form && (pl.ktb == 13 ? K <= 204 : true);
}
int nsdp_linear_wgrad_bf16x3_h0_f32(const float *dY, const float *X4, const float *W0, const float *b0, float *dW, float *db,
                                    long long M, int N, int K, int accumulate, float *workspace, size_t workspace_bytes,
                                    NsdpWgradReduceDesc *desc_out, void *stream) {
  NSDP_REQUIRE(nsdp_linear_wgrad_bf16x3_h0_supported(M, N, K),
               "linear_wgrad_bf16x3_h0: shape M=%lld N=%d K=%d outside the kernel's range", M, N, K);
  NSDP_REQUIRE(dY && X4 && W0 && dW && workspace, "linear_wgrad_bf16x3_h0: null pointer");
  NSDP_REQUIRE(workspace_bytes >= nsdp_linear_wgrad_bf16x3_workspace_bytes(M, N, K), "linear_wgrad_bf16x3_h0: workspace too small");
  NSDP_REQUIRE(((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(X4) | reinterpret_cast<uintptr_t>(W0)) & 15) == 0,
               "linear_wgrad_bf16x3_h0: operands must be 16-byte aligned");
  const X3Plan pl = plan_x3(M, N, K);
  WgX3Params p{dY, X4, nullptr, 0, workspace, M, N, K, pl.blocks_per_wg, db != nullptr, pl.kparts};
  p.h_w0 = W0; p.h_b0 = b0;
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kWgradX3, st, 2.0 * M * N * K, 4.0 * (static_cast<double>(M) * (4 + N)));
  if (pl.nta == 8) launch_wg<8, 8>(p, pl.grid, st);
  else if (pl.nta == 16) launch_wg<16, 8>(p, pl.grid, st);
  else launch_wg<13, 13>(p, pl.grid, st);
  const int rc = nsdp::launch_status("wgrad_bf16x3_kernel (h0)");
  if (rc) return rc;
  if (desc_out) {
    *desc_out = NsdpWgradReduceDesc{workspace, dW, db, pl.grid, pl.nta, pl.ktb, N, K, accumulate ? 1 : 0, 0};
    return 0;
  }
  const long long ne = static_cast<long long>(N) * K + N;
  hipLaunchKernelGGL(wgrad_bf16x3_reduce_kernel, dim3(static_cast<unsigned>((ne + 31) / 32)), dim3(256), 0, st,
                     workspace, pl.grid, pl.nta, pl.ktb, N, K, dW, db, accumulate);
  return nsdp::launch_status("wgrad_bf16x3_reduce_kernel");
}

int nsdp_wgrad_bf16x3_reduce_batched(const NsdpWgradReduceDesc *descs, int count, void *stream) {
  if (count <= 0) return 0;
  NSDP_REQUIRE(descs, "wgrad_bf16x3_reduce_batched: null descriptor array");
  hipStream_t st = nsdp::as_stream(stream);
  for (int base = 0; base < count; base += kReduceBatch) {
    ReduceBatch b;
    const int n = count - base < kReduceBatch ? count - base : kReduceBatch;
    long long blocks = 1;
    for (int i = 0; i < n; ++i) {
      const NsdpWgradReduceDesc &e = descs[base + i];
      NSDP_REQUIRE(e.ws && e.dW && e.S > 0 && e.N > 0 && e.K > 0 && e.nta > 0 && e.ktb > 0,
                   "wgrad_bf16x3_reduce_batched: bad descriptor %d", base + i);
      b.d[i] = e;
      const long long ne = static_cast<long long>(e.N) * e.K + (e.db ? e.N : 0);
      blocks = (ne + 31) / 32 > blocks ? (ne + 31) / 32 : blocks;
    }
    for (int i = n; i < kReduceBatch; ++i) b.d[i] = b.d[0];      // never indexed (grid.y = n)
    NSDP_TRACE("wgrad_bf16x3_reduce_batched x%d", n);
    hipLaunchKernelGGL(wgrad_bf16x3_reduce_batched_kernel, dim3(static_cast<unsigned>(blocks), n), dim3(256), 0, st, b);
    const int rc = nsdp::launch_status("wgrad_bf16x3_reduce_batched_kernel");
    if (rc) return rc;
  }
  return 0;
}

int nsdp_linear_wgrad_bf16x3_f32(const float *dY, const float *X, const float *mask, int relu_x, float *dW,
                                 float *db, long long M, int N, int K, int accumulate, float *workspace,
                                 size_t workspace_bytes, void *stream) {
  NSDP_REQUIRE(nsdp_linear_wgrad_bf16x3_supported(M, N, K),
               "linear_wgrad_bf16x3: shape M=%lld N=%d K=%d outside the kernel's range", M, N, K);
  NSDP_REQUIRE(dY && X && dW && workspace, "linear_wgrad_bf16x3: null pointer");
  NSDP_REQUIRE(workspace_bytes >= nsdp_linear_wgrad_bf16x3_workspace_bytes(M, N, K),
               "linear_wgrad_bf16x3: workspace too small");
  const X3Plan pl = plan_x3(M, N, K);
  WgX3Params p{dY, X, mask, relu_x, workspace, M, N, K, pl.blocks_per_wg, db != nullptr, pl.kparts};
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kWgradX3, st, 2.0 * M * N * K, 4.0 * (static_cast<double>(M) * (K + N)));
  {
    if (pl.nta == 8 && pl.ktb == 8) launch_wg<8, 8>(p, pl.grid, st);
    else if (pl.nta == 8) launch_wg<8, 13>(p, pl.grid, st);
    else if (pl.nta == 16) launch_wg<16, 8>(p, pl.grid, st);
    else if (pl.ktb == 8) launch_wg<13, 8>(p, pl.grid, st);
    else launch_wg<13, 13>(p, pl.grid, st);
    const int rc = nsdp::launch_status("wgrad_bf16x3_kernel");
    if (rc) return rc;
  }
  const long long ne = static_cast<long long>(N) * K + N;
  // (ablation, timing only, wrong results: NSDP_WG3_SKIP_REDUCE=1 drops the reduce launch -- the ceiling of what folding or
  // batching the 98 reduce launches of a step could buy)
  static const bool skip_reduce = getenv("NSDP_WG3_SKIP_REDUCE") && atoi(getenv("NSDP_WG3_SKIP_REDUCE")) != 0;
  if (skip_reduce) return 0;
  hipLaunchKernelGGL(wgrad_bf16x3_reduce_kernel, dim3(static_cast<unsigned>((ne + 31) / 32)), dim3(256), 0, st,
                     workspace, pl.grid, pl.nta, pl.ktb, N, K, dW, db, accumulate);
  return nsdp::launch_status("wgrad_bf16x3_reduce_kernel");
}

// nsdp_linear_wgrad_bf16x3_f32 / _partials_f32 with operands in the G16 layout (gemm_bf16x3_g16.hip): layout bit 0 = dY and the
// mask, bit 1 = X.  dW, db bit-identical to the row-major call on the same values.  desc_out != NULL: partial sums only (the
// reduction is described for nsdp_wgrad_bf16x3_reduce_batched); NULL: reduced here.
int nsdp_linear_wgrad_bf16x3_g16_supported(long long M, int N, int K, int layout, int has_mask) {
  if (!nsdp_linear_wgrad_bf16x3_supported(M, N, K) || M % 32 || N % 4 || K % 4) return 0;
  if (!(layout == 1 || (layout == 2 && !has_mask))) return 0;
  const X3Plan pl = plan_x3(M, N, K);
  return (pl.nta == 8 && pl.ktb == 8) || (pl.nta == 13 && pl.ktb == 13) || (pl.nta == 16 && pl.ktb == 8);
}
int nsdp_linear_wgrad_bf16x3_g16_f32(const float *dY, const float *X, const float *mask, int relu_x, float *dW, float *db,
                                     long long M, int N, int K, int accumulate, float *workspace, size_t workspace_bytes,
                                     NsdpWgradReduceDesc *desc_out, int layout, const unsigned char *mask_bits, void *stream) {
  NSDP_REQUIRE(nsdp_linear_wgrad_bf16x3_g16_supported(M, N, K, layout, mask != nullptr || mask_bits != nullptr),
               "linear_wgrad_bf16x3_g16: unsupported call M=%lld N=%d K=%d layout=%d mask=%d", M, N, K, layout, mask != nullptr || mask_bits != nullptr);
  NSDP_REQUIRE(!mask_bits || (layout == 1 && !mask), "linear_wgrad_bf16x3_g16: ReLU bits mask a G16 dY (layout 1), instead of `mask`");
  NSDP_REQUIRE(dY && X && dW && workspace, "linear_wgrad_bf16x3_g16: null pointer");
  NSDP_REQUIRE(workspace_bytes >= nsdp_linear_wgrad_bf16x3_workspace_bytes(M, N, K), "linear_wgrad_bf16x3_g16: workspace too small");
  const X3Plan pl = plan_x3(M, N, K);
  WgX3Params p{dY, X, mask, relu_x, workspace, M, N, K, pl.blocks_per_wg, db != nullptr, pl.kparts};
  p.bits = mask_bits;
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kWgradX3, st, 2.0 * M * N * K, 4.0 * (static_cast<double>(M) * (K + N)));
  const bool ok = pl.nta == 8 ? launch_wg_g16<8, 8>(p, layout, pl.grid, st)
                  : pl.nta == 16 ? launch_wg_g16<16, 8>(p, layout, pl.grid, st) : launch_wg_g16<13, 13>(p, layout, pl.grid, st);
  NSDP_REQUIRE(ok, "linear_wgrad_bf16x3_g16: form not instantiated");
  const int rc = nsdp::launch_status("wgrad_bf16x3_kernel (g16)");
  if (rc) return rc;
  if (desc_out) {
    *desc_out = NsdpWgradReduceDesc{workspace, dW, db, pl.grid, pl.nta, pl.ktb, N, K, accumulate ? 1 : 0, 0};
    return 0;
  }
  const long long ne = static_cast<long long>(N) * K + N;
  hipLaunchKernelGGL(wgrad_bf16x3_reduce_kernel, dim3(static_cast<unsigned>((ne + 31) / 32)), dim3(256), 0, st,
                     workspace, pl.grid, pl.nta, pl.ktb, N, K, dW, db, accumulate);
  return nsdp::launch_status("wgrad_bf16x3_reduce_kernel");
}

size_t nsdp_scatter_rows_onehot_f32_workspace_bytes(int B, long long rows, int N, int d) {
  if (B <= 0 || rows <= 0) return 0;
  (void)N;
  return plan_onehot(B, rows, d).ws_floats * sizeof(float);
}

int nsdp_scatter_rows_onehot_f32(const float *src, const int32_t *idx, int B, long long rows, int N, int d, float *table,
                                 float *workspace, size_t workspace_bytes, void *stream) {
  if (B <= 0 || N <= 0 || d <= 0) return 0;
  NSDP_REQUIRE(table, "scatter_rows_onehot_f32: null table");
  hipStream_t st = nsdp::as_stream(stream);
  if (rows <= 0) {
    NSDP_HIP_TRY(hipMemsetAsync(table, 0, sizeof(float) * static_cast<size_t>(B) * N * d, st));
    return 0;
  }
  NSDP_REQUIRE(src && idx && workspace, "scatter_rows_onehot_f32: null pointer");
  NSDP_REQUIRE(N <= 128 && d % 4 == 0 && d > 16 && d <= 208,
               "scatter_rows_onehot_f32: need N <= 128 table rows and 16 < d <= 208, d %% 4 == 0 (N=%d d=%d)", N, d);
  NSDP_REQUIRE(B <= 65535 && static_cast<double>(rows) * d * 4.0 < 4.0e9, "scatter_rows_onehot_f32: shape too large");
  NSDP_REQUIRE(workspace_bytes >= nsdp_scatter_rows_onehot_f32_workspace_bytes(B, rows, N, d),
               "scatter_rows_onehot_f32: workspace too small");
  const OneHotPlan pl = plan_onehot(B, rows, d);
  WgX3Params p{reinterpret_cast<const float *>(idx), src, nullptr, 0, workspace, rows, 128, d, pl.blocks_per_wg, 0, 1};
  nsdp::prof::Scope scope(nsdp::prof::kScatterRows, st, 0.0,
                          4.0 * (static_cast<double>(B) * rows * (d + 1) + static_cast<double>(B) * N * d));
  const dim3 g(pl.grid, B);
  const bool tail = (rows & 31) != 0;
  NSDP_TRACE("scatter_rows_onehot_f32<8,%d,%s>", pl.ktb, tail ? "tail" : "notail");
  if (pl.ktb == 8) {
    if (tail) hipLaunchKernelGGL((wgrad_bf16x3_rows_kernel<8, 8, false, true, true>), g, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((wgrad_bf16x3_rows_kernel<8, 8, false, false, true>), g, dim3(256), 0, st, p);
  } else {
    if (tail) hipLaunchKernelGGL((wgrad_bf16x3_rows_kernel<8, 13, false, true, true>), g, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((wgrad_bf16x3_rows_kernel<8, 13, false, false, true>), g, dim3(256), 0, st, p);
  }
  const int rc = nsdp::launch_status("wgrad_bf16x3_rows_kernel<onehot>");
  if (rc) return rc;
  const long long ne = static_cast<long long>(N) * d;
  hipLaunchKernelGGL(onehot_tables_reduce_kernel, dim3(static_cast<unsigned>((ne + 255) / 256), B), dim3(256), 0, st, workspace,
                     pl.grid, 8, pl.ktb, N, d, table);
  return nsdp::launch_status("onehot_tables_reduce_kernel");
}

#ifdef WG3_TIMING
void nsdp_debug_wg3_timers(unsigned long long *out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg3_timers), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_wg3_timers), z, sizeof(z)); }
}
#endif

}  // extern "C"
