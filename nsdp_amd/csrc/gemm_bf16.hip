// bf16-STORAGE dense layers for gfx950 (BASELINE config 3: activations and saved-for-backward tensors in bf16, fp32
// accumulation, fp32 master weights and fp32 weight gradients).
//
// With one bf16 product per multiply-add (v_mfma_f32_16x16x32_bf16, 16x the fp32-MFMA rate) every layer of the path
// (M ~ 10^5 .. 10^6 rows, N, K <= 256) is far below the ridge: 50 flop/B at 200 x 200 against 2500 TF / 8 TB/s = 312.
// These kernels are therefore organised as STREAMS, not as GEMMs:
//
//   nsdp_linear_bf16:  Y[M,N] = post( pre(X)[M,K] W[N,K]^T + b (+ residual) ).  The whole weight matrix (bf16,
//     fragment-major, <= 128 KiB) is loaded into LDS once per persistent workgroup; a wave owns 16 rows at a time,
//     reads them ONCE from HBM (16 B per lane and k block: lane group g holds k = 32 kb + 8 g .. + 7, so a row's four
//     lane groups read 64 contiguous bytes) a whole tile ahead, and writes 16 B per lane: the output channels of a
//     PAIR of 16-column tiles are interleaved in the weight pack (tile 2p row i <-> channel 32 p + 8 (i / 4) + i % 4,
//     tile 2p + 1 <-> the same + 4), so the eight accumulator values a lane holds for one row are eight CONSECUTIVE
//     channels = one 16-byte bf16 store, 64 contiguous bytes per row and instruction -- the same granularity as the
//     loads of the next layer.
//   nsdp_linear_wgrad_bf16:  dW[N,K] = pre(dY)^T pre(X), db = colsum(pre(dY)), fp32 out.  Rows are the MFMA k
//     dimension, so BOTH operands are needed column-wise.  256 B-contiguous row-major loads (a lane = 8 rows x 2
//     adjacent columns), a 16-op in-register 8 x 2 transpose, and 16-byte LDS writes straight into fragment order;
//     eight waves split the N x K output tiles; per-workgroup partials, fixed-order reduce (deterministic).
//
// Reference call sites: every nn.Linear / 1x1 nn.Conv1d of model/encoder/blocks.py and
// model/decoder/{blocks,crosstransformer_decoder}.py in the flow_arbitrary.py:30-48 train step.
#include <type_traits>

#include "common.h"
#include "prof.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ unsigned pack2(float a, float b) {   // (lo = a, hi = b), round to nearest even
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float hi_f(unsigned v) { return __builtin_bit_cast(float, v & 0xffff0000u); }
// bf16 > 0 as a sign/zero test on the raw halves (a ReLU mask is the saved ReLU output itself)
__device__ __forceinline__ unsigned keep_pos(unsigned v, unsigned m) {
  const unsigned lo = static_cast<int>(m << 16) > 0 ? 0x0000ffffu : 0u;
  const unsigned hi = static_cast<int>(m & 0xffff0000u) > 0 ? 0xffff0000u : 0u;
  return v & (lo | hi);
}
__device__ __forceinline__ unsigned relu2(unsigned v) {        // max(x, 0) on both halves (negative -> +0)
  const unsigned lo = (v & 0x00008000u) ? 0u : 0x0000ffffu;
  const unsigned hi = (v & 0x80000000u) ? 0u : 0xffff0000u;
  return v & (lo | hi);
}

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// output channel of MFMA row i of n tile nt (NT tiles in all): tiles are paired so that a lane's 2 x 4 accumulator
// values of one activation row are 8 consecutive channels; an unpaired last tile keeps the natural order
__host__ __device__ inline int chan_of(int nt, int i, int NT) {
  if ((nt | 1) < NT) return 32 * (nt >> 1) + 8 * (i >> 2) + 4 * (nt & 1) + (i & 3);
  return 16 * nt + i;
}

// ------------------------------------------------------------------------------------------------
// weight packs:  Wp [ceil(K/32)][ceil(N/16)][lane 16 g + i][8 bf16] = W[chan_of(nt, i)][32 kb + 8 g + j]
//                WpT[ceil(N/32)][ceil(K/16)][lane][8]               = W[32 nb + 8 g + j][chan_of(tk, i)]   (pack of W^T)
// one thread per (block, tile, lane) of each output, zero outside [N, K]
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack_b16_body(const float *__restrict__ W, int N, int K, u32x4 *__restrict__ Wp,
                                              u32x4 *__restrict__ WpT, long long q) {
  const int lane = static_cast<int>(q & 63), i = lane & 15, g = lane >> 4;
  const long long blk = q >> 6;
  if (Wp) {
    const int NT = (N + 15) >> 4, KB = (K + 31) >> 5;
    if (blk < static_cast<long long>(NT) * KB) {
      const int kb = static_cast<int>(blk / NT), nt = static_cast<int>(blk % NT);
      const int n = chan_of(nt, i, NT), k0 = kb * 32 + 8 * g;
      u32x4 v;
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        const int k = k0 + 2 * pr;
        const float a = n < N && k < K ? W[static_cast<long long>(n) * K + k] : 0.f;
        const float b = n < N && k + 1 < K ? W[static_cast<long long>(n) * K + k + 1] : 0.f;
        v[pr] = pack2(a, b);
      }
      Wp[blk * 64 + lane] = v;
    }
  }
  if (WpT) {
    const int KT = (K + 15) >> 4, NB = (N + 31) >> 5;
    if (blk < static_cast<long long>(KT) * NB) {
      const int nb = static_cast<int>(blk / KT), tk = static_cast<int>(blk % KT);
      const int k = chan_of(tk, i, KT), n0 = nb * 32 + 8 * g;
      u32x4 v;
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        const int n = n0 + 2 * pr;
        const float a = k < K && n < N ? W[static_cast<long long>(n) * K + k] : 0.f;
        const float b = k < K && n + 1 < N ? W[static_cast<long long>(n + 1) * K + k] : 0.f;
        v[pr] = pack2(a, b);
      }
      WpT[blk * 64 + lane] = v;
    }
  }
}

inline long long pack_b16_threads(int N, int K, bool fwd, bool transposed) {
  const long long b0 = fwd ? static_cast<long long>((K + 31) >> 5) * ((N + 15) >> 4) : 0;
  const long long b1 = transposed ? static_cast<long long>((N + 31) >> 5) * ((K + 15) >> 4) : 0;
  return (b0 > b1 ? b0 : b1) * 64;
}

struct PackOne {
  const float *W;
  u32x4 *Wp, *WpT;
  int N, K;
};
constexpr int kPackBatch = 64;
struct PackBatch {
  PackOne d[kPackBatch];
};

__global__ __launch_bounds__(256) void pack_b16_kernel(PackBatch b) {
  const PackOne &e = b.d[blockIdx.y];
  pack_b16_body(e.W, e.N, e.K, e.Wp, e.WpT, static_cast<long long>(blockIdx.x) * 256 + threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
// forward / dX
// ------------------------------------------------------------------------------------------------
struct B16Params {
  const unsigned short *X;         // [M,K] bf16
  const u32x4 *Wp;                 // pack, ceil(K/32) x ceil(N/16) KiB
  const float *bias;               // [N] fp32 or null
  const unsigned short *residual;  // [M,N] bf16 or null
  const unsigned short *mask;      // [M,K] bf16 or null: X * (mask > 0)
  const unsigned short *out_mask;  // [M,N] bf16 or null: Y * (out_mask > 0)
  void *Y;                         // [M,N] bf16, or fp32 when out_f32
  long long M;
  int N, K;
  int relu_in, relu_out, out_f32;
  int dbg;                         // ablation knob nsdp_debug_set(8, v): 1 no MFMA, 2 no stores (timing only)
};
int g_lin16_dbg = 0;

// WV = 8 waves per workgroup (two per SIMD, <= 256 registers per lane); a wave owns 16 rows at a time.  Every variant
// must be free of register spills AND of VGPR -> AGPR copies of the prefetch registers: a destination of an in-flight
// hand-issued load that is moved before the wait is moved before the data has arrived (tests/test_no_inflight_spills.py).

// hand-issued activation loads: hipcc sinks ordinary loads to their first use, which would serialise the next tile's
// HBM latency behind this tile's MFMAs (see gemm_bf16x3.hip); the registers become usable through xwait()
__device__ __forceinline__ void xload(u32x4 &dst, const void *lane_ptr) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(lane_ptr));
}

template <int NT, int KBM, bool MASK, int WV>
__global__ __launch_bounds__(WV * 64, 1) void linear_bf16_kernel(B16Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4 *wl = reinterpret_cast<u32x4 *>(smem_raw);                 // [KB][nt][64]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int K = p.K, N = p.N;
  const int KB = (K + 31) >> 5;
  const int ntiles = (N + 15) >> 4;            // tiles present in the pack (<= NT)

  // the whole weight matrix -> LDS, once per workgroup, as a full [KBM][NT] image (zero fragments where the pack has
  // none): the MFMA loop below is then branch-free -- one basic block whose LDS reads the compiler can pipeline
  for (int q = threadIdx.x; q < KBM * NT * 64; q += WV * 64) {
    const int blk = q >> 6, kb = blk / NT, nt = blk - kb * NT;
    wl[q] = (kb < KB && nt < ntiles) ? p.Wp[(kb * ntiles + nt) * 64 + (q & 63)] : u32x4{0u, 0u, 0u, 0u};
  }
  // ... and the bias behind it (read back as two broadcast ds_read_b128 per lane and tile pair: eight scalar global
  // loads per pair in the epilogue cost 25 % of the kernel)
  float *bl = reinterpret_cast<float *>(wl + KBM * NT * 64);
  for (int q = threadIdx.x; q < NT * 16; q += WV * 64) bl[q] = (p.bias && q < N) ? p.bias[q] : 0.f;
  __syncthreads();

  const long long tiles = (p.M + 15) >> 4;     // 16-row tiles, handed out wave by wave
  const long long stride = static_cast<long long>(gridDim.x) * WV;
  long long tile = static_cast<long long>(blockIdx.x) * WV + wave;
  if (tile >= tiles) return;                   // (no barriers below)

  // current tile, next tile (in flight during the MFMAs), and the next tile's mask: the mask is applied the moment
  // both have landed, so only one mask set is ever live
  u32x4 xc[KBM], xn[KBM], mn[MASK ? KBM : 1];
  auto issue = [&](long long t, u32x4 *x, u32x4 *m) {
    long long r = t * 16 + li;
    r = r < p.M ? r : (p.M - 1);
    const unsigned short *xr = p.X + r * K;
    const unsigned short *mr = MASK ? p.mask + r * K : nullptr;
    // UNCONDITIONAL: every one of the KBM loads is issued, k blocks beyond K re-read in-row data (their weight fragments
    // are zero).  A load under `if (kb < KB)` makes its destination a phi of "loaded" and "old value", and the register
    // allocator then issues the load into a temporary and copies it to the home register straight away -- a copy of a
    // register whose data has not arrived: the kernel silently computed with the PREVIOUS tile's rows
    // (tests/test_no_inflight_spills.py scans the ISA for exactly this).
#pragma unroll
    for (int kb = 0; kb < KBM; ++kb) {
      int ko = kb * 32 + 8 * g;
      ko = ko + 8 <= K ? ko : (K - 8);   // past the row end: re-read in-row data (the packed weights are zero there)
      xload(x[kb], xr + ko);
      if constexpr (MASK) xload(m[kb], mr + ko);
    }
  };
  auto xwait = [&](u32x4 *x, u32x4 *m) {      // ... and the prologue on the operand (mask / ReLU on the raw halves)
#pragma unroll
    for (int kb = 0; kb < KBM; ++kb) {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[kb]));
      if constexpr (MASK) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(m[kb]));
#pragma unroll
        for (int c = 0; c < 4; ++c) x[kb][c] = keep_pos(x[kb][c], m[kb][c]);
      }
      if (p.relu_in) {
#pragma unroll
        for (int c = 0; c < 4; ++c) x[kb][c] = relu2(x[kb][c]);
      }
    }
  };
  issue(tile, xc, mn);
  xwait(xc, mn);

  for (;;) {
    const long long next = tile + stride;
    const bool more = next < tiles;
    // (unconditional as well, for the same reason: the last tile prefetches itself again and drops the result)
    issue(more ? next : tile, xn, mn);

    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (the LDS image is loop-invariant: without an opaque offset LICM hoists all KB x NT fragment reads out of the
    // tile loop -- hundreds of live registers -> scratch, fatal next to in-flight hand-issued loads)
    int opaque = 0;
    asm volatile("" : "+s"(opaque));
    const u32x4 *wlane = wl + opaque + lane;
    if (!(p.dbg & 1)) {
#pragma unroll
      for (int kb = 0; kb < KBM; ++kb) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma_bf16(wlane[(kb * NT + nt) * 64], xc[kb], acc[nt]);
      }
    }

    // epilogue: lane (li, g) holds, for tile pair p, channels 32 p + 8 g .. + 7 of row 16 tile + li
    const long long row = tile * 16 + li;
    const bool rv = row < p.M && !(p.dbg & 2);
    const long long rowc = rv ? row : (p.M - 1);
    auto finish = [&](float *v, int c0, int cnt) {      // cnt = 8 or 4 consecutive channels from c0 (all < N)
      {
        const f32x4 b0 = *reinterpret_cast<const f32x4 *>(bl + c0);
        v[0] += b0[0]; v[1] += b0[1]; v[2] += b0[2]; v[3] += b0[3];
        if (cnt == 8) {
          const f32x4 b1 = *reinterpret_cast<const f32x4 *>(bl + c0 + 4);
          v[4] += b1[0]; v[5] += b1[1]; v[6] += b1[2]; v[7] += b1[3];
        }
      }
      if (p.residual) {
        const unsigned short *rr = p.residual + rowc * N + c0;
        if (cnt == 8) {
          const u32x4 r4 = *reinterpret_cast<const u32x4 *>(rr);
#pragma unroll
          for (int c = 0; c < 4; ++c) { v[2 * c] += lo_f(r4[c]); v[2 * c + 1] += hi_f(r4[c]); }
        } else {
          const u32x2 r2 = *reinterpret_cast<const u32x2 *>(rr);
#pragma unroll
          for (int c = 0; c < 2; ++c) { v[2 * c] += lo_f(r2[c]); v[2 * c + 1] += hi_f(r2[c]); }
        }
      }
      if (p.relu_out) {
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = fmaxf(v[c], 0.f);
      }
      if (p.out_f32) {
        float *yr = static_cast<float *>(p.Y) + rowc * N + c0;
        if (rv) {
#pragma unroll
          for (int c = 0; c < 8; ++c)
            if (c < cnt) yr[c] = v[c];
        }
        return;
      }
      unsigned short *yr = static_cast<unsigned short *>(p.Y) + rowc * N + c0;
      if (cnt == 8) {
        u32x4 o = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
        if (p.out_mask) {
          const u32x4 om = *reinterpret_cast<const u32x4 *>(p.out_mask + rowc * N + c0);
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] = keep_pos(o[c], om[c]);
        }
        if (rv) {
          if (p.dbg & 4) __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(yr));
          else *reinterpret_cast<u32x4 *>(yr) = o;
        }
      } else {
        u32x2 o = {pack2(v[0], v[1]), pack2(v[2], v[3])};
        if (p.out_mask) {
          const u32x2 om = *reinterpret_cast<const u32x2 *>(p.out_mask + rowc * N + c0);
          o[0] = keep_pos(o[0], om[0]); o[1] = keep_pos(o[1], om[1]);
        }
        if (rv) *reinterpret_cast<u32x2 *>(yr) = o;
      }
    };
    xwait(xn, mn);               // before the stores go out: vmcnt would otherwise also wait for them
#pragma unroll
    for (int pr = 0; pr < (NT + 1) / 2; ++pr) {
      if (2 * pr + 1 < ntiles) {                        // full pair
        const int c0 = 32 * pr + 8 * g;
        if (c0 < N) {
          float v[8] = {acc[2 * pr][0], acc[2 * pr][1], acc[2 * pr][2], acc[2 * pr][3],
                        acc[2 * pr + 1 < NT ? 2 * pr + 1 : 0][0], acc[2 * pr + 1 < NT ? 2 * pr + 1 : 0][1],
                        acc[2 * pr + 1 < NT ? 2 * pr + 1 : 0][2], acc[2 * pr + 1 < NT ? 2 * pr + 1 : 0][3]};
          if (c0 + 8 <= N) finish(v, c0, 8);
          else finish(v, c0, 4);                        // N % 8 == 4: the pair's last 4 channels are padding
        }
      } else if (2 * pr < ntiles) {                     // unpaired last tile: natural order, 4 channels per lane
        const int c0 = 32 * pr + 4 * g;
        if (c0 < N) {
          float v[8] = {acc[2 * pr][0], acc[2 * pr][1], acc[2 * pr][2], acc[2 * pr][3], 0.f, 0.f, 0.f, 0.f};
          if (p.out_f32 && c0 + 4 > N) {                // N % 4 != 0 (fc_out: N = 3), fp32 output only
            for (int c = 0; c < 4; ++c) v[c] += bl[c0 + c];
            if (p.relu_out) for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
            if (rv) for (int c = 0; c < 4; ++c) if (c0 + c < N) static_cast<float *>(p.Y)[rowc * N + c0 + c] = v[c];
          } else {
            finish(v, c0, 4);
          }
        }
      }
    }
    if (!more) break;
    tile = next;
#pragma unroll
    for (int kb = 0; kb < KBM; ++kb) xc[kb] = xn[kb];
  }
}

template <int NT, int KBM>
int launch_lin(const B16Params &p, hipStream_t st) {
  const size_t lds = static_cast<size_t>(KBM) * NT * 1024 + static_cast<size_t>(NT) * 64;
  const long long cus = nsdp::num_cus();
  auto go = [&](auto kern, const char *name, int wv) -> int {
    const long long wg_tiles = (p.M + wv * 16 - 1) / (wv * 16);
    const unsigned grid = static_cast<unsigned>(wg_tiles < cus ? wg_tiles : cus);
    if (lds > 64 * 1024)
      NSDP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds)));
    NSDP_TRACE("%s<%d,%d>", name, NT, KBM);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(wv * 64), lds, st, p);
    return nsdp::launch_status("linear_bf16_kernel");
  };
  if (p.mask) return go(linear_bf16_kernel<NT, KBM, true, 8>, "linear_bf16_mask", 8);
  return go(linear_bf16_kernel<NT, KBM, false, 8>, "linear_bf16", 8);
}

// ------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------
struct Wg16Params {
  const unsigned short *dY, *X, *mask;   // [M,N], [M,K], [M,N] bf16
  float *ws;                             // per-workgroup partials
  long long M;
  int N, K, relu_x, want_db;
  int slabs_per_wg;                      // 32-row slabs per workgroup
  int dbg;                               // ablation knob nsdp_debug_set(7, v): 1 no MFMA, 2 no transposition, 4 no DMA (timing only)
  int ring;                              // row-major slab images in the LDS ring (3 .. 6, what fits next to the 2 fragment images)
};
int g_wg16_dbg = 0;

constexpr int kWavesWg = 16;      // 4 per SIMD (<= 128 registers): the phases of a slab are latency chains, more waves hide them

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

// The kernel is a stream: per 32-row slab a workgroup moves 64 (N + K) bytes and does nt x kt MFMAs -- bytes in flight
// and instruction count are all that matter.
//  * 32 rows of a row-major tensor are ONE contiguous run: each slab goes global -> LDS as a linear copy by DMA
//    (global_load_lds, 1 KiB per wave instruction, no registers in flight, no address arithmetic), into a ring of RING
//    slots (as many as fit into the CU's 160 KiB: 3 .. 6), RING - 1 slabs ahead of its use (with 2 slabs in flight the
//    copy and the compute phases ran back to back: 262 us of DMA + 350 us of compute = 539 us); a wave waits for its own pieces with a counted s_waitcnt (every wave issues
//    exactly `pw` pieces per slab, surplus ones re-copy the last piece) in front of the one barrier per slab.
//  * A wave's 64 lanes then take 64 consecutive column pairs of one 8-row group of one tensor: 8 conflict-free
//    ds_read_b32 from the row-major image (mask / ReLU on the raw halves), an 8 x 2 transpose by v_perm_b32, two
//    ds_write_b128 into the MFMA fragment image (double buffered).  Tensor, rows and validity are wave-uniform.
//  * History: register-staged versions of this kernel were VALU-bound on per-lane 64-bit addresses (2.2 TB/s), and a
//    register ring with counted waits is unsafe in a loop (the register allocator swaps ring slots with v_mov across
//    the back edge -- copies of registers whose loads are still in flight).
// MODE 0: dW = dY^T X.  MODE 1: dY masked by (mask > 0).
// MODE 2 ("scatter as a GEMM"): the dY operand is the ONE-HOT matrix of a row -> table-row index, generated from the
// int32 index list (p.dY) in the transposition stage -- table[a][c] = sum_{r: idx[r] = a} X[r][c] comes out of the same
// MFMA loop, exact (a bf16 1.0 times a bf16 value, fp32 accumulation), with no atomics at all; grid.y = shapes, each with
// its own p.M rows and its own table.  2 M N K flops for a scatter is extravagant on paper and 0.1 ms in practice.
template <int TN, int TK, int TASKS, int MODE>
__global__ __launch_bounds__(kWavesWg * 64, 1) void wgrad_bf16_kernel(Wg16Params p) {
  constexpr bool MASK = MODE == 1, ONEHOT = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int N = p.N, K = p.K;
  if constexpr (ONEHOT) {                                  // this shape's rows
    p.X += static_cast<long long>(blockIdx.y) * p.M * K;
    p.dY = reinterpret_cast<const unsigned short *>(reinterpret_cast<const int *>(p.dY) + static_cast<long long>(blockIdx.y) * p.M);
  }
  const int nt = (N + 15) >> 4, kt = (K + 15) >> 4;
  const int tiles = nt + kt;
  const int img = tiles * 64;                          // u32x4 per fragment image
  // LDS: [2 fragment images][RING row-major slab images]; a slab image = dY piece region (nt KiB), X (kt KiB), mask (nt KiB)
  const int RING = p.ring;
  const int pieces = ONEHOT ? kt : tiles + (MASK ? nt : 0);          // 1 KiB pieces per slab (one-hot: X only)
  const int xoff = ONEHOT ? 0 : (nt << 10);            // X region inside a slab image
  u32x4 *smem = reinterpret_cast<u32x4 *>(smem_raw);
  unsigned char *rowimg = smem_raw + 2 * img * 16;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn0 = (wave >> 2) * TN, wk0 = (wave & 3) * TK;      // this wave's output tiles, waves laid out 4 (n) x 4 (k)
  constexpr int kThreads = kWavesWg * 64;

  const long long slab0 = static_cast<long long>(blockIdx.x) * p.slabs_per_wg;
  long long slab1 = slab0 + p.slabs_per_wg;
  const long long slabs_all = (p.M + 31) >> 5;
  slab1 = slab1 < slabs_all ? slab1 : slabs_all;

  // ---- DMA: piece q of a slab = 1 KiB number q of its (dY | X | mask) runs; wave w takes pieces w, w + 8, ...
  const int pw = (pieces + kWavesWg - 1) / kWavesWg;            // per wave and slab, surplus = re-copies of the last piece
  const long long bytes_dy = p.M * N * 2, bytes_x = p.M * K * 2;
  auto dma = [&](long long slab, int ring_idx) {   // (ring indices are carried along: a 64-bit modulo per slab is not free)
    if (p.dbg & 4) return;
    slab = slab < slab1 ? slab : (slab1 - 1);                    // past the end: dummy re-copy (keeps the count static)
    unsigned char *slot = rowimg + ring_idx * (pieces << 10);
    for (int i = 0; i < pw; ++i) {
      int q = wave + kWavesWg * i;
      q = q < pieces ? q : (pieces - 1);
      const bool isx = ONEHOT || (q >= nt && q < tiles);
      const int qq = ONEHOT ? q : (q < nt ? q : (q < tiles ? q - nt : q - tiles));   // piece within its tensor's run
      const unsigned char *base = reinterpret_cast<const unsigned char *>(isx ? p.X : (q < nt ? p.dY : p.mask));
      const long long total = isx ? bytes_x : bytes_dy;
      long long off = slab * 32 * (isx ? K : N) * 2 + (static_cast<long long>(qq) << 10) + lane * 16;
      off = off + 16 <= total ? off : (total - 16);              // the last slab's tail: re-read in-tensor bytes
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + off), (lds_ptr_t)(slot + (q << 10)), 16, 0, 0);
    }
  };

  // ---- transposition tasks: chunk = 64 column pairs of one (8-row group, tensor); chunk = wave + 8 t
  const int chn = ((N >> 1) + 63) >> 6, chk = ((K >> 1) + 63) >> 6, chs = chn + chk;     // chunks per row group
  int tsrc[TASKS], tmsk[TASKS], tdst[TASKS], trow[TASKS], tstride[TASKS], tcol[TASKS];
  bool tlive[TASKS], tisx[TASKS], tlane[TASKS];
#pragma unroll
  for (int t = 0; t < TASKS; ++t) {
    int chunk = wave + kWavesWg * t;
    tlive[t] = chunk < 4 * chs;                        // (uniform)
    chunk = tlive[t] ? chunk : 0;
    const int rg = chunk / chs, within = chunk - rg * chs;
    tisx[t] = within >= chn;
    const int c = (tisx[t] ? within - chn : within) * 64 + lane;
    const int cmax = (tisx[t] ? K : N) >> 1;
    tlane[t] = tlive[t] && c < cmax;
    const int cc = c < cmax ? c : 0;
    tstride[t] = (tisx[t] ? K : N) * 2;                // row stride in bytes
    trow[t] = rg * 8;
    tcol[t] = 2 * cc;
    tsrc[t] = (tisx[t] ? xoff : 0) + rg * 8 * tstride[t] + 4 * cc;     // byte offset inside a slab image
    tmsk[t] = (tiles << 10) + rg * 8 * tstride[t] + 4 * cc;
    tdst[t] = ((tisx[t] ? nt : 0) + (tcol[t] >> 4)) * 64 + rg * 16 + (tcol[t] & 15);
  }
  auto transpose = [&](long long slab, int parity, int ring_idx) {
    const unsigned char *slot = rowimg + ring_idx * (pieces << 10);
    u32x4 *dst = smem + parity * img;
#pragma unroll
    for (int t = 0; t < TASKS; ++t) {
      if (!tlive[t]) continue;                                   // uniform
      unsigned v[8];
      const long long rbase = slab * 32 + trow[t];
      if (ONEHOT && !tisx[t]) {                                  // uniform: lanes = pairs of table rows (2 c, 2 c + 1)
        // the 8 row indices of this task are wave-uniform: scalar loads (lgkmcnt -- a vector load here would make the
        // compiler wait vmcnt(0), i.e. for the whole DMA ring)
        const int *ip = reinterpret_cast<const int *>(p.dY) + __builtin_amdgcn_readfirstlane(static_cast<int>(rbase));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int a = rbase + j < p.M ? ip[j] : -1;
          v[j] = (a == tcol[t] ? 0x00003f80u : 0u) | (a == tcol[t] + 1 ? 0x3f800000u : 0u);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[j] = *reinterpret_cast<const unsigned *>(slot + tsrc[t] + j * tstride[t]);
          if constexpr (MASK) {
            if (!tisx[t]) v[j] = keep_pos(v[j], *reinterpret_cast<const unsigned *>(slot + tmsk[t] + j * tstride[t]));
          }
        }
      }
      if (rbase + 8 > p.M) {                                     // uniform, last slab only: rows past the end are zero
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (rbase + j >= p.M) v[j] = 0u;
      }
      if (tisx[t] && p.relu_x) {                                 // uniform
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = relu2(v[j]);
      }
      u32x4 even, odd;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        even[q] = __builtin_amdgcn_perm(v[2 * q + 1], v[2 * q], 0x05040100u);    // lo halves: column 2 c, rows 2q, 2q+1
        odd[q] = __builtin_amdgcn_perm(v[2 * q + 1], v[2 * q], 0x07060302u);     // hi halves: column 2 c + 1
      }
      if (tlane[t]) {
        dst[tdst[t]] = even;
        dst[tdst[t] + 1] = odd;
      }
    }
  };
  // all but this wave's youngest `younger` slabs of DMA pieces have landed
  auto dma_wait = [&](int younger) {
    // (pw is a runtime value <= 6: the immediate forms are selected by a uniform switch)
    const int n = younger * pw;
#define NSDP_VMCNT_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
      NSDP_VMCNT_CASE(0) NSDP_VMCNT_CASE(1) NSDP_VMCNT_CASE(2) NSDP_VMCNT_CASE(3) NSDP_VMCNT_CASE(4) NSDP_VMCNT_CASE(5)
      NSDP_VMCNT_CASE(6) NSDP_VMCNT_CASE(7) NSDP_VMCNT_CASE(8) NSDP_VMCNT_CASE(9) NSDP_VMCNT_CASE(10) NSDP_VMCNT_CASE(11)
      NSDP_VMCNT_CASE(12) NSDP_VMCNT_CASE(13) NSDP_VMCNT_CASE(14) NSDP_VMCNT_CASE(15) NSDP_VMCNT_CASE(16)
      NSDP_VMCNT_CASE(17) NSDP_VMCNT_CASE(18) NSDP_VMCNT_CASE(19) NSDP_VMCNT_CASE(20) NSDP_VMCNT_CASE(21)
      NSDP_VMCNT_CASE(22) NSDP_VMCNT_CASE(23) NSDP_VMCNT_CASE(24)
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;     // (host contract: (ring - 2) * pw <= 24)
    }
#undef NSDP_VMCNT_CASE
  };
  auto barrier = [&]() {   // raw: __syncthreads() carries a fence that also drains vmcnt -- the DMA this loop keeps in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  f32x4 acc[TN][TK], accb[TN];
#pragma unroll
  for (int a = 0; a < TN; ++a) {
    accb[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < TK; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // bias gradient = column sums of dY: one more MFMA per n tile against an all-ones operand, on the waves of k group 0
  const bool dbwave = p.want_db && (wave & 3) == 0;
  const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  // zero the padding columns of both fragment images once (tiles are written column pair by column pair)
  for (int q = threadIdx.x; q < 2 * img; q += kThreads) smem[q] = u32x4{0u, 0u, 0u, 0u};

  if (slab0 < slab1) {
    for (int d = 0; d < RING - 1; ++d) dma(slab0 + d, d);
    dma_wait(RING - 2);           // slab0 here (this wave's pieces) ...
    barrier();                    // ... and everybody's; the zero fill is complete as well
    if (!(p.dbg & 2)) transpose(slab0, 0, 0);
    dma(slab0 + RING - 1, RING - 1);
    dma_wait(RING - 2);           // slab0 + 1
    barrier();
    int ri = 0, par = 0;          // ring slot of slab s, fragment image of slab s
    for (long long s = slab0; s < slab1; ++s) {
      const int rn = ri + 1 == RING ? 0 : ri + 1;
      // slab s + 1: row-major image (complete since the last barrier) -> the other fragment image (free since then too)
      if (s + 1 < slab1 && !(p.dbg & 2)) transpose(s + 1, par ^ 1, rn);
      dma(s + RING, ri);          // into the ring slot of slab s, whose row image was consumed one iteration ago
      const u32x4 *step = smem + par * img + lane;
      ri = rn;
      par ^= 1;
      // the k-side fragments of this wave (TK of them) stay in registers, the n-side ones stream through
      u32x4 bf[TK];
#pragma unroll
      for (int b = 0; b < TK; ++b) bf[b] = (wk0 + b < kt) ? step[(nt + wk0 + b) * 64] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int a = 0; a < TN; ++a) {
        if (wn0 + a < nt && !(p.dbg & 1)) {
          const u32x4 af = step[(wn0 + a) * 64];
#pragma unroll
          for (int b = 0; b < TK; ++b) acc[a][b] = mfma_bf16(af, bf[b], acc[a][b]);
          if (dbwave) accb[a] = mfma_bf16(af, ones, accb[a]);      // (uniform) column sums of dY: D[n][*] = sum_r dY[r][n]
        }
      }
      dma_wait(RING - 2);         // slab s + 2 (outstanding: s + 2 .. s + RING)
      barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the trailing dummy copies
  __syncthreads();
  // partial of this workgroup: dW row-major [N][K] then db [N]
  float *out = p.ws + (static_cast<long long>(blockIdx.y) * gridDim.x + blockIdx.x) * (static_cast<long long>(N) * K + N);
  const int i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int a = 0; a < TN; ++a) {
    const int tn = wn0 + a;
    if (tn >= nt) continue;
#pragma unroll
    for (int b = 0; b < TK; ++b) {
      const int tkk = wk0 + b;
      if (tkk >= kt) continue;
      // D[row 4 g + r][col i]: row = output channel n, col = k
      const int k = tkk * 16 + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = tn * 16 + 4 * g + r;
        if (n < N && k < K) out[static_cast<long long>(n) * K + k] = acc[a][b][r];
      }
    }
  }
  if (dbwave && i == 0) {
#pragma unroll
    for (int a = 0; a < TN; ++a) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = (wn0 + a) * 16 + 4 * g + r;
        if (wn0 + a < nt && n < N) out[static_cast<long long>(N) * K + n] = accb[a][r];
      }
    }
  }
}

// dst[e] (+)= sum over the S partials, fixed order, 8 independent chains
__global__ __launch_bounds__(256) void reduce_b16_kernel(const float *__restrict__ ws, int S, long long stride,
                                                         long long nw, float *__restrict__ dW, long long nb,
                                                         float *__restrict__ db, int accumulate) {
  const long long e = blockIdx.x * 256LL + threadIdx.x;
  if (e >= nw + nb) return;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int c = 0;
  for (; c + 8 <= S; c += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] += ws[(c + u) * stride + e];
  }
  for (; c < S; ++c) acc[0] += ws[c * stride + e];
  const float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  if (e < nw) dW[e] = accumulate ? dW[e] + s : s;
  else if (db) db[e - nw] = accumulate ? db[e - nw] + s : s;
}

// the same sums for up to kReduceBatchB16 layers in one launch (descriptors by value, grid.y = layer): see
// nsdp_wgrad_bf16_reduce_batched -- the element order and the eight chains are those of reduce_b16_kernel (bit-identical)
constexpr int kReduceBatchB16 = 48;
struct ReduceBatchB16 {
  NsdpWgradB16ReduceDesc d[kReduceBatchB16];
};
__global__ __launch_bounds__(256) void reduce_b16_batched_kernel(ReduceBatchB16 b) {
  const NsdpWgradB16ReduceDesc &d = b.d[blockIdx.y];
  const long long e = blockIdx.x * 256LL + threadIdx.x;
  const long long nw = static_cast<long long>(d.N) * d.K, nb = d.db ? d.N : 0;
  if (e >= nw + nb) return;
  const float *__restrict__ ws = d.ws;
  const long long stride = nw + d.N;
  const int S = d.S;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int c = 0;
  for (; c + 8 <= S; c += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] += ws[(c + u) * stride + e];
  }
  for (; c < S; ++c) acc[0] += ws[c * stride + e];
  const float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  if (e < nw) d.dW[e] = d.accumulate ? d.dW[e] + s : s;
  else d.db[e - nw] = d.accumulate ? d.db[e - nw] + s : s;
}

// out[b][e] = sum over the S partials of shape b (one-hot scatter tables)
__global__ __launch_bounds__(256) void reduce_tables_kernel(const float *__restrict__ ws, int S, long long stride, long long nw,
                                                            float *__restrict__ out) {
  const long long e = blockIdx.x * 256LL + threadIdx.x;
  if (e >= nw) return;
  const float *w = ws + static_cast<long long>(blockIdx.y) * S * stride + e;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int c = 0;
  for (; c + 4 <= S; c += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] += w[(c + u) * stride];
  }
  for (; c < S; ++c) acc[0] += w[c * stride];
  out[static_cast<long long>(blockIdx.y) * nw + e] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// ------------------------------------------------------------------------------------------------
// weight gradient, transpose-read form: the MFMA operands come straight out of the row-major slab images
// ------------------------------------------------------------------------------------------------
// gfx950's ds_read_b64_tr_b16 hands lane i of a 16-lane group COLUMN i of the 4 x 16 block whose rows the group's lanes
// point at (lane i: row i / 4, columns 4 (i % 4) .. + 3; any row pitch) -- exactly half an MFMA operand of dY^T or X^T
// (lane (i, g): column i, rows 8 g .. 8 g + 7 = two such reads).  So the slab images the DMA ring holds ARE the operand
// store: no transposition pass, no fragment images, the whole LDS is ring (up to 6 slabs in flight).  A slab image is
// [32 rows][pitch]: the DMA's source addresses are chosen per lane so that every row starts `pitch` bytes after the last
// one, pitch = row bytes + 16 * pad = 32 (mod 64) (a 512-byte pitch would put all rows on the same banks).  Needs N % 8 == 0 and K % 8 == 0 (16-byte DMA granules); other shapes keep the kernel above.
struct Wg16TrParams {
  const unsigned char *dY, *X, *mask;    // bf16 [M,N], [M,K], [M,N]
  float *ws;
  long long M;
  int N, K, relu_x, want_db;
  int slabs_per_wg, dbg, ring;
  int pitch_a, pitch_b;                  // bytes per image row (dY / mask, X)
  int slot_bytes, pieces;                // bytes of one ring slot (multiple of 1 KiB), DMA pieces per slab
};

template <int I, int NN, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < NN) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, NN>(f);
  }
}

template <int OFF>
__device__ __forceinline__ void tr_read(unsigned long long &dst, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

// SR: a slab (one barrier, one ring slot) is SR x 32 rows = SR MFMA steps: narrow layers move too few bytes per 32 rows
// to pay for a barrier each
template <int TN, int TK, bool MASK, int SR>
__global__ __launch_bounds__(kWavesWg * 64, 1) void wgrad_bf16_tr_kernel(Wg16TrParams p) {
  constexpr int kRows = 32 * SR;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int N = p.N, K = p.K;
  const int nt = (N + 15) >> 4, kt = (K + 15) >> 4;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int wn0 = (wave >> 2) * TN, wk0 = (wave & 3) * TK;
  const int RING = p.ring;
  // images inside a slot start at KiB boundaries, so that a DMA piece (1 KiB) belongs to ONE tensor: uniform base pointer
  const int off_b = (kRows * p.pitch_a + 1023) & ~1023, off_m = off_b + ((kRows * p.pitch_b + 1023) & ~1023);

  const long long slab0 = static_cast<long long>(blockIdx.x) * p.slabs_per_wg;
  long long slab1 = slab0 + p.slabs_per_wg;
  const long long slabs_all = (p.M + kRows - 1) / kRows;
  slab1 = slab1 < slabs_all ? slab1 : slabs_all;

  // ---- DMA: this wave's pieces (1 KiB of a slot each): per lane the tensor, the byte offset inside a slab and the
  //      slab stride of its 16-byte granule; granules in the padding re-read granule 0 of dY (never used)
  constexpr int kPwMax = 3;
  // this wave's pieces: wave, wave + 16, ... (its own count: every wave waits for exactly what it issued)
  const int pw = (p.pieces - wave + kWavesWg - 1) / kWavesWg;     // 0 .. kPwMax (host)
  // per piece: the tensor (uniform) and, per lane, the byte offset of its 16-byte granule inside a slab (low 24 bits) and
  // its image row (high bits; 255 = a granule of the padding, which re-reads granule 0)
  unsigned goff[kPwMax];
#pragma unroll
  for (int u = 0; u < kPwMax; ++u) {
    const int q = wave + kWavesWg * u < p.pieces ? wave + kWavesWg * u : p.pieces - 1;
    const int o = (q << 10) + lane * 16;
    const bool isb = (q << 10) >= off_b && (q << 10) < off_m;       // (uniform)
    const int width = (isb ? K : N) * 2, pitch = isb ? p.pitch_b : p.pitch_a;
    const int rel = o - (isb ? off_b : ((q << 10) >= off_m ? off_m : 0));
    const int r = rel / pitch, w = rel - r * pitch;
    const bool real = r < kRows && w < width;
    goff[u] = real ? (static_cast<unsigned>(r * width + w) | (static_cast<unsigned>(r) << 24)) : (255u << 24);
  }
  auto dma = [&](long long slab, int ring_idx) {     // (the ring index is carried along: a 64-bit modulo per slab is not free)
    if (p.dbg & 4) return;
    slab = slab < slab1 ? slab : (slab1 - 1);                      // past the end: re-copy (keeps the count static)
    unsigned char *slot = smem_raw + ring_idx * p.slot_bytes;
    const long long left = p.M - slab * kRows;                     // (uniform) < kRows: the matrix's last, partial slab
    const int rows_here = left < kRows ? static_cast<int>(left) : kRows;
#pragma unroll
    for (int u = 0; u < kPwMax; ++u) {
      if (u >= pw) break;
      const int q = wave + kWavesWg * u;
      const bool isb = (q << 10) >= off_b && (q << 10) < off_m;
      const unsigned char *tensor = isb ? p.X : (MASK && (q << 10) >= off_m ? p.mask : p.dY);
      const long long stride = 2LL * kRows * (isb ? K : N);
      // rows past M: the same granule of the slab before (in-tensor bytes; zeroed in the operand registers)
      const unsigned char *sbase = tensor + slab * stride;          // (uniform)
      unsigned off = goff[u] & 0xffffffu;
      const int row = static_cast<int>(goff[u] >> 24);
      const unsigned char *src = sbase + off;
      if (rows_here < kRows && row >= rows_here && row < kRows) src -= stride;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(slot + (q << 10)), 16, 0, 0);
    }
  };
  auto dma_wait = [&](int younger) {
    const int n = younger * pw;
#define NSDP_VMCNT_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
      NSDP_VMCNT_CASE(0) NSDP_VMCNT_CASE(1) NSDP_VMCNT_CASE(2) NSDP_VMCNT_CASE(3) NSDP_VMCNT_CASE(4) NSDP_VMCNT_CASE(5)
      NSDP_VMCNT_CASE(6) NSDP_VMCNT_CASE(7) NSDP_VMCNT_CASE(8) NSDP_VMCNT_CASE(9) NSDP_VMCNT_CASE(10) NSDP_VMCNT_CASE(11)
      NSDP_VMCNT_CASE(12) NSDP_VMCNT_CASE(13) NSDP_VMCNT_CASE(14) NSDP_VMCNT_CASE(15) NSDP_VMCNT_CASE(16)
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;     // (host contract: (ring - 2) * pw <= 16)
    }
#undef NSDP_VMCNT_CASE
  };
  auto barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // bias gradient = column sums of dY: the waves of k group 0 add up the eight rows their dY operand holds per column (fp32
  // adds of exact bf16 values; one register per n tile -- an MFMA against ones would need four)
  f32x4 acc[TN][TK];
  float dbs[TN];
#pragma unroll
  for (int a = 0; a < TN; ++a) {
    dbs[a] = 0.f;
#pragma unroll
    for (int b = 0; b < TK; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool dbwave = p.want_db && (wave & 3) == 0;
  // operand addresses of this lane inside a slot: row 8 g + i / 4 (+ 4 for the second half), columns 16 t + 4 (i % 4)
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_ptr_t)smem_raw));
  // (which 8 of the slab's 32 rows a lane group contracts over is free as long as both operands agree: group g takes rows
  // 4 g .. 4 g + 3 and 16 + 4 g .. + 3, so that one read instruction covers 16 CONSECUTIVE rows -- with a pitch of 32 bytes
  // modulo 64 the eight rows of a 32-lane pass then tile the 64 banks exactly)
  const unsigned a_lo = (4 * g + (i >> 2)) * p.pitch_a + 8 * (i & 3) + 32 * wn0;
  const unsigned b_lo = off_b + (4 * g + (i >> 2)) * p.pitch_b + 8 * (i & 3) + 32 * wk0;

  if (slab0 < slab1) {
    for (int d = 0; d < RING - 1; ++d) dma(slab0 + d, d);
    dma_wait(RING - 2);
    barrier();
    int ri = 0;                   // ring slot of slab s
    for (long long s = slab0; s < slab1; ++s) {
      dma(s + RING - 1, ri == 0 ? RING - 1 : ri - 1);      // into the slot of slab s - 1, read by everybody before the last barrier
      const unsigned slot = lds0 + static_cast<unsigned>(ri) * p.slot_bytes;
      ri = ri + 1 == RING ? 0 : ri + 1;
#pragma unroll
      for (int sr = 0; sr < SR; ++sr) {
      const unsigned aa = slot + a_lo + sr * 32 * p.pitch_a, ab = aa + 16 * p.pitch_a;
      const unsigned ba = slot + b_lo + sr * 32 * p.pitch_b, bb = ba + 16 * p.pitch_b;
      // X operands of this wave up front, then one n tile at a time
      unsigned long long rb[TK][2];
      static_for<0, TK>([&](auto B) {
        constexpr int b = decltype(B)::value;
        tr_read<32 * b>(rb[b][0], ba); tr_read<32 * b>(rb[b][1], bb);
      });
#pragma unroll
      for (int b = 0; b < TK; ++b) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rb[b][0]), "+v"(rb[b][1]));
      const long long rl = p.M - s * kRows - sr * 32;             // < 32 only in the matrix's last slab (uniform)
      const int rows_left = rl < 32 ? (rl < 0 ? 0 : static_cast<int>(rl)) : 32;
      u32x4 bf[TK];
#pragma unroll
      for (int b = 0; b < TK; ++b) {
        bf[b] = u32x4{static_cast<unsigned>(rb[b][0]), static_cast<unsigned>(rb[b][0] >> 32),
                      static_cast<unsigned>(rb[b][1]), static_cast<unsigned>(rb[b][1] >> 32)};
        if (p.relu_x) {
#pragma unroll
          for (int c = 0; c < 4; ++c) bf[b][c] = relu2(bf[b][c]);
        }
      }
      static_for<0, TN>([&](auto A) {
        constexpr int a = decltype(A)::value;
        if (wn0 + a < nt && !(p.dbg & 1)) {      // (uniform)
          unsigned long long ra[2], rm[2];
          tr_read<32 * a>(ra[0], aa); tr_read<32 * a>(ra[1], ab);
          if constexpr (MASK) { tr_read<32 * a>(rm[0], aa + off_m); tr_read<32 * a>(rm[1], ab + off_m); }
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]));
          u32x4 af = {static_cast<unsigned>(ra[0]), static_cast<unsigned>(ra[0] >> 32),
                      static_cast<unsigned>(ra[1]), static_cast<unsigned>(ra[1] >> 32)};
          if constexpr (MASK) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rm[0]), "+v"(rm[1]));
            const u32x4 mf = {static_cast<unsigned>(rm[0]), static_cast<unsigned>(rm[0] >> 32),
                              static_cast<unsigned>(rm[1]), static_cast<unsigned>(rm[1] >> 32)};
#pragma unroll
            for (int c = 0; c < 4; ++c) af[c] = keep_pos(af[c], mf[c]);
          }
          if (rows_left < 32) {       // dword c of the operand = rows 16 (c / 2) + 4 g + 2 (c % 2), + 1
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const int r = 16 * (c >> 1) + 4 * g + 2 * (c & 1);
              af[c] = r + 1 < rows_left ? af[c] : (r < rows_left ? (af[c] & 0xffffu) : 0u);
            }
          }
#pragma unroll
          for (int b = 0; b < TK; ++b)
            if (wk0 + b < kt) acc[a][b] = mfma_bf16(af, bf[b], acc[a][b]);
          if (dbwave) {
            float t = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c)
              t += __builtin_bit_cast(float, af[c] << 16) + __builtin_bit_cast(float, af[c] & 0xffff0000u);
            dbs[a] += t;
          }
        }
      });
      }
      dma_wait(RING - 2);         // slab s + 1 (outstanding: s + 1 .. s + RING - 1)
      barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float *out = p.ws + static_cast<long long>(blockIdx.x) * (static_cast<long long>(N) * K + N);
#pragma unroll
  for (int a = 0; a < TN; ++a) {
    const int tn = wn0 + a;
    if (tn >= nt) continue;
#pragma unroll
    for (int b = 0; b < TK; ++b) {
      const int tkk = wk0 + b;
      if (tkk >= kt) continue;
      const int k = tkk * 16 + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = tn * 16 + 4 * g + r;
        if (n < N && k < K) out[static_cast<long long>(n) * K + k] = acc[a][b][r];
      }
    }
  }
  if (dbwave) {      // lane (i, g) holds column i's sum over the rows 8 g .. 8 g + 7 of every slab: add the four groups
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      float t = dbs[a];
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      const int n = (wn0 + a) * 16 + i;
      if (g == 0 && wn0 + a < nt && n < N) out[static_cast<long long>(N) * K + n] = t;
    }
  }
}

// row pitch of a slab image: row bytes + 16-byte granules of padding until pitch = 32 (mod 64) bytes: consecutive rows then
// start 8 dword banks apart, and the 8 (16) consecutive rows one transpose read of 32 (64) lanes touches, 32 bytes each,
// tile the 64 banks
static int tr_pitch(int width_elems) {
  int pitch = (width_elems * 2 + 15) / 16 * 16;      // (callers pass widths that are multiples of 8: whole granules)
  while (pitch % 64 != 32) pitch += 16;
  return pitch;
}

struct WgPlan {
  int grid, slabs_per_wg;
  size_t ws_floats;
};
WgPlan plan_wg16(long long M, int N, int K) {
  WgPlan pl;
  const long long slabs = (M + 31) >> 5;
  long long grid = nsdp::num_cus();
  if (grid > slabs / 4) grid = slabs / 4 > 0 ? slabs / 4 : 1;       // at least 4 slabs per workgroup (2: slower at B = 8)
  pl.slabs_per_wg = static_cast<int>((slabs + grid - 1) / grid);
  pl.grid = static_cast<int>((slabs + pl.slabs_per_wg - 1) / pl.slabs_per_wg);
  pl.ws_floats = static_cast<size_t>(pl.grid) * (static_cast<size_t>(N) * K + N);
  return pl;
}

// geometry of the transpose-read kernel for a shape: rows per slab (0: the shape stays on the transposing kernel)
struct TrGeom {
  int sr, pitch_a, pitch_b, slot_bytes, pieces, ring;
};
static TrGeom tr_geometry(long long M, int N, int K, bool mask) {
  TrGeom t{0, 0, 0, 0, 0, 0};
  if (N % 8 || K % 8 || (g_wg16_dbg & 8)) return t;
  t.pitch_a = tr_pitch(N);
  t.pitch_b = tr_pitch(K);
  // (two MFMA steps per barrier for the narrow layers -- SR = 2 -- measured slower in the step: 29.5 against 29.1 ms)
  for (int sr = 1; sr >= 1; --sr) {
    if (M < 2LL * 32 * sr) continue;
    const int img_a = (32 * sr * t.pitch_a + 1023) / 1024 * 1024, img_b = (32 * sr * t.pitch_b + 1023) / 1024 * 1024;
    const int slot = img_a * (mask ? 2 : 1) + img_b, pieces = slot / 1024, pw = (pieces + kWavesWg - 1) / kWavesWg;
    int ring = (160 * 1024) / slot;
    ring = ring > 6 ? 6 : ring;
    while (ring > 3 && (ring - 2) * pw > 16) --ring;
    if (ring < 3 || pw > 3) continue;
    t.sr = sr; t.slot_bytes = slot; t.pieces = pieces; t.ring = ring;
    return t;
  }
  return t;
}

WgPlan plan_wg16_any(long long M, int N, int K, bool mask) {      // partials are per workgroup: the plan follows the kernel
  const TrGeom t = tr_geometry(M, N, K, mask);
  if (t.sr == 0) return plan_wg16(M, N, K);
  WgPlan pl;
  const long long slabs = (M + 32 * t.sr - 1) / (32 * t.sr);
  long long grid = nsdp::num_cus();
  if (grid > slabs / 4) grid = slabs / 4 > 0 ? slabs / 4 : 1;
  pl.slabs_per_wg = static_cast<int>((slabs + grid - 1) / grid);
  pl.grid = static_cast<int>((slabs + pl.slabs_per_wg - 1) / pl.slabs_per_wg);
  pl.ws_floats = static_cast<size_t>(pl.grid) * (static_cast<size_t>(N) * K + N);
  return pl;
}

template <int TN, int TK>
int launch_wg16_tr(const Wg16Params &q, const TrGeom &t, int grid, hipStream_t st) {
  Wg16TrParams p{reinterpret_cast<const unsigned char *>(q.dY), reinterpret_cast<const unsigned char *>(q.X),
                 reinterpret_cast<const unsigned char *>(q.mask), q.ws, q.M, q.N, q.K, q.relu_x, q.want_db, q.slabs_per_wg,
                 q.dbg, t.ring, t.pitch_a, t.pitch_b, t.slot_bytes, t.pieces};
  const size_t lds = static_cast<size_t>(t.ring) * t.slot_bytes;
  auto go = [&](auto kern) -> int {
    if (lds > 64 * 1024)
      NSDP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds)));
    NSDP_TRACE("wgrad_bf16_tr<%d,%d,%s,%d>", TN, TK, p.mask ? "mask" : "plain", t.sr);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kWavesWg * 64), lds, st, p);
    return nsdp::launch_status("wgrad_bf16_tr_kernel");
  };
  if (p.mask) return go(wgrad_bf16_tr_kernel<TN, TK, true, 1>);
  return go(wgrad_bf16_tr_kernel<TN, TK, false, 1>);
}

template <int TN, int TK, int TASKS>
int launch_wg16(const Wg16Params &p_in, int grid, hipStream_t st) {
  const Wg16Params &p0 = p_in;
  const int tiles = ((p0.N + 15) >> 4) + ((p0.K + 15) >> 4);
  const int nt_ = (p0.N + 15) >> 4;
  const int pieces = tiles + (p0.mask ? nt_ : 0), pw = (pieces + kWavesWg - 1) / kWavesWg;
  int ring = (160 - 2 * tiles) / pieces;               // KiB: what fits next to the two fragment images
  ring = ring > 6 ? 6 : ring;
  while (ring > 3 && (ring - 2) * pw > 24) --ring;     // counted waits are available up to vmcnt(24)
  const size_t lds = static_cast<size_t>(2) * tiles * 1024 + static_cast<size_t>(ring) * pieces * 1024;
  Wg16Params p = p_in;
  p.ring = ring;
  if (ring < 3) {
    nsdp::set_error("linear_wgrad_bf16: N=%d, K=%d with a mask needs %zu bytes of LDS (pre-multiply the mask into dY)", p.N,
                    p.K, lds);
    return NSDP_ENOSUP;
  }
  auto go = [&](auto kern) -> int {
    if (lds > 64 * 1024)
      NSDP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds)));
    NSDP_TRACE("wgrad_bf16<%d,%d,%d,%s>", TN, TK, TASKS, p.mask ? "mask" : "plain");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kWavesWg * 64), lds, st, p);
    return nsdp::launch_status("wgrad_bf16_kernel");
  };
  if (p.mask) return go(wgrad_bf16_kernel<TN, TK, TASKS, 1>);
  return go(wgrad_bf16_kernel<TN, TK, TASKS, 0>);
}

}  // namespace

namespace nsdp {
void debug_set_wg16(int value) { g_wg16_dbg = value; }
void debug_set_lin16(int value) { g_lin16_dbg = value; }
}  // namespace nsdp

extern "C" {

long long nsdp_packed_weight_bf16_bytes(int N, int K, int transposed) {
  const long long blocks = transposed ? static_cast<long long>((N + 31) / 32) * ((K + 15) / 16)
                                      : static_cast<long long>((K + 31) / 32) * ((N + 15) / 16);
  return blocks * 1024;
}

int nsdp_pack_weights_bf16(const NsdpPackDesc *descs, int count, void *stream) {
  if (count <= 0) return 0;
  NSDP_REQUIRE(descs, "pack_weights_bf16: null descriptor array");
  hipStream_t st = nsdp::as_stream(stream);
  for (int base = 0; base < count; base += kPackBatch) {
    PackBatch b;
    const int n = count - base < kPackBatch ? count - base : kPackBatch;
    long long threads = 64;
    for (int i = 0; i < n; ++i) {
      const NsdpPackDesc &e = descs[base + i];
      NSDP_REQUIRE(e.W && (e.Wp || e.WpT) && e.N > 0 && e.K > 0, "pack_weights_bf16: bad descriptor %d", base + i);
      b.d[i] = PackOne{e.W, static_cast<u32x4 *>(e.Wp), static_cast<u32x4 *>(e.WpT), e.N, e.K};
      const long long t = pack_b16_threads(e.N, e.K, e.Wp != nullptr, e.WpT != nullptr);
      threads = t > threads ? t : threads;
    }
    for (int i = n; i < kPackBatch; ++i) b.d[i] = b.d[0];
    hipLaunchKernelGGL(pack_b16_kernel, dim3(static_cast<unsigned>((threads + 255) / 256), n), dim3(256), 0, st, b);
    const int rc = nsdp::launch_status("pack_b16_kernel");
    if (rc) return rc;
  }
  return 0;
}

int nsdp_linear_bf16(const void *X, const void *Wp, const float *bias, const void *residual, const void *mask,
                     const void *out_mask, void *Y, long long M, int N, int K, int relu_in, int relu_out, int out_f32,
                     void *stream) {
  if (M <= 0 || N <= 0) return 0;
  NSDP_REQUIRE(X && Wp && Y, "linear_bf16: null pointer");
  NSDP_REQUIRE(K >= 8 && K % 8 == 0 && K <= 256, "linear_bf16: K=%d must be a multiple of 8 in [8, 256]", K);
  NSDP_REQUIRE(N <= 256 && (out_f32 || N % 4 == 0), "linear_bf16: N=%d must be <= 256 (and a multiple of 4 for bf16 output)", N);
  NSDP_REQUIRE(!out_f32 || (!residual && !out_mask), "linear_bf16: fp32 output takes no residual / out_mask");
  NSDP_REQUIRE(((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Wp) | reinterpret_cast<uintptr_t>(Y) |
                 reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(mask) |
                 reinterpret_cast<uintptr_t>(out_mask)) & 15) == 0,
               "linear_bf16: all operands must be 16-byte aligned");
  B16Params p{static_cast<const unsigned short *>(X), static_cast<const u32x4 *>(Wp), bias,
              static_cast<const unsigned short *>(residual), static_cast<const unsigned short *>(mask),
              static_cast<const unsigned short *>(out_mask), Y, M, N, K, relu_in, relu_out, out_f32, g_lin16_dbg};
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kLinearB16, st, 2.0 * M * N * K,
                          2.0 * static_cast<double>(M) * (K * (mask ? 2 : 1) + N * (1 + (residual ? 1 : 0) + (out_mask ? 1 : 0))) +
                              2.0 * N * K);
  const int nt = (N + 15) / 16, kb = (K + 31) / 32;
  if (kb <= 4) {
    if (nt <= 4) return launch_lin<4, 4>(p, st);
    if (nt <= 8) return launch_lin<8, 4>(p, st);
    if (nt <= 13) return launch_lin<13, 4>(p, st);
    return launch_lin<16, 4>(p, st);
  }
  if (nt <= 4) return launch_lin<4, 8>(p, st);
  if (nt <= 8) return launch_lin<8, 8>(p, st);
  if (nt <= 13) return launch_lin<13, 8>(p, st);
  return launch_lin<16, 8>(p, st);
}

int nsdp_linear_wgrad_bf16_takes_mask(long long M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  const int nt = (N + 15) / 16, kt = (K + 15) / 16;
  if (tr_geometry(M, N, K, true).sr) return 1;          // transpose-read kernel: three images per ring slot
  return (160 - 2 * (nt + kt)) / (2 * nt + kt) >= 3;    // transposing kernel: 2 fragment images + >= 3 row images (KiB)
}

size_t nsdp_linear_wgrad_bf16_workspace_bytes(long long M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  // (with or without a mask: the larger of the two plans, the caller need not know)
  const size_t a = plan_wg16_any(M, N, K, false).ws_floats, b = plan_wg16_any(M, N, K, true).ws_floats;
  return (a > b ? a : b) * sizeof(float);
}

static int wgrad_bf16_impl(const void *dY, const void *X, const void *mask, int relu_x, float *dW, float *db, long long M, int N,
                           int K, int accumulate, float *workspace, size_t workspace_bytes, NsdpWgradB16ReduceDesc *desc_out,
                           void *stream);

int nsdp_linear_wgrad_bf16(const void *dY, const void *X, const void *mask, int relu_x, float *dW, float *db,
                           long long M, int N, int K, int accumulate, float *workspace, size_t workspace_bytes,
                           void *stream) {
  return wgrad_bf16_impl(dY, X, mask, relu_x, dW, db, M, N, K, accumulate, workspace, workspace_bytes, nullptr, stream);
}

int nsdp_linear_wgrad_bf16_partials(const void *dY, const void *X, const void *mask, int relu_x, float *dW, float *db,
                                    long long M, int N, int K, int accumulate, float *workspace, size_t workspace_bytes,
                                    NsdpWgradB16ReduceDesc *desc_out, void *stream) {
  NSDP_REQUIRE(desc_out, "linear_wgrad_bf16_partials: null descriptor");
  desc_out->ws = nullptr;
  return wgrad_bf16_impl(dY, X, mask, relu_x, dW, db, M, N, K, accumulate, workspace, workspace_bytes, desc_out, stream);
}

int nsdp_wgrad_bf16_reduce_batched(const NsdpWgradB16ReduceDesc *descs, int count, void *stream) {
  if (count <= 0) return 0;
  NSDP_REQUIRE(descs, "wgrad_bf16_reduce_batched: null descriptor array");
  hipStream_t st = nsdp::as_stream(stream);
  for (int base = 0; base < count; base += kReduceBatchB16) {
    ReduceBatchB16 b;
    const int n = count - base < kReduceBatchB16 ? count - base : kReduceBatchB16;
    long long blocks = 1;
    for (int i = 0; i < n; ++i) {
      const NsdpWgradB16ReduceDesc &e = descs[base + i];
      NSDP_REQUIRE(e.ws && e.dW && e.S > 0 && e.N > 0 && e.K > 0, "wgrad_bf16_reduce_batched: bad descriptor %d", base + i);
      b.d[i] = e;
      const long long ne = static_cast<long long>(e.N) * e.K + e.N;
      blocks = (ne + 255) / 256 > blocks ? (ne + 255) / 256 : blocks;
    }
    for (int i = n; i < kReduceBatchB16; ++i) b.d[i] = b.d[0];
    hipLaunchKernelGGL(reduce_b16_batched_kernel, dim3(static_cast<unsigned>(blocks), n), dim3(256), 0, st, b);
    const int rc = nsdp::launch_status("reduce_b16_batched_kernel");
    if (rc) return rc;
  }
  return 0;
}

static int wgrad_bf16_impl(const void *dY, const void *X, const void *mask, int relu_x, float *dW, float *db, long long M, int N,
                           int K, int accumulate, float *workspace, size_t workspace_bytes, NsdpWgradB16ReduceDesc *desc_out,
                           void *stream) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  NSDP_REQUIRE(dY && X && dW && workspace, "linear_wgrad_bf16: null pointer");
  NSDP_REQUIRE(N % 2 == 0 && K % 2 == 0 && N <= 256 && K <= 256, "linear_wgrad_bf16: N=%d, K=%d must be even and <= 256", N, K);
  const TrGeom geom = tr_geometry(M, N, K, mask != nullptr);
  const WgPlan pl = plan_wg16_any(M, N, K, mask != nullptr);
  NSDP_REQUIRE(workspace_bytes >= pl.ws_floats * sizeof(float), "linear_wgrad_bf16: workspace too small");
  Wg16Params p{static_cast<const unsigned short *>(dY), static_cast<const unsigned short *>(X),
               static_cast<const unsigned short *>(mask), workspace, M, N, K, relu_x, db != nullptr, pl.slabs_per_wg,
               g_wg16_dbg, 3};
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kWgradB16, st, 2.0 * M * N * K,
                          2.0 * static_cast<double>(M) * (N * (mask ? 2 : 1) + K));
  const int nt = (N + 15) / 16, kt = (K + 15) / 16;
  const int tn = (nt + 3) / 4, tk = (kt + 3) / 4;       // per-wave tile block (waves 4 x 4)
  // producer chunks (64 column pairs of one row group): 4 row groups x (chunks of dY + chunks of X) <= 16 = one per wave
  int rc;
  if (geom.sr) {      // operands by transpose reads straight from the slab images
    if (tn <= 2 && tk <= 2) rc = launch_wg16_tr<2, 2>(p, geom, pl.grid, st);
    else rc = launch_wg16_tr<4, 4>(p, geom, pl.grid, st);
  } else if (tn <= 2 && tk <= 2) rc = launch_wg16<2, 2, 1>(p, pl.grid, st);
  else rc = launch_wg16<4, 4, 1>(p, pl.grid, st);
  if (rc) return rc;
  if (desc_out) {      // the caller reduces this layer's partials with a batch (nsdp_wgrad_bf16_reduce_batched)
    *desc_out = NsdpWgradB16ReduceDesc{workspace, dW, db, pl.grid, N, K, accumulate ? 1 : 0, 0};
    return 0;
  }
  const long long nw = static_cast<long long>(N) * K, nb = db ? N : 0;
  hipLaunchKernelGGL(reduce_b16_kernel, dim3(static_cast<unsigned>((nw + N + 255) / 256)), dim3(256), 0, st, workspace,
                     pl.grid, nw + N, nw, dW, static_cast<long long>(nb), db, accumulate);
  return nsdp::launch_status("reduce_b16_kernel");
}

static WgPlan plan_onehot(int B, long long rows) {
  WgPlan pl;
  const long long slabs = (rows + 31) >> 5;
  long long grid = (2LL * nsdp::num_cus() + B - 1) / B;            // ~2 workgroups per CU over all shapes
  if (grid > slabs / 4) grid = slabs / 4 > 0 ? slabs / 4 : 1;
  pl.slabs_per_wg = static_cast<int>((slabs + grid - 1) / grid);
  pl.grid = static_cast<int>((slabs + pl.slabs_per_wg - 1) / pl.slabs_per_wg);
  pl.ws_floats = 0;
  return pl;
}

size_t nsdp_scatter_rows_onehot_bf16_workspace_bytes(int B, long long rows, int N, int d) {
  if (B <= 0 || rows <= 0) return 0;
  const int Np = (N + 1) & ~1;
  return static_cast<size_t>(B) * plan_onehot(B, rows).grid * (static_cast<size_t>(Np) * d + Np) * sizeof(float);
}

int nsdp_scatter_rows_onehot_bf16(const void *src, const int32_t *idx, int B, long long rows, int N, int d, float *table,
                                  float *workspace, size_t workspace_bytes, void *stream) {
  if (B <= 0 || rows <= 0 || N <= 0 || d <= 0) return 0;
  NSDP_REQUIRE(src && idx && table && workspace, "scatter_rows_onehot_bf16: null pointer");
  NSDP_REQUIRE(N <= 128 && N % 2 == 0 && d % 8 == 0 && d <= 256, "scatter_rows_onehot_bf16: need an even N <= 128 and d %% 8 == 0, d <= 256 (N=%d d=%d)", N, d);
  NSDP_REQUIRE(B <= 65535, "scatter_rows_onehot_bf16: batch too large");
  NSDP_REQUIRE(workspace_bytes >= nsdp_scatter_rows_onehot_bf16_workspace_bytes(B, rows, N, d), "scatter_rows_onehot_bf16: workspace too small");
  const WgPlan pl = plan_onehot(B, rows);
  Wg16Params p{reinterpret_cast<const unsigned short *>(idx), static_cast<const unsigned short *>(src), nullptr, workspace, rows,
               N, d, 0, 0, pl.slabs_per_wg, 0, 3};
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kAttnBwd, st, 0.0, static_cast<double>(B) * rows * (2.0 * d + 4.0));
  const int nt = (N + 15) >> 4, kt = (d + 15) >> 4, tiles = nt + kt;
  int ring = (160 - 2 * tiles) / kt;
  ring = ring > 6 ? 6 : ring;
  p.ring = ring;
  const size_t lds = static_cast<size_t>(2) * tiles * 1024 + static_cast<size_t>(ring) * kt * 1024;
  auto kern = wgrad_bf16_kernel<2, 4, 1, 2>;              // n tiles <= 8 (4 x 2), k tiles <= 16 (4 x 4), one chunk per wave
  if (lds > 64 * 1024)
    NSDP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(lds)));
  NSDP_TRACE("scatter_rows_onehot_bf16");
  hipLaunchKernelGGL(kern, dim3(pl.grid, B), dim3(kWavesWg * 64), lds, st, p);
  int rc = nsdp::launch_status("wgrad_bf16_kernel<onehot>");
  if (rc) return rc;
  const long long nw = static_cast<long long>(N) * d;
  hipLaunchKernelGGL(reduce_tables_kernel, dim3(static_cast<unsigned>((nw + 255) / 256), B), dim3(256), 0, st, workspace,
                     pl.grid, nw + N, nw, table);
  return nsdp::launch_status("reduce_tables_kernel");
}

}  // extern "C"
