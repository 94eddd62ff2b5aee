// fp32 dense layers on the CDNA4 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, 157 TF peak).
//
// Every dense layer of the TDNet hot path is "tall-skinny": rows M = B*n*k is huge (10^5..10^6) while
// N, K <= 256 (120 / 128 / 200 / 256).  A generic square-tile LDS GEMM is the wrong shape for that, so:
//
//   nsdp_linear_f32        Y[M,N] = act( pre(X)[M,K] * W[N,K]^T + b ) (+ residual)
//       one wave owns MT*16 complete rows and ALL N columns: X is read exactly once (HBM), the
//       <=256 KiB weight matrix is re-read per wave from L1/L2.  Both operands go global -> VGPR as
//       float4 (16 B/lane) with a k-permuted fragment convention (lane group g supplies k = 16*kb+4g+s
//       at MFMA step s for A and B alike), so no LDS staging/transposition is needed at all and a wave
//       never waits on a barrier.  Prologue fusions: relu(X), X * (mask > 0) (ReLU backward);
//       epilogue fusions: bias, ReLU, residual add.
//   nsdp_linear_wgrad_f32  dW[N,K] = pre(dY)[M,N]^T * X[M,K],  db[N] = colsum(pre(dY))
//       split over row chunks (deterministic two-stage reduction through a workspace, no atomics);
//       rows are the MFMA k dimension, so both operands are read in their natural row-major layout.
//
// Activations stay fp32 end to end: the north-star parity bar (1e-4 L2 vs the CPU reference) leaves no
// room for bf16 inputs, and gfx950 has no TF32/xf32 path.
#include <type_traits>

#include <stdlib.h>

#include "common.h"
#include "k4.h"
#include "pack_bodies.h"
#include "prof.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

int g_nt_dbg = 0;
int g_nt_pipe = 1;  // nsdp_debug_set(3, v): 1 = fenced two-buffer software pipeline, 0 = register-lean loop

struct LinearParams {
  const float *X, *W, *bias, *residual, *mask, *out_mask;
  float *Y;
  int dbg;  // experiment knob (nsdp_debug_set(4, v)): 1 = skip epilogue stores, 2 = A rows all the same (L2-hot X)
  long long M;
  int N, K;
  int relu_in, relu_out;
};

// PRE: 0 = plain X, 1 = X * (mask > 0), 2 = relu(X).  All global loads are UNCONDITIONAL (indices are
// clamped into the tensor instead of predicated): hipcc turns a per-load predicate into a branch plus
// s_waitcnt vmcnt(0), which serialises the whole operand stream.  Rows >= M and columns >= N compute
// garbage that is never stored; the ragged last k-block (K % 16 != 0) is handled by zeroing the weight
// fragment with a select while the activation fragment re-reads in-row (finite) data.
// WP: W is the fragment-major pack written by nsdp_pack_weight_f32 ([n tile][k block][lane][4], zero padded):
// every wave-wide weight load is one contiguous KiB instead of 16 rows x 64 B, and needs no clamps or k-tail fix-up.
// DEEP > 0: ring of DEEP operand buffers instead of two.  The small-M split-N launch (MT = 1, NT = 4) has only
// 16 MFMAs (~0.2 us) per k block to hide an L2 round trip (~0.6 us) behind: with two buffers every k block stalled
// (13.9 us for 3200 x 256 x 256, 3.4 us of it MFMA time); its 20 operand registers per block allow eight in flight.
template <int MT, int NT, int PRE, bool PIPE, bool WP = false, int DEEP = 0>
__global__ __launch_bounds__(256) void linear_nt_kernel(LinearParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const long long row0 = (static_cast<long long>(blockIdx.x) * 4 + wave) * (MT * 16);
  if (row0 >= p.M) return;  // whole wave out of range (no barriers in this kernel)
  const int K = p.K, N = p.N;
  const int ntile0 = blockIdx.y * NT;  // column-tile offset (grid.y > 1 only for the small-M split-N launch)

  f32x4 acc[MT][NT];
  if (p.residual) {  // residual add fused as the accumulator's initial value (loads overlap the k loop)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        int col = (ntile0 + nt) * 16 + li;
        col = col < N ? col : (N - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          long long row = row0 + mt * 16 + g * 4 + r;
          row = row < p.M ? row : (p.M - 1);
          acc[mt][nt][r] = p.residual[row * N + col];
        }
      }
  } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const float *xa[MT];
  const float *ma[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    long long r = row0 + mt * 16 + li;
    r = r < p.M ? r : (p.M - 1);
    if (p.dbg == 2) r = r & 4095;
    xa[mt] = p.X + r * K;
    ma[mt] = PRE == 1 ? p.mask + r * K : nullptr;
  }
  const int KB = (K + 15) >> 4;
  const float *wb[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    if (WP) {
      int t = ntile0 + nt;
      t = t * 16 < N ? t : ((N + 15) / 16 - 1);
      wb[nt] = p.W + (static_cast<long long>(t) * KB * 64 + lane) * 4;
    } else {
      int n = (ntile0 + nt) * 16 + li;
      n = n < N ? n : (N - 1);
      wb[nt] = p.W + static_cast<long long>(n) * K;
    }
  }

  // Software pipeline, two register buffers.  Phase = { issue the raw loads of block kb+1 ; MFMAs of
  // block kb }.  hipcc's scheduler otherwise sinks the prefetch loads down to their first use (minimum
  // register pressure) and the wave then sits in s_waitcnt for a full L2/HBM round trip per k-block,
  // so the phases are fenced with sched_barrier and the loads are interleaved with the first MFMAs by
  // sched_group_barrier.  The operand fix-ups (k-tail zeroing, ReLU / mask prologue) run at the START
  // of the phase that consumes the buffer, a whole MFMA phase after its loads were issued.
  struct Frag {
    float4 a[MT];
    float4 m[PRE == 1 ? MT : 1];
    float4 b[NT];
  };
  auto issue = [&](int kb, Frag &f) {
    int ko = kb * 16 + 4 * g;
    ko = ko < K ? ko : (K - 4);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f.a[mt] = *reinterpret_cast<const float4 *>(xa[mt] + ko);
      if (PRE == 1) f.m[mt] = *reinterpret_cast<const float4 *>(ma[mt] + ko);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) f.b[nt] = *reinterpret_cast<const float4 *>(wb[nt] + (WP ? kb * 256 : ko));
  };
  auto fixup = [&](int kb, Frag &f) {
    const bool kv = kb * 16 + 4 * g < K;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float4 v = f.a[mt];
      if (PRE == 1) {
        const float4 m = f.m[mt];
        v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
        v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
      }
      if (PRE == 2) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      f.a[mt] = v;
    }
    if (!WP) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float4 v = f.b[nt];
        v.x = kv ? v.x : 0.f; v.y = kv ? v.y : 0.f; v.z = kv ? v.z : 0.f; v.w = kv ? v.w : 0.f;
        f.b[nt] = v;
      }
    }
  };
  auto mma = [&](const Frag &f) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[mt].x, f.b[nt].x, acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[mt].y, f.b[nt].y, acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[mt].z, f.b[nt].z, acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[mt].w, f.b[nt].w, acc[mt][nt], 0, 0, 0);
  };
  constexpr int kLoads = MT * (PRE == 1 ? 2 : 1) + NT;
  auto interleave = [&]() {  // 1 VMEM read, then 4 MFMAs, kLoads times; the rest of the MFMAs follow
#pragma unroll
    for (int i = 0; i < kLoads; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
  };

  if constexpr (DEEP > 0) {
    Frag f[DEEP];
#pragma unroll
    for (int d = 0; d < DEEP; ++d) issue(d < KB ? d : KB - 1, f[d]);
    __builtin_amdgcn_sched_barrier(0);
    // KB % DEEP == 0 (checked at launch): straight-line phases, so that the compiler's s_waitcnt counts stay exact
    // (a conditional phase made it drain vmcnt(0) at every loop header)
    for (int kb = 0; kb < KB; kb += DEEP) {
#pragma unroll
      for (int d = 0; d < DEEP; ++d) {
        fixup(kb + d, f[d]);
        mma(f[d]);
        __builtin_amdgcn_sched_barrier(0);
        const int nx = kb + d + DEEP;
        issue(nx < KB ? nx : KB - 1, f[d]);   // past the end: re-loads a valid block that is never used
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else if (PIPE) {
    Frag f0, f1;
    issue(0, f0);
    __builtin_amdgcn_sched_barrier(0);
    int kb = 0;
    for (; kb + 2 <= KB; kb += 2) {
      fixup(kb, f0);
      issue(kb + 1, f1);
      mma(f0);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      fixup(kb + 1, f1);
      // kb + 2 == KB re-loads a valid block whose result is never used: keeps the loop branch-free
      issue(kb + 2 < KB ? kb + 2 : kb, f0);
      mma(f1);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kb < KB) {
      fixup(kb, f0);
      mma(f0);
    }
  } else {  // register-lean form: no explicit prefetch, two waves per SIMD overlap each other's loads
    for (int kb = 0; kb < KB; ++kb) {
      Frag f;
      issue(kb, f);
      fixup(kb, f);
      mma(f);
    }
  }

  // epilogue: C/D layout col = lane & 15, row = (lane >> 4) * 4 + reg.  Loads are unconditional from
  // clamped indices (see above).  Fast path (wave-uniform): all MT*16 rows in range -> the stores of
  // every complete 16-column tile are unconditional straight-line code; only the ragged last tile
  // (N % 16 != 0) and the last row block of the matrix take the predicated form.
  const bool full_rows = row0 + MT * 16 <= p.M;
  // all bias values up front: one L2 round trip instead of one per n tile (each tile below is its own basic block)
  float bias_t[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bias_t[nt] = 0.f;
  if (p.bias) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int c = (ntile0 + nt) * 16 + li;
      bias_t[nt] = p.bias[c < N ? c : (N - 1)];
    }
  }
  auto tile = [&](int nt, auto has_omask, auto guarded) {
    const int col = (ntile0 + nt) * 16 + li;
    const bool cv = col < N;
    const int colc = cv ? col : (N - 1);
    const float bv = bias_t[nt];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long row = row0 + mt * 16 + g * 4 + r;
        const bool rv = !decltype(guarded)::value || row < p.M;
        const long long rowc = rv ? row : (p.M - 1);
        float v = acc[mt][nt][r] + bv;
        if (p.relu_out) v = fmaxf(v, 0.f);
        if (decltype(has_omask)::value) v = p.out_mask[rowc * N + colc] > 0.f ? v : 0.f;
        if (decltype(guarded)::value) {
          if (cv && rv && p.dbg != 1) p.Y[rowc * N + colc] = v;
        } else {
          p.Y[row * N + col] = v;
        }
      }
    }
  };
  auto epilogue = [&](auto has_omask) {
    const int full_tiles = N >> 4;  // tiles whose 16 columns are all valid
    if (full_rows && p.dbg != 1) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (ntile0 + nt < full_tiles) tile(nt, has_omask, std::false_type{});
        else if ((ntile0 + nt) * 16 < N) tile(nt, has_omask, std::true_type{});
      }
    } else {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        if ((ntile0 + nt) * 16 < N) tile(nt, has_omask, std::true_type{});
    }
  };
  if (p.out_mask) epilogue(std::true_type{});
  else epilogue(std::false_type{});
}

// ------------------------------------------------------------------------------------------------
// LDS-staged variant: the weight k-block (NT KiB) is DMA'd into LDS once per workgroup
// (global_load_lds_dwordx4, one 1 KiB piece per wave-instruction, lane-linear = exactly the MFMA
// fragment order) instead of being re-loaded into registers by each of the 4 waves, and the
// activations take the same route through a wave-private LDS slot.  A wave then issues 2 + NT/4 DMA
// pieces per k-block instead of 2 + NT register loads (VMEM issue competes with MFMA issue inside a
// wave), holds only one B fragment at a time (two workgroups per CU fit), and the loop is the
// two-buffer STAGE -> ds_read + MFMA -> vmcnt(0) + barrier structure.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

template <int MT, int NT, int PRE>
__global__ __launch_bounds__(256) void linear_nt_lds_kernel(LinearParams p) {
  constexpr int kASlots = (PRE == 1 ? 2 : 1) * MT;          // activation (+ mask) tiles per wave
  __shared__ __attribute__((aligned(16))) float lds[2 * (NT + 4 * kASlots) * 256];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4;
  const long long row0 = (static_cast<long long>(blockIdx.x) * 4 + wave) * (MT * 16);
  const int K = p.K, N = p.N;
  const int KB = (K + 15) >> 4;
  const bool ragged_k = (K & 15) != 0;
  constexpr int kBufFloats = (NT + 4 * kASlots) * 256;
  float *bslot = lds;                                        // [buf][NT][256]
  float *aslot = lds + NT * 256 + wave * kASlots * 256;      // [buf][wave][kASlots][256]

  f32x4 acc[MT][NT];
  if (p.residual) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        int col = nt * 16 + li;
        col = col < N ? col : (N - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          long long row = row0 + mt * 16 + g * 4 + r;
          row = row < p.M ? row : (p.M - 1);
          acc[mt][nt][r] = p.residual[row * N + col];
        }
      }
  } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // per-lane global source rows (clamped into the tensors: out-of-range rows/columns are never stored)
  const float *xa[MT];
  const float *ma[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    long long r = row0 + mt * 16 + li;
    r = r < p.M ? r : (p.M - 1);
    xa[mt] = p.X + r * K;
    ma[mt] = PRE == 1 ? p.mask + r * K : nullptr;
  }
  constexpr int kMyB = (NT + 3) / 4;                          // weight tiles DMA'd by this wave
  const float *wb[kMyB];
#pragma unroll
  for (int t = 0; t < kMyB; ++t) {
    int nt = wave + 4 * t;
    nt = nt < NT ? nt : (NT - 1);
    int n = nt * 16 + li;
    n = n < N ? n : (N - 1);
    wb[t] = p.W + static_cast<long long>(n) * K;
  }

  auto stage = [&](int kb, int buf) {
    int ko = kb * 16 + 4 * g;
    ko = ko < K ? ko : (K - 4);
    float *bb = bslot + buf * kBufFloats;
    float *ab = aslot + buf * kBufFloats;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(xa[mt] + ko), (lds_ptr_t)(ab + mt * 256), 16, 0, 0);
      if (PRE == 1)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(ma[mt] + ko), (lds_ptr_t)(ab + (MT + mt) * 256), 16, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < kMyB; ++t) {
      const int nt = wave + 4 * t;
      if (nt < NT)  // wave-uniform
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wb[t] + ko), (lds_ptr_t)(bb + nt * 256), 16, 0, 0);
    }
  };
  auto compute = [&](int kb, int buf) {
    const float *bb = bslot + buf * kBufFloats;
    const float *ab = aslot + buf * kBufFloats;
    float4 a[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float4 v = *reinterpret_cast<const float4 *>(ab + mt * 256 + lane * 4);
      if (PRE == 1) {
        const float4 m = *reinterpret_cast<const float4 *>(ab + (MT + mt) * 256 + lane * 4);
        v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
        v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
      }
      if (PRE == 2) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      a[mt] = v;
    }
    const bool zero_tail = ragged_k && kb == KB - 1 && (kb * 16 + 4 * g >= K);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float4 b = *reinterpret_cast<const float4 *>(bb + nt * 256 + lane * 4);
      if (ragged_k) {  // uniform; the select only bites in the last block
        b.x = zero_tail ? 0.f : b.x; b.y = zero_tail ? 0.f : b.y;
        b.z = zero_tail ? 0.f : b.z; b.w = zero_tail ? 0.f : b.w;
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, b.x, acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, b.y, acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, b.z, acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, b.w, acc[mt][nt], 0, 0, 0);
    }
  };

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kb = 0; kb < KB; ++kb) {
    const int cur = kb & 1;
    if (kb + 1 < KB) stage(kb + 1, cur ^ 1);   // DMA of the next block runs under this block's MFMAs
    compute(kb, cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  if (row0 >= p.M) return;  // (after the last barrier)
  const bool full_rows = row0 + MT * 16 <= p.M;
  auto tile = [&](int nt, auto has_omask, auto guarded) {
    const int col = nt * 16 + li;
    const bool cv = col < N;
    const int colc = cv ? col : (N - 1);
    const float bv = p.bias ? p.bias[colc] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long row = row0 + mt * 16 + g * 4 + r;
        const bool rv = !decltype(guarded)::value || row < p.M;
        const long long rowc = rv ? row : (p.M - 1);
        float v = acc[mt][nt][r] + bv;
        if (p.relu_out) v = fmaxf(v, 0.f);
        if (decltype(has_omask)::value) v = p.out_mask[rowc * N + colc] > 0.f ? v : 0.f;
        if (decltype(guarded)::value) {
          if (cv && rv && p.dbg != 1) p.Y[rowc * N + colc] = v;
        } else {
          p.Y[row * N + col] = v;
        }
      }
    }
  };
  auto epilogue = [&](auto has_omask) {
    const int full_tiles = N >> 4;  // tiles whose 16 columns are all valid
    if (full_rows && p.dbg != 1) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (nt < full_tiles) tile(nt, has_omask, std::false_type{});
        else if (nt * 16 < N) tile(nt, has_omask, std::true_type{});
      }
    } else {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        if (nt * 16 < N) tile(nt, has_omask, std::true_type{});
    }
  };
  if (p.out_mask) epilogue(std::true_type{});
  else epilogue(std::false_type{});
}

// K = 4 forward (first layer of a position-encoding MLP: 3-d relative coordinates, zero-padded): 8 flops per output
// float -- a pure output stream.  N/4 lanes per row, each with the 4 x 4 weights and the bias of its four channels in
// registers, one 16-byte load of the row's coordinates and one float4 store per row.  WP selects where a weight
// row lives (fragment-major pack: row n of the only k block is float4 number (n / 16) * 64 + n % 16).
// pre-activation of a K = 4 layer: ONE expression for the forward kernel and for the weight-gradient kernel that
// recomputes the layer's ReLU mask from its 16-byte input rows instead of reading the [M, N] output back
using nsdp::k4_preact_n;      // (k4.h: the rounding is pinned per output channel)

template <bool WP>
__global__ __launch_bounds__(256) void linear_k4_fwd_kernel(LinearParams p) {
  const int N = p.N;
  const int lpr = N >> 2, slots = 256 / lpr;
  const int sub = threadIdx.x / lpr, cq = threadIdx.x - sub * lpr;
  if (sub >= slots) return;
  float4 w[4], b = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int n = 4 * cq + c;
    w[c] = *reinterpret_cast<const float4 *>(p.W + (WP ? (static_cast<long long>(n >> 4) * 64 + (n & 15)) * 4
                                                       : static_cast<long long>(n) * 4));
  }
  if (p.bias) b = *reinterpret_cast<const float4 *>(p.bias + 4 * cq);
  const long long stride = static_cast<long long>(gridDim.x) * slots;
  constexpr int U = 4;
  for (long long r = static_cast<long long>(blockIdx.x) * slots + sub; r < p.M; r += U * stride) {
    float4 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long rr = r + u * stride;
      x[u] = *reinterpret_cast<const float4 *>(p.X + (rr < p.M ? rr : p.M - 1) * 4);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long rr = r + u * stride;
      float4 xv = x[u];
      if (p.relu_in) { xv.x = fmaxf(xv.x, 0.f); xv.y = fmaxf(xv.y, 0.f); xv.z = fmaxf(xv.z, 0.f); xv.w = fmaxf(xv.w, 0.f); }
      float4 y;
      y.x = k4_preact_n(xv, w[0], b.x, 0);      // (channel 4 cq + c: its parity is c's)
      y.y = k4_preact_n(xv, w[1], b.y, 1);
      y.z = k4_preact_n(xv, w[2], b.z, 0);
      y.w = k4_preact_n(xv, w[3], b.w, 1);
      if (p.relu_out) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
      if (rr < p.M) *reinterpret_cast<float4 *>(p.Y + rr * N + 4 * cq) = y;
    }
  }
}

template <int MT, int NT>
int launch_nt(const LinearParams &p, hipStream_t st, int grid_y = 1, bool wp = false) {
  const int pre = p.mask ? 1 : (p.relu_in ? 2 : 0);
  const long long rows_per_wg = 4LL * MT * 16;
  const long long grid = (p.M + rows_per_wg - 1) / rows_per_wg;
  nsdp::prof::Scope scope(nsdp::prof::kLinear, st, 2.0 * p.M * p.N * p.K,
                          4.0 * (static_cast<double>(p.M) * (p.K + p.N) + static_cast<double>(p.N) * p.K));
  const dim3 gr(static_cast<unsigned>(grid), grid_y);
  // small-M split-N launch with at most one wave per SIMD: nothing else hides the operand latency -> deep operand
  // ring (with more waves per SIMD its 192 registers cost more occupancy than the ring wins: 16000 rows, 29 -> 40 us)
  if (wp && MT == 1 && NT == 4 && ((p.K + 15) / 16) % 8 == 0 && grid * grid_y * 4 <= 4LL * nsdp::num_cus()) {
    constexpr int kDeep = (MT == 1 && NT == 4) ? 8 : 0;
    if (pre == 0) hipLaunchKernelGGL((linear_nt_kernel<MT, NT, 0, true, true, kDeep>), gr, dim3(256), 0, st, p);
    else if (pre == 1) hipLaunchKernelGGL((linear_nt_kernel<MT, NT, 1, true, true, kDeep>), gr, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((linear_nt_kernel<MT, NT, 2, true, true, kDeep>), gr, dim3(256), 0, st, p);
  } else if (wp) {
    if (pre == 0) hipLaunchKernelGGL((linear_nt_kernel<MT, NT, 0, true, true>), gr, dim3(256), 0, st, p);
    else if (pre == 1) hipLaunchKernelGGL((linear_nt_kernel<MT, NT, 1, true, true>), gr, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((linear_nt_kernel<MT, NT, 2, true, true>), gr, dim3(256), 0, st, p);
  } else if (g_nt_pipe == 2 && grid_y == 1) {
    if (pre == 0) hipLaunchKernelGGL((linear_nt_lds_kernel<MT, NT, 0>), gr, dim3(256), 0, st, p);
    else if (pre == 1) hipLaunchKernelGGL((linear_nt_lds_kernel<MT, NT, 1>), gr, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((linear_nt_lds_kernel<MT, NT, 2>), gr, dim3(256), 0, st, p);
  } else if (g_nt_pipe) {
    if (pre == 0) hipLaunchKernelGGL((linear_nt_kernel<MT, NT, 0, true>), gr, dim3(256), 0, st, p);
    else if (pre == 1) hipLaunchKernelGGL((linear_nt_kernel<MT, NT, 1, true>), gr, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((linear_nt_kernel<MT, NT, 2, true>), gr, dim3(256), 0, st, p);
  } else {
    if (pre == 0) hipLaunchKernelGGL((linear_nt_kernel<MT, NT, 0, false>), gr, dim3(256), 0, st, p);
    else if (pre == 1) hipLaunchKernelGGL((linear_nt_kernel<MT, NT, 1, false>), gr, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((linear_nt_kernel<MT, NT, 2, false>), gr, dim3(256), 0, st, p);
  }
  return nsdp::launch_status("linear_nt_kernel");
}

// ------------------------------------------------------------------------------------------------
// weight gradient: dW[N,K] = pre(dY)^T X over a chunk of rows per workgroup
// ------------------------------------------------------------------------------------------------
struct WgradParams {
  const float *dY, *X, *mask;
  int relu_x;
  float *ws;      // [S][N*K + N] partials
  long long M;
  int N, K;
  long long rows_per_chunk;  // multiple of 16
  int want_db;
  const float *W0 = nullptr, *b0 = nullptr;   // K = 4 only: recompute the ReLU mask (W0 row-major [N][4], b0 [N] or null)
};

// kWgN = n-tiles per wave (the workgroup's 4 waves cover 4*kWgN n-tiles >= N/16), TK = k-tiles per wave
// (grid.y covers K).  Rows are the MFMA reduction dimension: lane (li, g) feeds row 16*blk + 4g + s at
// step s to both operands, so dY and X are read in their natural row-major layout (64-B segments).
// Loads are unconditional from clamped indices and software-pipelined one 16-row block ahead (same
// reasons as linear_nt_kernel); rows past the chunk end are zeroed by a select on the dY operand.
template <int kWgN, int TK, bool PIPE>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(WgradParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int N = p.N, K = p.K;
  if (wave * kWgN * 16 >= N) return;  // idle wave (no barriers / LDS in this kernel)
  const int ktile0 = blockIdx.y * TK;
  const long long m_begin = static_cast<long long>(blockIdx.x) * p.rows_per_chunk;
  const long long m_end = min(p.M, m_begin + p.rows_per_chunk);

  f32x4 acc[kWgN][TK];
#pragma unroll
  for (int a = 0; a < kWgN; ++a)
#pragma unroll
    for (int b = 0; b < TK; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbsum[kWgN];
#pragma unroll
  for (int a = 0; a < kWgN; ++a) dbsum[a] = 0.f;

  int ncol[kWgN];  // clamped column indices: out-of-range tiles compute garbage that is never stored
#pragma unroll
  for (int a = 0; a < kWgN; ++a) {
    const int c = (wave * kWgN + a) * 16 + li;
    ncol[a] = c < N ? c : (N - 1);
  }
  int kcol[TK];
#pragma unroll
  for (int b = 0; b < TK; ++b) {
    const int c = (ktile0 + b) * 16 + li;
    kcol[b] = c < K ? c : (K - 1);
  }

  struct Frag {
    float a[4][kWgN];
    float m[4][kWgN];
    float b[4][TK];
  };
  const bool has_mask = p.mask != nullptr;
  auto issue = [&](long long mb, Frag &f) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      long long m = mb + 4 * g + s;
      m = m < p.M ? m : (p.M - 1);
#pragma unroll
      for (int a = 0; a < kWgN; ++a) {
        f.a[s][a] = p.dY[m * N + ncol[a]];
        if (has_mask) f.m[s][a] = p.mask[m * N + ncol[a]];
      }
#pragma unroll
      for (int b = 0; b < TK; ++b) f.b[s][b] = p.X[m * K + kcol[b]];
    }
  };
  auto fixup = [&](long long mb, Frag &f) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bool mv = mb + 4 * g + s < m_end;
#pragma unroll
      for (int a = 0; a < kWgN; ++a) {
        float v = f.a[s][a];
        if (has_mask) v = f.m[s][a] > 0.f ? v : 0.f;
        f.a[s][a] = mv ? v : 0.f;
      }
      if (p.relu_x) {
#pragma unroll
        for (int b = 0; b < TK; ++b) f.b[s][b] = fmaxf(f.b[s][b], 0.f);
      }
    }
  };
  auto mma = [&](const Frag &f) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int a = 0; a < kWgN; ++a) {
        dbsum[a] += f.a[s][a];
#pragma unroll
        for (int b = 0; b < TK; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[s][a], f.b[s][b], acc[a][b], 0, 0, 0);
      }
    }
  };

  if (PIPE) {
    Frag f0, f1;
    issue(m_begin, f0);
    __builtin_amdgcn_sched_barrier(0);
    long long mb = m_begin;
    for (; mb + 32 <= m_end; mb += 32) {
      fixup(mb, f0);
      issue(mb + 16, f1);
      mma(f0);
      __builtin_amdgcn_sched_barrier(0);
      fixup(mb + 16, f1);
      issue(mb + 32, f0);  // may run past the chunk: clamped to a valid row, zeroed by fixup or unused
      mma(f1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (mb < m_end) {  // one or two trailing 16-row blocks
      fixup(mb, f0);
      if (mb + 16 < m_end) issue(mb + 16, f1);
      mma(f0);
      __builtin_amdgcn_sched_barrier(0);
      if (mb + 16 < m_end) {
        fixup(mb + 16, f1);
        mma(f1);
      }
    }
  } else {  // register-lean form: two waves per SIMD hide the load latency instead
    for (long long mb = m_begin; mb < m_end; mb += 16) {
      Frag f;
      issue(mb, f);
      fixup(mb, f);
      mma(f);
    }
  }

  float *out = p.ws + static_cast<long long>(blockIdx.x) * (static_cast<long long>(N) * K + N);
#pragma unroll
  for (int a = 0; a < kWgN; ++a) {
#pragma unroll
    for (int b = 0; b < TK; ++b) {
      const int kc = (ktile0 + b) * 16 + li;  // D col = lane & 15 -> k index
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = (wave * kWgN + a) * 16 + g * 4 + r;  // D row -> n index
        if (n < N && kc < K) out[static_cast<long long>(n) * K + kc] = acc[a][b][r];
      }
    }
  }
  if (p.want_db && blockIdx.y == 0) {
#pragma unroll
    for (int a = 0; a < kWgN; ++a) {
      float v = dbsum[a];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      const int n = (wave * kWgN + a) * 16 + li;
      if (g == 0 && n < N) out[static_cast<long long>(N) * K + n] = v;
    }
  }
}

// Output-stationary weight gradient for FEW rows (the 100-anchor level of an 8-shape batch: 800 rows, 35 layers per step).  The
// row-split kernels above give every workgroup a chunk of rows and the WHOLE [N, K] output: at 800 rows that is 25 workgroups
// writing 256 KiB of partial sums each (13 us of a 26 us launch at one CU's share of the HBM write rate) for a second kernel to
// add up.  Here a workgroup owns a 32 x 32 block of dW and walks ALL rows: its sixteen waves take the 32-row blocks round-robin
// (operands straight from L2: the two tensors are a few hundred KiB), their accumulators meet in LDS in fixed order, and the
// block is written once -- no partials, no reduce launch, deterministic.  Exact-fp32 MFMA like the kernels it replaces.
// Lane (li, g) feeds rows 4 s + g of a block to both operands (v_mfma_f32_16x16x4_f32: A[i = li][g], B[g][j = li]).
template <bool MASK>
__global__ __launch_bounds__(1024) void linear_wgrad_direct_kernel(WgradParams p, float *__restrict__ dW, float *__restrict__ db,
                                                                   int accumulate) {
  constexpr int WV = 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int N = p.N, K = p.K;
  const int n0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  int na[2], kb[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {      // columns past the edge re-read the last one (their results are never stored)
    na[t] = n0 + 16 * t + li < N ? n0 + 16 * t + li : N - 1;
    kb[t] = k0 + 16 * t + li < K ? k0 + 16 * t + li : K - 1;
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbs[2] = {0.f, 0.f};
  const long long nblk = (p.M + 31) >> 5;
  // no software pipeline: sixteen waves (four per SIMD) each take whole 32-row blocks -- 32 independent loads, then 32 MFMAs --
  // and hide each other's L2 round trips; at 800 rows a wave sees one or two blocks
  for (long long blk = wave; blk < nblk; blk += WV) {
    float a[2][8], x[2][8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const long long r = blk * 32 + 4 * s + g;
      const bool rv = r < p.M;
      const long long rc = rv ? r : p.M - 1;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float v = p.dY[rc * N + na[t]];
        if constexpr (MASK) v = p.mask[rc * N + na[t]] > 0.f ? v : 0.f;
        a[t][s] = rv ? v : 0.f;
        const float xv = p.X[rc * K + kb[t]];
        x[t][s] = p.relu_x ? fmaxf(xv, 0.f) : xv;
      }
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
      for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta][s], x[tb][s], acc[ta][tb], 0, 0, 0);
      dbs[0] += a[0][s];
      dbs[1] += a[1][s];
    }
  }
  __shared__ f32x4 red[WV][4][64];
  __shared__ float red_db[WV][32];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) red[wave][a * 2 + b][lane] = acc[a][b];
  if (p.want_db && blockIdx.y == 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float v = dbs[t];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (g == 0) red_db[wave][16 * t + li] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x < 256) {      // D layout: lane (li, g) of tile (a, b) holds dW[n0 + 16 a + 4 g + r][k0 + 16 b + li], r = 0..3
    const int tt = threadIdx.x >> 6, l = threadIdx.x & 63;
    f32x4 t = red[0][tt][l];
#pragma unroll
    for (int w = 1; w < WV; ++w) t += red[w][tt][l];      // fixed order: deterministic
    const int k = k0 + 16 * (tt & 1) + (l & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 16 * (tt >> 1) + 4 * (l >> 4) + r;
      if (n < N && k < K) {
        float *o = dW + static_cast<long long>(n) * K + k;
        *o = accumulate ? *o + t[r] : t[r];
      }
    }
  } else if (threadIdx.x < 288 && p.want_db && blockIdx.y == 0) {
    const int c = threadIdx.x - 256;
    float t = red_db[0][c];
#pragma unroll
    for (int w = 1; w < WV; ++w) t += red_db[w][c];
    if (n0 + c < N) db[n0 + c] = accumulate ? db[n0 + c] + t : t;
  }
}
// rows up to which the output-stationary form runs (NSDP_WGRAD_DIRECT_ROWS, 0 = never; read once)
inline long long wgrad_direct_rows() {
  static const long long v = getenv("NSDP_WGRAD_DIRECT_ROWS") ? atoll(getenv("NSDP_WGRAD_DIRECT_ROWS")) : 2048;
  return v;
}
inline bool wgrad_direct_ok(long long M, int N, int K) { return M > 0 && M <= wgrad_direct_rows() && K > 4 && N >= 16 && K >= 16; }

int g_wgrad_vec4 = 1;  // nsdp_debug_set(5, v): 1 = float4-operand weight-gradient kernel (default), 0 = dword form
int g_wgrad_pipe = 0;  // measured on MI355X: the register-lean form wins at every layer shape of the path
                       // (52-65 vs 38-48 TF); 1 = software-pipelined form (nsdp_debug_set(1, v))

// Vector-load form of the weight-gradient kernel.  MFMA tiles are free to cover ANY 16 rows/columns of
// the output, so a group of 4 tiles is defined over 64 consecutive columns with tile t owning columns
// {4*li + t}: a lane then reads its 4 tiles' operands with ONE float4 (dY[m][64*ng + 4*li .. +3], and the
// same for X) -- 4x fewer VMEM instructions than one dword per tile, whole 256-B row segments per 16
// lanes.  Output element (tile t, D row i = g*4 + r) is dW row n = 64*ng + 4*i + t; D column j of k-tile
// t' is k = 64*kg + 4*j + t'.
// Waves: waves_n = ceil(N/64) of them split the columns; the remaining factor row_split = 4/waves_n
// splits the chunk's 16-row blocks, each row split writing its own partial slot.
template <int TK4, int THREADS>
__global__ __launch_bounds__(THREADS) void linear_wgrad4_kernel(WgradParams p, int waves_n, int kparts) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int N = p.N, K = p.K;
  // waves = waves_n (column groups of dY) x kparts (column halves of X) x row_split (16-row blocks): the k
  // halves live in the SAME workgroup, so the dY rows they both need come from L1/L2 once instead of
  // being fetched from HBM by two different workgroups
  const int row_split = (THREADS / 64) / (waves_n * kparts);
  const int ng = wave % waves_n;
  const int kh = (wave / waves_n) % kparts;
  const int rs = wave / (waves_n * kparts);
  const int kg0 = kh * TK4;
  const long long m_begin = static_cast<long long>(blockIdx.x) * p.rows_per_chunk;
  const long long m_end = min(p.M, m_begin + p.rows_per_chunk);

  f32x4 acc[4][4 * TK4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4 * TK4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 dbsum = make_float4(0.f, 0.f, 0.f, 0.f);

  // clamped float4 column offsets (N, K are multiples of 4): out-of-range columns feed tiles never stored
  int ncol = 64 * ng + 4 * li;
  ncol = ncol < N ? ncol : (N - 4);
  int kcol[TK4];
#pragma unroll
  for (int b = 0; b < TK4; ++b) {
    const int c = 64 * (kg0 + b) + 4 * li;
    kcol[b] = c < K ? c : (K - 4);
  }
  const bool has_mask = p.mask != nullptr;

  for (long long mb = m_begin + 16LL * rs; mb < m_end; mb += 16LL * row_split) {
    float4 av[4], mv4[4], bv[4][TK4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      long long m = mb + 4 * g + s4;
      m = m < p.M ? m : (p.M - 1);
      av[s4] = *reinterpret_cast<const float4 *>(p.dY + m * N + ncol);
      if (has_mask) mv4[s4] = *reinterpret_cast<const float4 *>(p.mask + m * N + ncol);
#pragma unroll
      for (int b = 0; b < TK4; ++b) bv[s4][b] = *reinterpret_cast<const float4 *>(p.X + m * K + kcol[b]);
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const bool ok = mb + 4 * g + s4 < m_end;  // rows past the chunk contribute zero
      float4 a4 = av[s4];
      if (has_mask) {
        a4.x = mv4[s4].x > 0.f ? a4.x : 0.f; a4.y = mv4[s4].y > 0.f ? a4.y : 0.f;
        a4.z = mv4[s4].z > 0.f ? a4.z : 0.f; a4.w = mv4[s4].w > 0.f ? a4.w : 0.f;
      }
      a4.x = ok ? a4.x : 0.f; a4.y = ok ? a4.y : 0.f; a4.z = ok ? a4.z : 0.f; a4.w = ok ? a4.w : 0.f;
      dbsum.x += a4.x; dbsum.y += a4.y; dbsum.z += a4.z; dbsum.w += a4.w;
      const float aa[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int b = 0; b < TK4; ++b) {
        float4 x4 = bv[s4][b];
        if (p.relu_x) {
          x4.x = fmaxf(x4.x, 0.f); x4.y = fmaxf(x4.y, 0.f); x4.z = fmaxf(x4.z, 0.f); x4.w = fmaxf(x4.w, 0.f);
        }
        const float bb[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
          for (int tb = 0; tb < 4; ++tb)
            acc[ta][4 * b + tb] =
                __builtin_amdgcn_mfma_f32_16x16x4f32(aa[ta], bb[tb], acc[ta][4 * b + tb], 0, 0, 0);
      }
    }
  }

  float *out = p.ws + (static_cast<long long>(blockIdx.x) * row_split + rs) * (static_cast<long long>(N) * K + N);
#pragma unroll
  for (int ta = 0; ta < 4; ++ta) {
#pragma unroll
    for (int b = 0; b < TK4; ++b) {
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) {
        const int kc = 64 * (kg0 + b) + 4 * li + tb;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = 64 * ng + 4 * (g * 4 + r) + ta;
          if (n < N && kc < K) out[static_cast<long long>(n) * K + kc] = acc[ta][4 * b + tb][r];
        }
      }
    }
  }
  if (p.want_db && kh == 0) {
    // lane (li, g) holds the column sums of columns 64*ng + 4*li + {0..3} over its rows (4g+s): fold the 4 g groups
    float v[4] = {dbsum.x, dbsum.y, dbsum.z, dbsum.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      v[t] += __shfl_xor(v[t], 16);
      v[t] += __shfl_xor(v[t], 32);
      const int n = 64 * ng + 4 * li + t;
      if (g == 0 && n < N) out[static_cast<long long>(N) * K + n] = v[t];
    }
  }
}

template <int kWgN, int TK>
void launch_wgrad(const WgradParams &p, unsigned chunks, int ktiles, hipStream_t st) {
  const bool pipe = g_wgrad_pipe > 0;
  const dim3 grid(chunks, (ktiles + TK - 1) / TK);
  if (pipe) hipLaunchKernelGGL((linear_wgrad_kernel<kWgN, TK, true>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((linear_wgrad_kernel<kWgN, TK, false>), grid, dim3(256), 0, st, p);
}

template <int kWgN>
void dispatch_wgrad_k(const WgradParams &p, unsigned chunks, int ktiles, hipStream_t st) {
  // TK chosen so that ceil(ktiles / TK) * TK wastes the fewest tiles while kWgN * TK <= 32 accumulators
  constexpr int kMaxTK = 32 / kWgN > 16 ? 16 : 32 / kWgN;
  if (ktiles <= 1) return launch_wgrad<kWgN, 1>(p, chunks, ktiles, st);
  if (ktiles <= 4) return launch_wgrad<kWgN, 4>(p, chunks, ktiles, st);
  if (ktiles == 13 || ktiles == 7) return launch_wgrad<kWgN, 7>(p, chunks, ktiles, st);
  if (ktiles <= 8 || kMaxTK == 8) return launch_wgrad<kWgN, 8>(p, chunks, ktiles, st);
  return launch_wgrad<kWgN, 16>(p, chunks, ktiles, st);
}

// K = 4 (the 3-d relative coordinates of a position-encoding layer, zero-padded): two multiply-adds per loaded
// float -- nothing for the matrix pipe, a pure stream over dY (and its ReLU mask).  N/4 lanes per row read the row as
// float4 (fully coalesced, 4 rows in flight per lane), each lane keeps the 4 x 4 products of its four channels and
// their column sums; the row slots of a workgroup are combined through LDS in fixed order and written as one partial
// in the layout of reduce_partials_kernel.  The MFMA kernel it replaces here ran at 2.9 TB/s with a quarter of its
// tile rows padding.
// MASK: 0 none, 1 dY *= (mask > 0) with mask [M, N] read from memory, 2 the same mask RECOMPUTED from the layer's input
// rows (already loaded: they are the X operand) and its weights: relu'(x W0^T + b0), bit for bit the forward kernel's
template <int MASK>
__global__ __launch_bounds__(256) void linear_wgrad_k4_kernel(WgradParams p) {
  __shared__ float red[256][21];       // 16 products + 4 column sums per thread (padded: conflict-free columns)
  const int N = p.N;
  const int lpr = N >> 2;                              // lanes per row (N % 4 == 0, N <= 256)
  const int slots = 256 / lpr;                         // rows per workgroup iteration
  const int sub = threadIdx.x / lpr, cq = threadIdx.x - sub * lpr;
  const bool active = sub < slots;
  const long long r0 = static_cast<long long>(blockIdx.x) * p.rows_per_chunk;
  long long r1 = r0 + p.rows_per_chunk;
  r1 = r1 < p.M ? r1 : p.M;
  float a[4][4], bsum[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    bsum[c] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) a[c][k] = 0.f;
  }
  float4 w0[4], b0 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MASK == 2 && active) {
#pragma unroll
    for (int c = 0; c < 4; ++c) w0[c] = *reinterpret_cast<const float4 *>(p.W0 + static_cast<long long>(4 * cq + c) * 4);
    if (p.b0) b0 = *reinterpret_cast<const float4 *>(p.b0 + 4 * cq);
  }
  if (active) {
    constexpr int U = 4;
    for (long long r = r0 + sub; r < r1; r += static_cast<long long>(U) * slots) {
      float4 dy[U], mk[U], x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        long long rr = r + static_cast<long long>(u) * slots;
        rr = rr < r1 ? rr : (r1 - 1);                  // clamped re-read, zeroed below
        dy[u] = *reinterpret_cast<const float4 *>(p.dY + rr * N + 4 * cq);
        if (MASK == 1) mk[u] = *reinterpret_cast<const float4 *>(p.mask + rr * N + 4 * cq);
        x[u] = *reinterpret_cast<const float4 *>(p.X + rr * 4);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool rv = r + static_cast<long long>(u) * slots < r1;
        float d[4] = {dy[u].x, dy[u].y, dy[u].z, dy[u].w};
        if (MASK == 1) {
          d[0] = mk[u].x > 0.f ? d[0] : 0.f; d[1] = mk[u].y > 0.f ? d[1] : 0.f;
          d[2] = mk[u].z > 0.f ? d[2] : 0.f; d[3] = mk[u].w > 0.f ? d[3] : 0.f;
        }
        float xv[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
        if (p.relu_x) {
#pragma unroll
          for (int k = 0; k < 4; ++k) xv[k] = fmaxf(xv[k], 0.f);
        }
        if (MASK == 2) {
          const float4 xq = make_float4(xv[0], xv[1], xv[2], xv[3]);
          d[0] = k4_preact_n(xq, w0[0], b0.x, 0) > 0.f ? d[0] : 0.f; d[1] = k4_preact_n(xq, w0[1], b0.y, 1) > 0.f ? d[1] : 0.f;
          d[2] = k4_preact_n(xq, w0[2], b0.z, 0) > 0.f ? d[2] : 0.f; d[3] = k4_preact_n(xq, w0[3], b0.w, 1) > 0.f ? d[3] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float dc = rv ? d[c] : 0.f;
          bsum[c] += dc;
#pragma unroll
          for (int k = 0; k < 4; ++k) a[c][k] += dc * xv[k];
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int k = 0; k < 4; ++k) red[threadIdx.x][4 * c + k] = a[c][k];
    red[threadIdx.x][16 + c] = bsum[c];
  }
  __syncthreads();
  // thread t < lpr * 20: value v of channel quad cq' = t / 20, summed over the row slots in fixed order
  float *out = p.ws + static_cast<long long>(blockIdx.x) * (static_cast<long long>(N) * 4 + N);
  for (int t = threadIdx.x; t < lpr * 20; t += 256) {
    const int q = t / 20, v = t - q * 20;
    float s = 0.f;
    for (int u = 0; u < slots; ++u) s += red[u * lpr + q][v];
    if (v < 16) out[(4 * q + (v >> 2)) * 4 + (v & 3)] = s;            // dW[n][k], n = 4 q + v / 4
    else if (p.want_db) out[static_cast<long long>(N) * 4 + 4 * q + (v - 16)] = s;
  }
}

__global__ void reduce_partials_kernel(const float *__restrict__ ws, int S, long long stride,
                                       long long nw, float *__restrict__ dW, long long nb,
                                       float *__restrict__ db, int accumulate) {
  const long long e = blockIdx.x * 256LL + threadIdx.x;
  if (e >= nw + nb) return;
  // 8 independent partial sums keep 8 loads in flight per lane (a single dependent chain is latency-bound)
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int c = 0;
  for (; c + 8 <= S; c += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] += ws[(c + u) * stride + e];
  }
  for (; c < S; ++c) acc[0] += ws[c * stride + e];
  const float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  if (e < nw) {
    dW[e] = accumulate ? dW[e] + s : s;
  } else if (db) {
    db[e - nw] = accumulate ? db[e - nw] + s : s;
  }
}

// Fragment-major weight packs for linear_nt_kernel<.., WP = true>:
//   Wp  [ceil(N/16)][ceil(K/16)][64][4]: Wp [((tn*KB + kb)*64 + 16g + li)*4 + c] = W[16tn + li][16kb + 4g + c]
//   WpT [ceil(K/16)][ceil(N/16)][64][4]: WpT[((tk*NB + nb)*64 + 16g + li)*4 + c] = W[16nb + 4g + c][16tk + li]
// (the operand of dX = dY * W, i.e. the pack of W^T), zero outside [N,K].  One thread per float4 of each output.
__global__ __launch_bounds__(256) void pack_weight_kernel(const float *__restrict__ W, int N, int K,
                                                          float *__restrict__ Wp, float *__restrict__ WpT) {
  nsdp::pack::fp32_body(W, N, K, Wp, WpT, static_cast<long long>(blockIdx.x) * 256 + threadIdx.x);
}

}  // namespace

namespace nsdp {
void debug_set_x3(int value);   // gemm_bf16x3.hip
void debug_set_wg16(int value);  // gemm_bf16.hip
void debug_set_lin16(int value);
void debug_set_wg3(int value);   // wgrad_bf16x3.hip
void debug_set_x3_reserve(int value);   // gemm_bf16x3.hip
void debug_set_knn(int value);   // knn.hip
void debug_set_bn(int value);    // batchnorm.hip
void debug_set_search(int value);   // pointnet2_ops.hip
}

extern "C" {

void nsdp_debug_set(int key, int value) {
  if (key == 1) g_wgrad_pipe = value;
  if (key == 3) g_nt_pipe = value;
  if (key == 4) g_nt_dbg = value;
  if (key == 5) g_wgrad_vec4 = value;
  if (key == 6) nsdp::debug_set_x3(value);
  if (key == 7) nsdp::debug_set_wg16(value);
  if (key == 8) nsdp::debug_set_lin16(value);
  if (key == 9) { nsdp::debug_set_wg3(value); nsdp::debug_set_x3_reserve(value); }
  if (key == 10) nsdp::debug_set_knn(value);
  if (key == 11) nsdp::debug_set_bn(value);
  if (key == 12) nsdp::debug_set_search(value);
}

static int linear_dispatch(const float *X, const float *W, const float *bias, const float *residual,
                           const float *mask, const float *out_mask, float *Y, long long M, int N, int K,
                           int relu_in, int relu_out, void *stream, bool wp) {
  if (M <= 0 || N <= 0) return 0;
  NSDP_REQUIRE(X && W && Y, "linear: null pointer");
  NSDP_REQUIRE(K > 0 && K % 4 == 0, "linear: K=%d must be a positive multiple of 4", K);
  NSDP_REQUIRE(N <= 256, "linear: N=%d > 256 is not supported", N);
  NSDP_REQUIRE(((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W)) & 15) == 0 &&
                   (!mask || (reinterpret_cast<uintptr_t>(mask) & 15) == 0),
               "linear: X/W/mask must be 16-byte aligned");
  LinearParams p{X, W, bias, residual, mask, out_mask, Y, g_nt_dbg, M, N, K, relu_in, relu_out};
  hipStream_t st = nsdp::as_stream(stream);
  const int nt = (N + 15) / 16;
  if (K == 4 && N % 4 == 0 && N >= 16 && M >= 4096 && !residual && !mask && !out_mask &&
      ((reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0) {
    nsdp::prof::Scope scope(nsdp::prof::kLinear, st, 2.0 * M * N * K,
                            4.0 * (static_cast<double>(M) * (K + N) + static_cast<double>(N) * K));
    const int slots = 256 / (N >> 2);
    long long grid = (M + 4LL * slots - 1) / (4LL * slots);
    const long long cap = 8LL * nsdp::num_cus();
    grid = grid < cap ? grid : cap;
    if (wp) hipLaunchKernelGGL((linear_k4_fwd_kernel<true>), dim3(static_cast<unsigned>(grid)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((linear_k4_fwd_kernel<false>), dim3(static_cast<unsigned>(grid)), dim3(256), 0, st, p);
    return nsdp::launch_status("linear_k4_fwd_kernel");
  }
  // small M (per-point layers at the 500/100-anchor levels): one 16-row tile x 4 column tiles per wave and
  // the column tiles spread over grid.y, so that a few thousand rows still fill the chip
  if (M <= 32768 && nt > 4 && (wp || g_nt_pipe != 2)) return launch_nt<1, 4>(p, st, (nt + 3) / 4, wp);
  if (nt <= 1) return launch_nt<4, 1>(p, st, 1, wp);
  if (nt <= 4) return launch_nt<4, 4>(p, st, 1, wp);
  if (nt <= 8) return launch_nt<4, 8>(p, st, 1, wp);
  if (nt <= 13) return launch_nt<2, 13>(p, st, 1, wp);
  return launch_nt<2, 16>(p, st, 1, wp);
}

int nsdp_linear_f32(const float *X, const float *W, const float *bias, const float *residual,
                    const float *mask, const float *out_mask, float *Y, long long M, int N, int K,
                    int relu_in, int relu_out, void *stream) {
  return linear_dispatch(X, W, bias, residual, mask, out_mask, Y, M, N, K, relu_in, relu_out, stream, false);
}

int nsdp_linear_wp_f32(const float *X, const float *Wp, const float *bias, const float *residual,
                       const float *mask, const float *out_mask, float *Y, long long M, int N, int K,
                       int relu_in, int relu_out, void *stream) {
  return linear_dispatch(X, Wp, bias, residual, mask, out_mask, Y, M, N, K, relu_in, relu_out, stream, true);
}

long long nsdp_packed_weight_floats(int N, int K) {
  return static_cast<long long>((N + 15) / 16) * ((K + 15) / 16) * 256;
}

int nsdp_pack_weight_f32(const float *W, int N, int K, float *Wp, float *WpT, void *stream) {
  if (N <= 0 || K <= 0) return 0;
  NSDP_REQUIRE(W && (Wp || WpT), "pack_weight: null pointer");
  const long long quads = nsdp_packed_weight_floats(N, K) / 4;
  hipStream_t st = nsdp::as_stream(stream);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(static_cast<unsigned>((quads + 255) / 256)), dim3(256), 0, st, W, N, K,
                     Wp, WpT);
  return nsdp::launch_status("pack_weight_kernel");
}

}  // extern "C"

namespace {
// Row chunking of the weight-gradient reduction: one residency wave of workgroups (2 per CU with the
// register-lean kernel = 512 slots) so that no tail wave runs at low occupancy, >= 128 rows per chunk.
struct WgradPlan {
  long long chunks, rows, slots;  // slots = partial buffers written (chunks x row splits)
  int grid_y;
  bool vec4;
  int tk4, waves_n, kparts, threads;
};
WgradPlan plan_wgrad(long long M, int N, int K) {
  const int ktiles = (K + 15) / 16;
  WgradPlan pl;
  pl.vec4 = (g_wgrad_vec4 != 0) && K > 16 && K <= 256 && (N % 4 == 0) && (K % 4 == 0) && N >= 4;      // (K > 256: more k parts than the float4 form's eight waves)
  long long slots_per_chunk = 1;
  if (pl.vec4) {
    const int kgroups = (K + 63) / 64;
    pl.tk4 = kgroups >= 2 ? 2 : 1;
    pl.kparts = (kgroups + pl.tk4 - 1) / pl.tk4;         // 1 or 2 (K <= 256)
    pl.grid_y = 1;
    pl.waves_n = N <= 64 ? 1 : (N <= 128 ? 2 : 4);
    pl.threads = pl.waves_n * pl.kparts > 4 ? 512 : 256;
    slots_per_chunk = (pl.threads / 64) / (pl.waves_n * pl.kparts);
  } else {
    const int wn = N <= 64 ? 1 : (N <= 128 ? 2 : 4);
    const int max_tk = 32 / wn > 16 ? 16 : 32 / wn;
    int tk;
    if (ktiles <= 1) tk = 1;
    else if (ktiles <= 4) tk = 4;
    else if (ktiles == 13 || ktiles == 7) tk = 7;
    else if (ktiles <= 8 || max_tk == 8) tk = 8;
    else tk = 16;
    pl.grid_y = (ktiles + tk - 1) / tk;
    pl.tk4 = 0;
    pl.waves_n = 0;
    pl.kparts = 1;
    pl.threads = 256;
  }
  long long max_chunks = (pl.vec4 && pl.threads == 512) ? 256 : 512 / pl.grid_y;
  // the partial buffers cost 2 * slots * N*K*4 bytes of extra traffic (write + reduce-read): keep that
  // below ~1/4 of the operand traffic M*(N+K)*4, i.e. slots <= M*(N+K) / (8*N*K)
  const long long traffic_cap = (M * static_cast<long long>(N + K)) / (8LL * N * K) / slots_per_chunk;
  if (max_chunks > traffic_cap) max_chunks = traffic_cap < 96 / slots_per_chunk ? 96 / slots_per_chunk / pl.grid_y : traffic_cap;
  if (max_chunks < 1) max_chunks = 1;
  // below ~2048 rows (the 100-anchor level of an 8-shape batch: 800 rows) the kernel is latency-bound, not traffic-bound: a
  // workgroup walks its 16-row blocks one global round trip at a time, and seven workgroups of 128 rows took 48 us for 0.1 GFLOP.
  // 32-row chunks: four times the workgroups, a quarter of the chain (the partials are a few MB).
  static const int small_rows = getenv("NSDP_WGRAD_SMALL_ROWS") ? atoi(getenv("NSDP_WGRAD_SMALL_ROWS")) : 32;      // (experiment knob)
  const long long min_rows = (M <= 2048 ? small_rows : 128) * slots_per_chunk;
  long long chunks = (M + min_rows - 1) / min_rows;
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  long long rows = (M + chunks - 1) / chunks;
  rows = (rows + 15) / 16 * 16;
  pl.rows = rows;
  pl.chunks = (M + rows - 1) / rows;
  pl.slots = pl.chunks * slots_per_chunk;
  return pl;
}
}  // namespace

extern "C" {

size_t nsdp_linear_wgrad_workspace_bytes(long long M, int N, int K) {
  if (M <= 0) return 0;
  const WgradPlan pl = plan_wgrad(M, N, K);
  return static_cast<size_t>(pl.slots) * (static_cast<size_t>(N) * K + N) * sizeof(float);
}

static int wgrad_f32_impl(const float *dY, const float *X, const float *mask, const float *remask_W,
                          const float *remask_bias, int relu_x, float *dW, float *db, long long M, int N, int K,
                          int accumulate, float *workspace, size_t workspace_bytes, void *stream,
                          NsdpWgradB16ReduceDesc *desc_out = nullptr) {
  if (N <= 0 || K <= 0) return 0;
  NSDP_REQUIRE(dW, "linear_wgrad: null output");
  NSDP_REQUIRE(N <= 256, "linear_wgrad: N=%d > 256 is not supported", N);
  hipStream_t st = nsdp::as_stream(stream);
  if (M <= 0) {
    if (!accumulate) {
      NSDP_HIP_TRY(hipMemsetAsync(dW, 0, sizeof(float) * static_cast<size_t>(N) * K, st));
      if (db) NSDP_HIP_TRY(hipMemsetAsync(db, 0, sizeof(float) * static_cast<size_t>(N), st));
    }
    return 0;
  }
  NSDP_REQUIRE(dY && X && workspace, "linear_wgrad: null pointer");
  NSDP_REQUIRE(workspace_bytes >= nsdp_linear_wgrad_workspace_bytes(M, N, K),
               "linear_wgrad: workspace too small");
  if (!remask_W && wgrad_direct_ok(M, N, K)) {      // few rows: one launch, no partial sums (desc_out: nothing left to reduce, S = 0)
    WgradParams p{dY, X, mask, relu_x, workspace, M, N, K, 32, db != nullptr};
    nsdp::prof::Scope scope(nsdp::prof::kWgrad, st, 2.0 * M * N * K, 4.0 * (static_cast<double>(M) * (K + N)));
    const dim3 grid((N + 31) / 32, (K + 31) / 32);
    NSDP_TRACE("linear_wgrad_direct<%s>", mask ? "mask" : "plain");
    if (mask) hipLaunchKernelGGL((linear_wgrad_direct_kernel<true>), grid, dim3(1024), 0, st, p, dW, db, accumulate);
    else hipLaunchKernelGGL((linear_wgrad_direct_kernel<false>), grid, dim3(1024), 0, st, p, dW, db, accumulate);
    if (desc_out) *desc_out = NsdpWgradB16ReduceDesc{workspace, dW, db, 0, N, K, accumulate ? 1 : 0, 0};
    return nsdp::launch_status("linear_wgrad_direct_kernel");
  }
  const WgradPlan pl = plan_wgrad(M, N, K);
  const long long chunks = pl.chunks, rows = pl.rows;
  WgradParams p{dY, X, mask, relu_x, workspace, M, N, K, rows, db != nullptr};
  const int ktiles = (K + 15) / 16;
  {
    nsdp::prof::Scope scope(nsdp::prof::kWgrad, st, 2.0 * M * N * K,
                            4.0 * (static_cast<double>(M) * (K + N)));
    const unsigned gx = static_cast<unsigned>(chunks);
    const bool k4 = K == 4 && N % 4 == 0 && N >= 16 && pl.slots == chunks && pl.grid_y == 1 &&
        ((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(mask)) & 15) == 0;
    if (remask_W) {
      NSDP_REQUIRE(k4 && !mask, "linear_wgrad_k4_remask: K = 4, N %% 4 == 0, N >= 16, 16-byte aligned operands (N=%d K=%d)", N, K);
      NSDP_REQUIRE(((reinterpret_cast<uintptr_t>(remask_W) | reinterpret_cast<uintptr_t>(remask_bias)) & 15) == 0,
                   "linear_wgrad_k4_remask: weights / bias must be 16-byte aligned");
      p.W0 = remask_W;
      p.b0 = remask_bias;
    }
    if (k4) {
      if (p.W0) hipLaunchKernelGGL((linear_wgrad_k4_kernel<2>), dim3(gx), dim3(256), 0, st, p);
      else if (mask) hipLaunchKernelGGL((linear_wgrad_k4_kernel<1>), dim3(gx), dim3(256), 0, st, p);
      else hipLaunchKernelGGL((linear_wgrad_k4_kernel<0>), dim3(gx), dim3(256), 0, st, p);
    } else if (pl.vec4) {
      const dim3 grid(gx, 1);
      if (pl.threads == 512)
        hipLaunchKernelGGL((linear_wgrad4_kernel<2, 512>), grid, dim3(512), 0, st, p, pl.waves_n, pl.kparts);
      else if (pl.tk4 == 2)
        hipLaunchKernelGGL((linear_wgrad4_kernel<2, 256>), grid, dim3(256), 0, st, p, pl.waves_n, pl.kparts);
      else
        hipLaunchKernelGGL((linear_wgrad4_kernel<1, 256>), grid, dim3(256), 0, st, p, pl.waves_n, pl.kparts);
    } else if (N <= 64) dispatch_wgrad_k<1>(p, gx, ktiles, st);
    else if (N <= 128) dispatch_wgrad_k<2>(p, gx, ktiles, st);
    else dispatch_wgrad_k<4>(p, gx, ktiles, st);
    int rc = nsdp::launch_status("linear_wgrad_kernel");
    if (rc) return rc;
  }
  if (desc_out) {      // the caller sums these partials with a batch: same layout ([slots][N K + N]) and the same eight chains
    *desc_out = NsdpWgradB16ReduceDesc{workspace, dW, db, static_cast<int>(pl.slots), N, K, accumulate ? 1 : 0, 0};
    return 0;
  }
  const long long nw = static_cast<long long>(N) * K;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(static_cast<unsigned>((nw + N + 255) / 256)), dim3(256), 0,
                     st, workspace, static_cast<int>(pl.slots), nw + N, nw, dW, static_cast<long long>(N), db,
                     accumulate);
  return nsdp::launch_status("reduce_partials_kernel");
}

int nsdp_linear_wgrad_f32(const float *dY, const float *X, const float *mask, int relu_x, float *dW,
                          float *db, long long M, int N, int K, int accumulate, float *workspace,
                          size_t workspace_bytes, void *stream) {
  return wgrad_f32_impl(dY, X, mask, nullptr, nullptr, relu_x, dW, db, M, N, K, accumulate, workspace, workspace_bytes,
                        stream);
}

int nsdp_linear_wgrad_partials_f32(const float *dY, const float *X, const float *mask, int relu_x, float *dW, float *db,
                                   long long M, int N, int K, int accumulate, float *workspace, size_t workspace_bytes,
                                   NsdpWgradB16ReduceDesc *desc_out, void *stream) {
  NSDP_REQUIRE(desc_out, "linear_wgrad_partials: null descriptor");
  desc_out->ws = nullptr;
  NSDP_REQUIRE(M > 0, "linear_wgrad_partials: M must be positive (the empty case has no partials: call nsdp_linear_wgrad_f32)");
  return wgrad_f32_impl(dY, X, mask, nullptr, nullptr, relu_x, dW, db, M, N, K, accumulate, workspace, workspace_bytes, stream,
                        desc_out);
}

int nsdp_linear_wgrad_k4_remask_f32(const float *dY, const float *X, const float *W, const float *bias, float *dW,
                                    float *db, long long M, int N, int accumulate, float *workspace,
                                    size_t workspace_bytes, void *stream) {
  NSDP_REQUIRE(W, "linear_wgrad_k4_remask: null weights");
  return wgrad_f32_impl(dY, X, nullptr, W, bias, 0, dW, db, M, N, 4, accumulate, workspace, workspace_bytes, stream);
}

}  // extern "C"
