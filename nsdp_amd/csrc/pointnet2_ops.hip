// Gather / group / ball-query / 3-NN / 3-interpolate operators for gfx950 -- the rest of the
// `pointnet2_ops._ext` surface (reference: _ext-src/src/{sampling,group_points,ball_query,interpolate}_gpu.cu)
// plus the row-major gather/scatter that the model's `index_points` (model/utils.py:58-70) needs.
//
// The reference launches ONE block per batch element for most of these (grid = B), which cannot fill
// 256 CUs; here every kernel is a flat grid-stride launch over output elements with the fastest-moving
// output index on consecutive lanes (coalesced stores; gathers hit L2).  All of them are HBM-bound byte
// movers -- no LDS reuse exists except for the ball-query / 3-NN source cloud, which is LDS-tiled.
#include <cfloat>

#include "common.h"
#include "prof.h"

#pragma clang fp contract(off)

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 256 * 8;  // 8 workgroups per CU, grid-stride beyond that

inline int grid_for(long long work) {
  long long g = (work + kThreads - 1) / kThreads;
  if (g > kMaxBlocks) g = kMaxBlocks;
  if (g < 1) g = 1;
  return static_cast<int>(g);
}

// ---- channel-major gathers: out[b,c,e] = points[b,c,idx[b,e]]  (sampling_gpu.cu:8-20: e = j; group_points_gpu.cu:8-28:
// e = (j,k)) and their gradients grad_points[b,c,idx[b,e]] += grad_out[b,c,e]  (sampling_gpu.cu:34-47,
// group_points_gpu.cu:43-64).
//
// The reference walks (b, c, e) with one thread per element, re-reads idx[b,e] for every channel and scatters with global
// atomics.  Measured here in that form: gathers 0.08-0.28 of the HBM peak (two 64-bit divisions per element; then, with
// the divisions gone, 64 scattered 4-byte reads per wave instruction -- one cache line per lane through the texture path),
// gradients 100-440 GB/s (global fp32 atomics).
//
// Now both directions are LDS-staged: a (b, c) row of the source / target is N floats and fits LDS.  One workgroup of
// 1024 threads owns CH whole rows of one shape (CH x N x 4 B <= 64 KiB: two workgroups per CU; up to 128 KiB for long
// rows) and one slice of the E index entries:
//   gather:   rows -> LDS (coalesced), then per 4 consecutive e: one int4 of indices (loaded once for all CH channels),
//             CH x 4 ds_read_b32 gathers, CH coalesced 16-byte stores;
//   gradient: LDS table zeroed, per 4 consecutive e: one int4 of indices, CH float4 loads issued TOGETHER (a load -> wait
//             -> atomics chain per channel was latency-bound at 0.9 TB/s), CH x 4 ds_add_f32, then every target element is
//             written exactly once (plain stores; partial tables of an E-split are combined with global atomics).
// No division anywhere.  Rows too long for LDS (N > 32768) fall back to the flat kernels.
constexpr int kBig = 1024;                     // threads of the LDS-staged kernels
constexpr int kLdsTableBytes = 64 * 1024;      // two workgroups per CU
constexpr int kLdsTableMax = 128 * 1024;
constexpr int kListSortMax = 1024;             // nsdp_knn_invert leaves longer lists unsorted (csrc/segment.hip kSortMax)

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

__device__ __forceinline__ void load_idx4(const int32_t *ib, int nv, bool vec, int (&i)[4]) {
  if (vec) {
    const int4 v = *reinterpret_cast<const int4 *>(ib);
    i[0] = v.x; i[1] = v.y; i[2] = v.z; i[3] = v.w;
  } else {
    i[0] = ib[0];
    i[1] = nv > 1 ? ib[1] : i[0];
    i[2] = nv > 2 ? ib[2] : i[0];
    i[3] = nv > 3 ? ib[3] : i[0];
  }
}

template <int CH, bool VEC>
__global__ __launch_bounds__(kBig) void gather_cm_lds_kernel(const float *__restrict__ points, const int32_t *__restrict__ idx,
                                                             int C, int N, int E, int e_per_wg, float *__restrict__ out) {
  extern __shared__ float table[];      // [CH][N]
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * CH;
  const int cn = C - c0 < CH ? C - c0 : CH;
  const float *src = points + (static_cast<long long>(b) * C + c0) * N;
  if ((N & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    for (int t = threadIdx.x * 4; t < cn * N; t += kBig * 4)
      *reinterpret_cast<float4 *>(table + t) = *reinterpret_cast<const float4 *>(src + t);
  } else {
    for (int t = threadIdx.x; t < cn * N; t += kBig) table[t] = src[t];
  }
  __syncthreads();
  const int e_begin = blockIdx.z * e_per_wg;
  const int e_end = e_begin + e_per_wg < E ? e_begin + e_per_wg : E;
  const int32_t *ib = idx + static_cast<long long>(b) * E;
  float *o0 = out + (static_cast<long long>(b) * C + c0) * E;
  for (int e0 = e_begin + threadIdx.x * 4; e0 < e_end; e0 += kBig * 4) {
    const int nv = e_end - e0 < 4 ? e_end - e0 : 4;
    int i[4];
    load_idx4(ib + e0, nv, VEC, i);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (c < cn) {
        const float *t = table + c * N;
        float *o = o0 + static_cast<long long>(c) * E + e0;
        if (VEC) {
          *reinterpret_cast<float4 *>(o) = make_float4(t[i[0]], t[i[1]], t[i[2]], t[i[3]]);
        } else {
          o[0] = t[i[0]];
          if (nv > 1) o[1] = t[i[1]];
          if (nv > 2) o[2] = t[i[2]];
          if (nv > 3) o[3] = t[i[3]];
        }
      }
    }
  }
}

template <int CH, bool VEC>
__global__ __launch_bounds__(kBig) void scatter_cm_lds_kernel(const float *__restrict__ grad_out,
                                                              const int32_t *__restrict__ idx, int C, int N, int E, int e_per_wg,
                                                              int combine, float *__restrict__ grad_points) {
  extern __shared__ float table[];      // [CH][N]
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * CH;
  const int cn = C - c0 < CH ? C - c0 : CH;
  for (int t = threadIdx.x; t < cn * N; t += kBig) table[t] = 0.f;
  __syncthreads();
  const int e_begin = blockIdx.z * e_per_wg;
  const int e_end = e_begin + e_per_wg < E ? e_begin + e_per_wg : E;
  const int32_t *ib = idx + static_cast<long long>(b) * E;
  const float *g0 = grad_out + (static_cast<long long>(b) * C + c0) * E;
  for (int e0 = e_begin + threadIdx.x * 4; e0 < e_end; e0 += kBig * 4) {
    const int nv = e_end - e0 < 4 ? e_end - e0 : 4;
    int i[4];
    load_idx4(ib + e0, nv, VEC, i);
    float4 v[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {        // all loads of the chunk in flight together
      const float *g = g0 + static_cast<long long>(c < cn ? c : 0) * E + e0;
      if (VEC) {
        v[c] = *reinterpret_cast<const float4 *>(g);
      } else {
        v[c].x = g[0];
        v[c].y = nv > 1 ? g[1] : 0.f;
        v[c].z = nv > 2 ? g[2] : 0.f;
        v[c].w = nv > 3 ? g[3] : 0.f;
      }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (c < cn) {
        float *t = table + c * N;
        atomicAdd(t + i[0], v[c].x);
        if (VEC || nv > 1) atomicAdd(t + i[1], v[c].y);
        if (VEC || nv > 2) atomicAdd(t + i[2], v[c].z);
        if (VEC || nv > 3) atomicAdd(t + i[3], v[c].w);
      }
    }
  }
  __syncthreads();
  float *o = grad_points + (static_cast<long long>(b) * C + c0) * N;
  if (combine) {      // E was split over several workgroups: partial tables meet in the (zero-filled) target
    for (int t = threadIdx.x; t < cn * N; t += kBig) {
      const float v = table[t];
      if (v != 0.f) atomicAdd(o + t, v);
    }
  } else {
    for (int t = threadIdx.x; t < cn * N; t += kBig) o[t] = table[t];
  }
}

// The same gradient WITHOUT atomics, for callers that hold the inverse of the index map (nsdp_knn_invert: offsets
// [B][N+1], entries [B][E], every list ascending): grad_points[b,c,s] = sum over the list of s of grad_out[b,c,entry].
// LDS fp32 atomics run at ~0.4 per clock and CU on this part -- the table kernel above is bound by them at 0.9 TB/s.
// Here a workgroup stages CH whole rows of grad_out (E floats each) in LDS with coalesced float4 loads and every thread
// owns target points: it walks its list once for all CH channels (LDS reads), sums in list order and writes each target
// element exactly once -- a stream over grad_out, deterministic.  The lists depend on idx[b, :] only: one build serves all
// channels (and is cached by the Python host on the index tensor).
template <int CH>
__global__ __launch_bounds__(kBig) void scatter_cm_lists_kernel(const float *__restrict__ grad_out,
                                                                const int32_t *__restrict__ offsets,
                                                                const int32_t *__restrict__ entries, int C, int N, int E,
                                                                float *__restrict__ grad_points) {
  extern __shared__ float table[];      // [CH][E]
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * CH;
  const int cn = C - c0 < CH ? C - c0 : CH;
  const float *src = grad_out + (static_cast<long long>(b) * C + c0) * E;
  if ((E & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    for (int t = threadIdx.x * 4; t < cn * E; t += kBig * 4)
      *reinterpret_cast<float4 *>(table + t) = *reinterpret_cast<const float4 *>(src + t);
  } else {
    for (int t = threadIdx.x; t < cn * E; t += kBig) table[t] = src[t];
  }
  __syncthreads();
  const int32_t *off = offsets + static_cast<long long>(b) * (N + 1);
  const int32_t *ent = entries + static_cast<long long>(b) * E;
  float *o0 = grad_points + (static_cast<long long>(b) * C + c0) * N;
  for (int s = threadIdx.x; s < N; s += kBig) {
    const int lo = off[s], hi = off[s + 1];
    float acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = 0.f;
    for (int i = lo; i < hi; ++i) {
      const int e = ent[i];
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] += table[(c < cn ? c : 0) * E + e];
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (c < cn) o0[static_cast<long long>(c) * N + s] = acc[c];
  }
}

// three_interpolate_grad through the inverse lists of its (B, n, 3) index map (interpolate_gpu.cu:116-143: three global
// atomics per element): entry e = 3 j + t of the list of source m contributes grad_out[b][c][j] * weight[b][j][t].  The
// workgroup stages CH rows of grad_out (n floats each), a thread owns sources and walks its list once for all CH channels;
// the weights (shared by all channels) come from global memory, once per entry and chunk.
template <int CH>
__global__ __launch_bounds__(kBig) void three_interp_lists_kernel(const float *__restrict__ grad_out,
                                                                  const float *__restrict__ weight,
                                                                  const int32_t *__restrict__ offsets,
                                                                  const int32_t *__restrict__ entries, int C, int n, int m,
                                                                  float *__restrict__ grad_points) {
  extern __shared__ float table[];      // [CH][n]
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * CH;
  const int cn = C - c0 < CH ? C - c0 : CH;
  const float *src = grad_out + (static_cast<long long>(b) * C + c0) * n;
  if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    for (int t = threadIdx.x * 4; t < cn * n; t += kBig * 4)
      *reinterpret_cast<float4 *>(table + t) = *reinterpret_cast<const float4 *>(src + t);
  } else {
    for (int t = threadIdx.x; t < cn * n; t += kBig) table[t] = src[t];
  }
  __syncthreads();
  const int32_t *off = offsets + static_cast<long long>(b) * (m + 1);
  const int32_t *ent = entries + static_cast<long long>(b) * 3 * n;
  const float *wb = weight + static_cast<long long>(b) * 3 * n;
  float *o0 = grad_points + (static_cast<long long>(b) * C + c0) * m;
  for (int s = threadIdx.x; s < m; s += kBig) {
    const int lo = off[s], hi = off[s + 1];
    float acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = 0.f;
    for (int i = lo; i < hi; ++i) {
      const int e = ent[i];
      const float w = wb[e];
      const int j = e / 3;
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] += table[(c < cn ? c : 0) * n + j] * w;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (c < cn) o0[static_cast<long long>(c) * m + s] = acc[c];
  }
}

// scatter_cm_lists for rows LONGER than LDS (E floats per (b, c) row do not fit: PointNet++ SSG shapes, 2048 x 32 entries
// and more): the row is staged in SLICES of `slice` entries; every list is ascending, so the part of a list that falls into
// a slice is a contiguous run -- a thread keeps a cursor per owned target (TP of them, in registers, next to TP x CH
// accumulators), advances it through the slice and writes each target once at the end.  No atomics, one pass over grad_out.
// W3 (three_interpolate_grad for rows longer than LDS): the lists index the (n, 3) weight / index map flattened to 3 E entries --
// entry e = 3 j + t contributes grad_out[.., j] * weight[e]; the staged rows hold E = n floats and a slice of `slice` of them
// covers entries [3 e0, 3 (e0 + len)).
template <int CH, int TP, bool PREFETCH, bool W3 = false>
__global__ __launch_bounds__(kBig) void scatter_cm_lists_sliced_kernel(const float *__restrict__ grad_out,
                                                                       const int32_t *__restrict__ offsets,
                                                                       const int32_t *__restrict__ entries, int C, int N, int E,
                                                                       int slice, float *__restrict__ grad_points,
                                                                       const float *__restrict__ weight = nullptr) {
  extern __shared__ float table[];      // [CH][slice]
  constexpr int kDone = 0x7fffffff;
  constexpr int kMul = W3 ? 3 : 1;      // list entries per staged element
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * CH;
  const int cn = C - c0 < CH ? C - c0 : CH;
  const float *src = grad_out + (static_cast<long long>(b) * C + c0) * E;
  const int32_t *off = offsets + static_cast<long long>(b) * (N + 1);
  const int32_t *ent = entries + static_cast<long long>(b) * E * kMul;
  const float *wb = W3 ? weight + static_cast<long long>(b) * E * kMul : nullptr;
  float *o0 = grad_points + (static_cast<long long>(b) * C + c0) * N;
  for (int s0 = 0; s0 < N; s0 += kBig * TP) {      // (one round for N <= kBig * TP targets)
    // per owned target: cursor, list end, the NEXT entry already in a register (the walk is a chain of dependent loads;
    // with the next entry prefetched the TP chains of a thread advance in parallel)
    int cur[TP], end[TP], nxt[TP];
    bool unsorted[TP];
    float acc[TP][CH];
#pragma unroll
    for (int t = 0; t < TP; ++t) {
      const int s = s0 + threadIdx.x + t * kBig;
      cur[t] = s < N ? off[s] : 0;
      end[t] = s < N ? off[s + 1] : 0;
      // a list longer than nsdp_knn_invert sorts (a hot source point) is in arbitrary order: every slice walks the whole
      // list and takes what falls into it (correct; slow; its summation order is the list's)
      unsorted[t] = end[t] - cur[t] > kListSortMax;
      nxt[t] = (cur[t] < end[t] && !unsorted[t]) ? ent[cur[t]] : kDone;
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[t][c] = 0.f;
    }
    // staging: every thread moves kPre float4 per channel and slice (slice = 4 * kBig * kPre entries).  With 16-byte aligned
    // rows the NEXT slice is loaded into registers before the current one is consumed (the list walk hides the load
    // latency) and written to LDS behind the barrier; ragged rows take the plain path.
    // (PREFETCH is off in the 16-targets-per-thread forms: 48 cursor registers + 64 accumulators leave no room for it
    // under the 128 registers of a 1024-thread workgroup)
    constexpr int kPre = PREFETCH ? 8 / CH : 1;
    const bool vec = PREFETCH && (E & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && slice == 4 * kBig * kPre;
    float4 pre[PREFETCH ? CH : 1][kPre];
    auto fetch = [&](int e0) {
#pragma unroll
      for (int c = 0; c < (PREFETCH ? CH : 1); ++c)
#pragma unroll
        for (int q = 0; q < kPre; ++q) {
          const int t = (q * kBig + threadIdx.x) * 4;
          pre[c][q] = (c < cn && e0 + t < E) ? *reinterpret_cast<const float4 *>(src + static_cast<long long>(c) * E + e0 + t)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if constexpr (PREFETCH) {
      if (vec) fetch(0);
    }
    for (int e0 = 0; e0 < E; e0 += slice) {
      const int len = E - e0 < slice ? E - e0 : slice;
      __syncthreads();                             // the previous slice has been consumed
      bool staged = false;
      if constexpr (PREFETCH) {
        if (vec) {
#pragma unroll
          for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int q = 0; q < kPre; ++q)
              *reinterpret_cast<float4 *>(table + c * slice + (q * kBig + threadIdx.x) * 4) = pre[c][q];
          if (e0 + slice < E) fetch(e0 + slice);     // in flight during this slice's walk
          staged = true;
        }
      }
      if (!staged) {
        for (int c = 0; c < cn; ++c) {
          const float *row = src + static_cast<long long>(c) * E + e0;
          if ((len & 3) == 0 && (reinterpret_cast<uintptr_t>(row) & 15) == 0) {
            for (int t = threadIdx.x * 4; t < len; t += kBig * 4)
              *reinterpret_cast<float4 *>(table + c * slice + t) = *reinterpret_cast<const float4 *>(row + t);
          } else {
            for (int t = threadIdx.x; t < len; t += kBig) table[c * slice + t] = row[t];
          }
        }
      }
      __syncthreads();
      const int e1 = (e0 + len) * kMul;
      bool more = true;
      while (more) {                               // rounds: every target with an entry in this slice consumes ONE
        more = false;
#pragma unroll
        for (int t = 0; t < TP; ++t) {
          if (nxt[t] < e1) {
            const int e = nxt[t] / kMul - e0;
            const float w = W3 ? wb[nxt[t]] : 1.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[t][c] += W3 ? table[(c < cn ? c : 0) * slice + e] * w : table[(c < cn ? c : 0) * slice + e];
            ++cur[t];
            nxt[t] = cur[t] < end[t] ? ent[cur[t]] : kDone;
            more = true;
          }
        }
      }
#pragma unroll
      for (int t = 0; t < TP; ++t) {
        if (unsorted[t]) {
          for (int q = cur[t]; q < end[t]; ++q) {
            const int e = ent[q];
            if (e >= e0 * kMul && e < e1) {
              const float w = W3 ? wb[e] : 1.f;
#pragma unroll
              for (int c = 0; c < CH; ++c) acc[t][c] += table[(c < cn ? c : 0) * slice + (e / kMul - e0)] * w;
            }
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < TP; ++t) {
      const int s = s0 + threadIdx.x + t * kBig;
      if (s < N) {
#pragma unroll
        for (int c = 0; c < CH; ++c)
          if (c < cn) o0[static_cast<long long>(c) * N + s] = acc[t][c];
      }
    }
  }
}

// flat forms (rows that do not fit LDS)
__global__ void gather_cm_flat_kernel(const float *__restrict__ points, const int32_t *__restrict__ idx, long long total, int C,
                                      int N, int E, float *__restrict__ out) {
  for (long long e = blockIdx.x * static_cast<long long>(kThreads) + threadIdx.x; e < total;
       e += static_cast<long long>(gridDim.x) * kThreads) {
    const int j = static_cast<int>(e % E);
    const long long bc = e / E;
    const int b = static_cast<int>(bc / C);
    out[e] = points[bc * N + idx[static_cast<long long>(b) * E + j]];
  }
}

__global__ void scatter_cm_atomic_kernel(const float *__restrict__ grad_out, const int32_t *__restrict__ idx, long long total,
                                         int C, int N, int E, float *__restrict__ grad_points) {
  for (long long e = blockIdx.x * static_cast<long long>(kThreads) + threadIdx.x; e < total;
       e += static_cast<long long>(gridDim.x) * kThreads) {
    const int j = static_cast<int>(e % E);
    const long long bc = e / E;
    const int b = static_cast<int>(bc / C);
    atomicAdd(grad_points + bc * N + idx[static_cast<long long>(b) * E + j], grad_out[e]);
  }
}

// channels per workgroup (power of two <= 8) and the split of E, so that the launch has >= ~2 workgroups per CU
struct CmPlan {
  int ch, zsplit, e_per_wg;
  size_t lds;
};
inline CmPlan plan_cm(int B, int C, int N, int E) {
  const long long row_bytes = 4LL * N;
  int ch = static_cast<int>((row_bytes <= kLdsTableBytes ? kLdsTableBytes : kLdsTableMax) / row_bytes);
  ch = ch >= 8 ? 8 : ch >= 4 ? 4 : ch >= 2 ? 2 : 1;
  const long long want = 2LL * nsdp::num_cus();
  while (ch > 1 && static_cast<long long>(B) * ((C + ch - 1) / ch) < want) ch >>= 1;
  long long wgs = static_cast<long long>(B) * ((C + ch - 1) / ch);
  int zsplit = 1;
  // an E slice must stay much longer than the rows it stages (4 * kBig entries = one pass of the workgroup)
  while (wgs * zsplit < want && E / (zsplit * 2) >= 4 * kBig && E / (zsplit * 2) >= 2 * N && zsplit < 64) zsplit *= 2;
  int e_per = (E + zsplit - 1) / zsplit;
  e_per = (e_per + 3) & ~3;                    // slices start on 16-byte boundaries
  return {ch, (E + e_per - 1) / e_per, e_per, static_cast<size_t>(ch) * N * 4};
}

// Opt a kernel in to more than 64 KiB of dynamic LDS.  The attribute belongs to the CURRENT device's function object and the
// library serves several devices per process (on_device(...) on the Python side): once per (kernel, device), tracked in a
// bitmask per kernel instance; the return code is the caller's to report.
template <auto Kernel>
bool allow_big_lds() {
  static unsigned long long done = 0;      // bit d: set on device d (benign race: setting the attribute twice is harmless)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return false;
  if (dev < 64 && ((done >> dev) & 1ull)) return true;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           kLdsTableMax);
  if (e != hipSuccess) {
    nsdp::set_error("pointnet2_ops: opting a kernel in to %d bytes of LDS failed on device %d: %s", kLdsTableMax, dev,
                    hipGetErrorString(e));
    return false;
  }
  if (dev < 64) done |= 1ull << dev;
  return true;
}

template <int CH>
int launch_gather_cm_t(const CmPlan &pl, bool vec, const float *points, const int32_t *idx, int B, int C, int N, int E,
                       float *out, hipStream_t st) {
  const dim3 grid((C + CH - 1) / CH, B, pl.zsplit);
  if (!(vec ? allow_big_lds<&gather_cm_lds_kernel<CH, true>>() : allow_big_lds<&gather_cm_lds_kernel<CH, false>>())) return NSDP_EINVAL;
  if (vec) hipLaunchKernelGGL((gather_cm_lds_kernel<CH, true>), grid, dim3(kBig), pl.lds, st, points, idx, C, N, E, pl.e_per_wg, out);
  else hipLaunchKernelGGL((gather_cm_lds_kernel<CH, false>), grid, dim3(kBig), pl.lds, st, points, idx, C, N, E, pl.e_per_wg, out);
  return 0;
}

int launch_gather_cm(const float *points, const int32_t *idx, int B, int C, int N, int E, float *out, hipStream_t st) {
  if (4LL * N > kLdsTableMax || B > 65535) {
    const long long total = static_cast<long long>(B) * C * E;
    NSDP_TRACE("gather_cm_flat");
    hipLaunchKernelGGL(gather_cm_flat_kernel, dim3(grid_for(total)), dim3(kThreads), 0, st, points, idx, total, C, N, E, out);
    return 0;
  }
  const CmPlan pl = plan_cm(B, C, N, E);
  const bool vec = E % 4 == 0 && aligned16(idx) && aligned16(out);
  NSDP_TRACE("gather_cm_lds<%d>%s z=%d", pl.ch, vec ? "" : " ragged", pl.zsplit);
  switch (pl.ch) {
    case 8: return launch_gather_cm_t<8>(pl, vec, points, idx, B, C, N, E, out, st);
    case 4: return launch_gather_cm_t<4>(pl, vec, points, idx, B, C, N, E, out, st);
    case 2: return launch_gather_cm_t<2>(pl, vec, points, idx, B, C, N, E, out, st);
    default: return launch_gather_cm_t<1>(pl, vec, points, idx, B, C, N, E, out, st);
  }
}

template <int CH>
int launch_scatter_cm_t(const CmPlan &pl, bool vec, const float *grad_out, const int32_t *idx, int B, int C, int N, int E,
                         float *grad_points, hipStream_t st) {
  const dim3 grid((C + CH - 1) / CH, B, pl.zsplit);
  if (!(vec ? allow_big_lds<&scatter_cm_lds_kernel<CH, true>>() : allow_big_lds<&scatter_cm_lds_kernel<CH, false>>())) return NSDP_EINVAL;
  const int combine = pl.zsplit > 1;
  if (vec)
    hipLaunchKernelGGL((scatter_cm_lds_kernel<CH, true>), grid, dim3(kBig), pl.lds, st, grad_out, idx, C, N, E, pl.e_per_wg, combine, grad_points);
  else
    hipLaunchKernelGGL((scatter_cm_lds_kernel<CH, false>), grid, dim3(kBig), pl.lds, st, grad_out, idx, C, N, E, pl.e_per_wg, combine, grad_points);
  return 0;
}

int launch_scatter_cm(const float *grad_out, const int32_t *idx, int B, int C, int N, int E, float *grad_points,
                      hipStream_t st) {
  if (4LL * N > kLdsTableMax || B > 65535) {
    const long long total = static_cast<long long>(B) * C * E;
    if (hipMemsetAsync(grad_points, 0, sizeof(float) * static_cast<size_t>(B) * C * N, st) != hipSuccess) return 1;
    NSDP_TRACE("scatter_cm_atomic");
    hipLaunchKernelGGL(scatter_cm_atomic_kernel, dim3(grid_for(total)), dim3(kThreads), 0, st, grad_out, idx, total, C, N, E,
                       grad_points);
    return 0;
  }
  const CmPlan pl = plan_cm(B, C, N, E);
  if (pl.zsplit > 1 && hipMemsetAsync(grad_points, 0, sizeof(float) * static_cast<size_t>(B) * C * N, st) != hipSuccess) return 1;
  const bool vec = E % 4 == 0 && aligned16(idx) && aligned16(grad_out);
  NSDP_TRACE("scatter_cm_lds<%d>%s z=%d", pl.ch, vec ? "" : " ragged", pl.zsplit);
  switch (pl.ch) {
    case 8: return launch_scatter_cm_t<8>(pl, vec, grad_out, idx, B, C, N, E, grad_points, st);
    case 4: return launch_scatter_cm_t<4>(pl, vec, grad_out, idx, B, C, N, E, grad_points, st);
    case 2: return launch_scatter_cm_t<2>(pl, vec, grad_out, idx, B, C, N, E, grad_points, st);
    default: return launch_scatter_cm_t<1>(pl, vec, grad_out, idx, B, C, N, E, grad_points, st);
  }
  return 0;
}

constexpr int kSrcTile = 1024;

// ball_query_gpu.cu:9-44: first `nsample` indices (in index order) with d2 < r^2; the first hit also
// pre-fills every slot; no hit leaves zeros (output is zero-filled before the launch).
__global__ __launch_bounds__(kThreads) void ball_query_kernel(
    const float *__restrict__ new_xyz_all, const float *__restrict__ xyz_all, int N, int M,
    float radius2, int nsample, int32_t *__restrict__ idx_all) {
  __shared__ float4 tile[kSrcTile];
  const int b = blockIdx.y;
  const float *xyz = xyz_all + static_cast<size_t>(b) * N * 3;
  const int j = blockIdx.x * kThreads + threadIdx.x;
  const bool active = j < M;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (active) {
    const float *q = new_xyz_all + (static_cast<size_t>(b) * M + j) * 3;
    qx = q[0]; qy = q[1]; qz = q[2];
  }
  int32_t *out = idx_all + (static_cast<size_t>(b) * M + (active ? j : 0)) * nsample;
  int cnt = active ? 0 : nsample;
  for (int base = 0; base < N; base += kSrcTile) {
    const int n_tile = min(kSrcTile, N - base);
    __syncthreads();
    for (int t = threadIdx.x; t < n_tile; t += kThreads) {
      const float *p = xyz + static_cast<size_t>(base + t) * 3;
      tile[t] = make_float4(p[0], p[1], p[2], 0.f);
    }
    __syncthreads();
    for (int t = 0; t < n_tile && cnt < nsample; ++t) {
      const float4 s = tile[t];
      const float d2 = nsdp::sq_dist3(qx, qy, qz, s.x, s.y, s.z);
      if (d2 < radius2) {
        const int k = base + t;
        if (cnt == 0)
          for (int l = 0; l < nsample; ++l) out[l] = k;
        out[cnt] = k;
        ++cnt;
      }
    }
  }
}

// interpolate_gpu.cu:9-59: three nearest `known` points per `unknown` point, bests kept in double.
__global__ __launch_bounds__(kThreads) void three_nn_kernel(const float *__restrict__ unknown_all,
                                                            const float *__restrict__ known_all, int n,
                                                            int m, float *__restrict__ dist2_all,
                                                            int32_t *__restrict__ idx_all) {
  __shared__ float4 tile[kSrcTile];
  const int b = blockIdx.y;
  const float *known = known_all + static_cast<size_t>(b) * m * 3;
  const int j = blockIdx.x * kThreads + threadIdx.x;
  const bool active = j < n;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (active) {
    const float *u = unknown_all + (static_cast<size_t>(b) * n + j) * 3;
    ux = u[0]; uy = u[1]; uz = u[2];
  }
  double best1 = 1e40, best2 = 1e40, best3 = 1e40;
  int besti1 = 0, besti2 = 0, besti3 = 0;
  for (int base = 0; base < m; base += kSrcTile) {
    const int n_tile = min(kSrcTile, m - base);
    __syncthreads();
    for (int t = threadIdx.x; t < n_tile; t += kThreads) {
      const float *p = known + static_cast<size_t>(base + t) * 3;
      tile[t] = make_float4(p[0], p[1], p[2], 0.f);
    }
    __syncthreads();
    for (int t = 0; t < n_tile; ++t) {
      const float4 s = tile[t];
      const double d = static_cast<double>(nsdp::sq_dist3(ux, uy, uz, s.x, s.y, s.z));
      const int k = base + t;
      if (d < best1) {
        best3 = best2; besti3 = besti2;
        best2 = best1; besti2 = besti1;
        best1 = d; besti1 = k;
      } else if (d < best2) {
        best3 = best2; besti3 = besti2;
        best2 = d; besti2 = k;
      } else if (d < best3) {
        best3 = d; besti3 = k;
      }
    }
  }
  if (active) {
    float *d2 = dist2_all + (static_cast<size_t>(b) * n + j) * 3;
    int32_t *io = idx_all + (static_cast<size_t>(b) * n + j) * 3;
    d2[0] = static_cast<float>(best1); d2[1] = static_cast<float>(best2); d2[2] = static_cast<float>(best3);
    io[0] = besti1; io[1] = besti2; io[2] = besti3;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The two searches above with the scan of csrc/knn.hip (knn_split_queue_kernel): the source tile as three coordinate planes
// with 16 sentinel entries behind the last point (x = FLT_MAX squares to +inf, which neither `d < r^2` nor `d < best` admits:
// no range checks), FOUR lanes per query -- a quad walks the tile 16 candidates at a time, lane `sub` takes candidates
// 4 sub .. 4 sub + 3 of every group (one 64-byte LDS read per quad and plane, the same for all 16 quads of a wave: a
// broadcast) -- and the distances two at a time on packed fp32 operations, still ((dx*dx + dy*dy) + dz*dz) with one rounding
// per operation (contraction is off in this file).  With one lane per query (the kernels above) the encoder-sized problems
// are one wave per SIMD and the scan is latency-bound: 0.04-0.18 of the VALU peak in distance tests (profiles/r4_grouping.txt).
// Results are bit-identical to the one-lane kernels (tests/test_geometry_gpu.py holds both to the CPU checker); nsdp_debug_set(12, 0)
// switches back (A/B).
// ---------------------------------------------------------------------------------------------------------------------------
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int kQuad = 4;
constexpr int kQuadQueries = kThreads / kQuad;      // queries per workgroup
constexpr int kBallMaxSample = 64;                  // the LDS output rows of ball_query_quad_kernel

struct PlaneTile {
  __attribute__((aligned(16))) float x[kSrcTile + 16];
  __attribute__((aligned(16))) float y[kSrcTile + 16];
  __attribute__((aligned(16))) float z[kSrcTile + 16];
};

// (the caller has a barrier in front: nobody reads the previous tile any more)
__device__ __forceinline__ void load_plane_tile(PlaneTile &t, const float *__restrict__ src, int cnt) {
  for (int i = threadIdx.x; i < cnt; i += kThreads) {
    const float *p = src + static_cast<size_t>(i) * 3;
    t.x[i] = p[0]; t.y[i] = p[1]; t.z[i] = p[2];
  }
  if (threadIdx.x < 16) {
    t.x[cnt + threadIdx.x] = FLT_MAX; t.y[cnt + threadIdx.x] = 0.f; t.z[cnt + threadIdx.x] = 0.f;
  }
}

// squared distances of the query to candidates i .. i + 3 of the tile (i % 4 == 0)
__device__ __forceinline__ void dist4(const PlaneTile &t, int i, float qx, float qy, float qz, float (&d)[4]) {
  const f32x4_t X = *reinterpret_cast<const f32x4_t *>(&t.x[i]);
  const f32x4_t Y = *reinterpret_cast<const f32x4_t *>(&t.y[i]);
  const f32x4_t Z = *reinterpret_cast<const f32x4_t *>(&t.z[i]);
  const f32x2_t q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
  {
    const f32x2_t dx = q2x - f32x2_t{X[0], X[1]}, dy = q2y - f32x2_t{Y[0], Y[1]}, dz = q2z - f32x2_t{Z[0], Z[1]};
    const f32x2_t r = (dx * dx + dy * dy) + dz * dz;
    d[0] = r[0]; d[1] = r[1];
  }
  {
    const f32x2_t dx = q2x - f32x2_t{X[2], X[3]}, dy = q2y - f32x2_t{Y[2], Y[3]}, dz = q2z - f32x2_t{Z[2], Z[3]};
    const f32x2_t r = (dx * dx + dy * dy) + dz * dz;
    d[2] = r[0]; d[3] = r[1];
  }
}

__global__ __launch_bounds__(kThreads) void ball_query_quad_kernel(const float *__restrict__ new_xyz_all,
                                                                  const float *__restrict__ xyz_all, int N, int M, float radius2,
                                                                  int nsample, int32_t *__restrict__ idx_all) {
  __shared__ PlaneTile tile;
  __shared__ int32_t obuf[kQuadQueries * kBallMaxSample];      // [query][nsample]: written out in one contiguous run
  const int b = blockIdx.y;
  const float *xyz = xyz_all + static_cast<size_t>(b) * N * 3;
  const int ql = threadIdx.x / kQuad, sub = threadIdx.x % kQuad;
  const int j0 = blockIdx.x * kQuadQueries, j = j0 + ql;
  const bool active = j < M;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (active) {
    const float *q = new_xyz_all + (static_cast<size_t>(b) * M + j) * 3;
    qx = q[0]; qy = q[1]; qz = q[2];
  }
  const int lane = threadIdx.x & 63, quad0 = lane & ~3;
  int cnt = active ? 0 : nsample;       // hits so far (the same value in the four lanes of a query)
  int first = 0;                        // the first hit: what the reference pre-fills every slot with
  for (int base = 0; base < N; base += kSrcTile) {
    if (__syncthreads_and(cnt >= nsample)) break;          // (a barrier: the previous tile is no longer read)
    const int n_tile = min(kSrcTile, N - base);
    load_plane_tile(tile, xyz + static_cast<size_t>(base) * 3, n_tile);
    __syncthreads();
    const int steps = (n_tile + 15) >> 4;
    for (int s = 0; s < steps; ++s) {
      if (__builtin_amdgcn_ballot_w64(cnt < nsample) == 0) break;
      const int t = 16 * s + 4 * sub;
      float d[4];
      dist4(tile, t, qx, qy, qz, d);
      const unsigned mask = (d[0] < radius2 ? 1u : 0u) | (d[1] < radius2 ? 2u : 0u) | (d[2] < radius2 ? 4u : 0u) |
                            (d[3] < radius2 ? 8u : 0u);
      if (__builtin_amdgcn_ballot_w64(mask != 0) == 0) continue;
      const int h = __builtin_popcount(mask);
      const int h0 = __shfl(h, quad0), h1 = __shfl(h, quad0 + 1), h2 = __shfl(h, quad0 + 2), h3 = __shfl(h, quad0 + 3);
      const int pre = (sub > 0 ? h0 : 0) + (sub > 1 ? h1 : 0) + (sub > 2 ? h2 : 0);
      const int total = h0 + h1 + h2 + h3;
      // candidates of a group in index order = (lane of the quad, slot of the lane): lane `sub` appends behind the hits of
      // the lanes before it
      const int mine = mask ? base + t + __builtin_ctz(mask) : 0x7fffffff;
      const int f01 = min(__shfl(mine, quad0), __shfl(mine, quad0 + 1));
      const int f23 = min(__shfl(mine, quad0 + 2), __shfl(mine, quad0 + 3));
      if (cnt == 0 && total) first = min(f01, f23);
      if (cnt < nsample) {
        int pos = cnt + pre;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (mask & (1u << u)) {
            if (pos < nsample) obuf[ql * nsample + pos] = base + t + u;
            ++pos;
          }
        }
        cnt += total;
      }
    }
  }
  // slots behind the last hit hold the first hit (ball_query_gpu.cu:33-37); a query without a hit keeps torch::zeros
  const int filled = cnt < nsample ? cnt : nsample;
  if (active)
    for (int l = filled + sub; l < nsample; l += kQuad) obuf[ql * nsample + l] = cnt > 0 ? first : 0;
  __syncthreads();
  const int rows = min(kQuadQueries, M - j0);
  int32_t *out = idx_all + (static_cast<size_t>(b) * M + j0) * nsample;
  for (int e = threadIdx.x; e < rows * nsample; e += kThreads) out[e] = obuf[e];
}

__global__ __launch_bounds__(kThreads) void three_nn_quad_kernel(const float *__restrict__ unknown_all,
                                                                const float *__restrict__ known_all, int n, int m,
                                                                float *__restrict__ dist2_all, int32_t *__restrict__ idx_all) {
  __shared__ PlaneTile tile;
  const int b = blockIdx.y;
  const float *known = known_all + static_cast<size_t>(b) * m * 3;
  const int ql = threadIdx.x / kQuad, sub = threadIdx.x % kQuad;
  const int j = blockIdx.x * kQuadQueries + ql;
  const bool active = j < n;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (active) {
    const float *u = unknown_all + (static_cast<size_t>(b) * n + j) * 3;
    ux = u[0]; uy = u[1]; uz = u[2];
  }
  // (the reference keeps its bests in double, initialised to 1e40: a float distance compares the same against +inf, and
  // (float)1e40 = +inf is what it stores when fewer than three points exist)
  const float inf = __builtin_inff();
  float b1 = inf, b2 = inf, b3 = inf;
  int i1 = 0, i2 = 0, i3 = 0;
  for (int base = 0; base < m; base += kSrcTile) {
    const int n_tile = min(kSrcTile, m - base);
    __syncthreads();
    load_plane_tile(tile, known + static_cast<size_t>(base) * 3, n_tile);
    __syncthreads();
    const int steps = (n_tile + 15) >> 4;
    for (int s = 0; s < steps; ++s) {
      const int t = 16 * s + 4 * sub;
      float d[4];
      dist4(tile, t, ux, uy, uz, d);
      const bool any = d[0] < b3 || d[1] < b3 || d[2] < b3 || d[3] < b3;
      if (__builtin_amdgcn_ballot_w64(any) == 0) continue;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float dd = d[u];
        const int k = base + t + u;
        if (dd < b1) {
          b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = dd; i1 = k;
        } else if (dd < b2) {
          b3 = b2; i3 = i2; b2 = dd; i2 = k;
        } else if (dd < b3) {
          b3 = dd; i3 = k;
        }
      }
    }
  }
  // A lane met its candidates in increasing index order and kept them by strict `<`: its list is ascending in (distance,
  // index).  The quad's four lists merge by (distance, index) into lane 0 -- the order the reference's single scan produces.
  const int lane = threadIdx.x & 63, quad0 = lane & ~3;
  float m1 = b1, m2 = b2, m3 = b3;
  int k1 = i1, k2 = i2, k3 = i3;
#pragma unroll
  for (int o = 1; o < kQuad; ++o) {
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const float dd = __shfl(e == 0 ? b1 : e == 1 ? b2 : b3, quad0 + o);
      const int k = __shfl(e == 0 ? i1 : e == 1 ? i2 : i3, quad0 + o);
      if (dd < m1 || (dd == m1 && k < k1)) {
        m3 = m2; k3 = k2; m2 = m1; k2 = k1; m1 = dd; k1 = k;
      } else if (dd < m2 || (dd == m2 && k < k2)) {
        m3 = m2; k3 = k2; m2 = dd; k2 = k;
      } else if (dd < m3 || (dd == m3 && k < k3)) {
        m3 = dd; k3 = k;
      }
    }
  }
  if (active && sub == 0) {
    float *d2 = dist2_all + (static_cast<size_t>(b) * n + j) * 3;
    int32_t *io = idx_all + (static_cast<size_t>(b) * n + j) * 3;
    d2[0] = m1; d2[1] = m2; d2[2] = m3;
    io[0] = k1; io[1] = k2; io[2] = k3;
  }
}

// rel4[b, i, j] = (sign * (q[b, i] - s[b, idx[b, i, j]]), 0): the relative coordinates every attention block feeds to its
// position-encoding MLP (reference model/encoder/blocks.py:104-106, :285-286, model/decoder/blocks.py:72-78: index_points +
// a broadcast subtraction; here followed by the zero-padding of K = 3 to the K = 4 layer's 16-byte rows) as one kernel with
// one float4 store per (centre, neighbour) instead of a gather, a subtraction, a fill and a strided copy.  One rounding per
// component (q - s, exactly negated for sign < 0), as torch.sub.
__global__ __launch_bounds__(kThreads) void rel_coords4_kernel(const float *__restrict__ q, const float *__restrict__ s,
                                                              const int32_t *__restrict__ idx, long long total, int n, int m,
                                                              int k, int negate, float4 *__restrict__ out) {
  for (long long e = blockIdx.x * static_cast<long long>(kThreads) + threadIdx.x; e < total;
       e += static_cast<long long>(gridDim.x) * kThreads) {
    const long long row = e / k;                 // (b, i)
    const long long b = row / n;
    const float *qp = q + row * 3;
    const float *sp = s + (b * m + idx[e]) * 3;
    float dx = qp[0] - sp[0], dy = qp[1] - sp[1], dz = qp[2] - sp[2];
    if (negate) {
      dx = -dx; dy = -dy; dz = -dz;
    }
    out[e] = make_float4(dx, dy, dz, 0.f);
  }
}

int g_search_quad = 1;      // nsdp_debug_set(12, v), NSDP_SEARCH_QUAD: 0 = the one-lane-per-query kernels (A/B)

// out[b,l,j] = sum_t points[b,l,idx[b,j,t]] * weight[b,j,t]  (interpolate_gpu.cu:72-101), LDS-staged like the gathers:
// a workgroup owns CH whole (b, l) rows of m floats; a thread owns one j at a time -- its three indices and weights are
// loaded once for all CH channels -- and the stores of a wave are 256 contiguous bytes of one channel row.
// (Expression order of the reference: (p1 w1 + p2 w2) + p3 w3, one rounding per operation -- this TU is built with
// fp contraction off.)
template <int CH>
__global__ __launch_bounds__(kBig) void three_interpolate_lds_kernel(const float *__restrict__ points,
                                                                     const int32_t *__restrict__ idx,
                                                                     const float *__restrict__ weight, int c, int m, int n,
                                                                     int j_per_wg, float *__restrict__ out) {
  extern __shared__ float table[];      // [CH][m]
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * CH;
  const int cn = c - c0 < CH ? c - c0 : CH;
  const float *src = points + (static_cast<long long>(b) * c + c0) * m;
  for (int t = threadIdx.x; t < cn * m; t += kBig) table[t] = src[t];
  __syncthreads();
  const int j_begin = blockIdx.z * j_per_wg;
  const int j_end = j_begin + j_per_wg < n ? j_begin + j_per_wg : n;
  float *o0 = out + (static_cast<long long>(b) * c + c0) * n;
  for (int j = j_begin + threadIdx.x; j < j_end; j += kBig) {
    const long long bj = static_cast<long long>(b) * n + j;
    const float w1 = weight[bj * 3 + 0], w2 = weight[bj * 3 + 1], w3 = weight[bj * 3 + 2];
    const int i1 = idx[bj * 3 + 0], i2 = idx[bj * 3 + 1], i3 = idx[bj * 3 + 2];
#pragma unroll
    for (int l = 0; l < CH; ++l) {
      if (l < cn) {
        const float *p = table + l * m;
        o0[static_cast<long long>(l) * n + j] = p[i1] * w1 + p[i2] * w2 + p[i3] * w3;
      }
    }
  }
}

// interpolate_gpu.cu:116-143: grad_points[b,l,idx[b,j,t]] += grad_out[b,l,j] * weight[b,j,t] -- the LDS-table form.
template <int CH>
__global__ __launch_bounds__(kBig) void three_interpolate_grad_lds_kernel(const float *__restrict__ grad_out,
                                                                          const int32_t *__restrict__ idx,
                                                                          const float *__restrict__ weight, int c, int n,
                                                                          int m, int j_per_wg, int combine,
                                                                          float *__restrict__ grad_points) {
  extern __shared__ float table[];      // [CH][m]
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * CH;
  const int cn = c - c0 < CH ? c - c0 : CH;
  for (int t = threadIdx.x; t < cn * m; t += kBig) table[t] = 0.f;
  __syncthreads();
  const int j_begin = blockIdx.z * j_per_wg;
  const int j_end = j_begin + j_per_wg < n ? j_begin + j_per_wg : n;
  const float *g0 = grad_out + (static_cast<long long>(b) * c + c0) * n;
  for (int j = j_begin + threadIdx.x; j < j_end; j += kBig) {
    const long long bj = static_cast<long long>(b) * n + j;
    const float w1 = weight[bj * 3 + 0], w2 = weight[bj * 3 + 1], w3 = weight[bj * 3 + 2];
    const int i1 = idx[bj * 3 + 0], i2 = idx[bj * 3 + 1], i3 = idx[bj * 3 + 2];
    float go[CH];
#pragma unroll
    for (int l = 0; l < CH; ++l) go[l] = g0[static_cast<long long>(l < cn ? l : 0) * n + j];
#pragma unroll
    for (int l = 0; l < CH; ++l) {
      if (l < cn) {
        float *t = table + l * m;
        atomicAdd(t + i1, go[l] * w1);
        atomicAdd(t + i2, go[l] * w2);
        atomicAdd(t + i3, go[l] * w3);
      }
    }
  }
  __syncthreads();
  float *o = grad_points + (static_cast<long long>(b) * c + c0) * m;
  if (combine) {
    for (int t = threadIdx.x; t < cn * m; t += kBig) {
      const float v = table[t];
      if (v != 0.f) atomicAdd(o + t, v);
    }
  } else {
    for (int t = threadIdx.x; t < cn * m; t += kBig) o[t] = table[t];
  }
}

// flat forms (rows that do not fit LDS)
__global__ void three_interpolate_flat_kernel(const float *__restrict__ points, const int32_t *__restrict__ idx,
                                              const float *__restrict__ weight, long long total, int c, int m, int n,
                                              float *__restrict__ out) {
  for (long long e = blockIdx.x * static_cast<long long>(kThreads) + threadIdx.x; e < total;
       e += static_cast<long long>(gridDim.x) * kThreads) {
    const int j = static_cast<int>(e % n);
    const long long bl = e / n;
    const long long bj = (bl / c) * n + j;
    const float *p = points + bl * m;
    const float w1 = weight[bj * 3 + 0], w2 = weight[bj * 3 + 1], w3 = weight[bj * 3 + 2];
    const int i1 = idx[bj * 3 + 0], i2 = idx[bj * 3 + 1], i3 = idx[bj * 3 + 2];
    out[e] = p[i1] * w1 + p[i2] * w2 + p[i3] * w3;
  }
}

__global__ void three_interpolate_grad_kernel(const float *__restrict__ grad_out,
                                              const int32_t *__restrict__ idx,
                                              const float *__restrict__ weight, long long total, int c,
                                              int n, int m, float *__restrict__ grad_points) {
  for (long long e = blockIdx.x * static_cast<long long>(kThreads) + threadIdx.x; e < total;
       e += static_cast<long long>(gridDim.x) * kThreads) {
    const int j = static_cast<int>(e % n);
    const long long bl = e / n;
    const long long bj = (bl / c) * n + j;
    float *g = grad_points + bl * m;
    const float go = grad_out[e];
    atomicAdd(g + idx[bj * 3 + 0], go * weight[bj * 3 + 0]);
    atomicAdd(g + idx[bj * 3 + 1], go * weight[bj * 3 + 1]);
    atomicAdd(g + idx[bj * 3 + 2], go * weight[bj * 3 + 2]);
  }
}

// index_points on row-major features: out[b,s,:] = points[b,idx[b,s],:]; VEC floats per lane.
template <int VEC>
__global__ void gather_rows_kernel(const float *__restrict__ points, const int32_t *__restrict__ idx,
                                   long long total_vec, int N, int CV, int S,
                                   float *__restrict__ out) {
  for (long long e = blockIdx.x * static_cast<long long>(kThreads) + threadIdx.x; e < total_vec;
       e += static_cast<long long>(gridDim.x) * kThreads) {
    const int cv = static_cast<int>(e % CV);
    const long long bs = e / CV;
    const long long b = bs / S;
    const long long src = (b * N + idx[bs]) * CV + cv;
    if (VEC == 4)
      reinterpret_cast<float4 *>(out)[e] = reinterpret_cast<const float4 *>(points)[src];
    else
      out[e] = points[src];
  }
}

__global__ void scatter_add_rows_kernel(const float *__restrict__ grad_out,
                                        const int32_t *__restrict__ idx, long long total, int N, int C,
                                        int S, float *__restrict__ grad_points) {
  for (long long e = blockIdx.x * static_cast<long long>(kThreads) + threadIdx.x; e < total;
       e += static_cast<long long>(gridDim.x) * kThreads) {
    const int c = static_cast<int>(e % C);
    const long long bs = e / C;
    const long long b = bs / S;
    atomicAdd(grad_points + (b * N + idx[bs]) * C + c, grad_out[e]);
  }
}

}  // namespace

namespace nsdp {
void debug_set_search(int value) { g_search_quad = value; }
}  // namespace nsdp

extern "C" {

int nsdp_gather_points(const float *points, const int32_t *idx, int B, int C, int N, int M, float *out,
                       void *stream) {
  const long long total = static_cast<long long>(B) * C * M;
  if (total <= 0) return 0;
  NSDP_REQUIRE(points && idx && out && N > 0, "gather_points: bad argument");
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kGatherRows, st, 0.0, 4.0 * (static_cast<double>(B) * M * (1 + C) + static_cast<double>(B) * C * N));
  if (launch_gather_cm(points, idx, B, C, N, M, out, st)) return 1;
  return nsdp::launch_status("gather_cm_kernel");
}

int nsdp_gather_points_grad(const float *grad_out, const int32_t *idx, int B, int C, int N, int M,
                            float *grad_points, void *stream) {
  const long long total = static_cast<long long>(B) * C * M;
  NSDP_REQUIRE(grad_points || static_cast<long long>(B) * C * N == 0, "gather_points_grad: null output");
  hipStream_t st = nsdp::as_stream(stream);
  if (total <= 0) {
    if (static_cast<long long>(B) * C * N > 0)
      NSDP_HIP_TRY(hipMemsetAsync(grad_points, 0, sizeof(float) * static_cast<size_t>(B) * C * N, st));
    return 0;
  }
  NSDP_REQUIRE(grad_out && idx, "gather_points_grad: null pointer");
  nsdp::prof::Scope scope(nsdp::prof::kScatterRows, st, 0.0, 4.0 * (static_cast<double>(B) * M * (1 + C) + static_cast<double>(B) * C * N));
  if (launch_scatter_cm(grad_out, idx, B, C, N, M, grad_points, st)) return 1;
  return nsdp::launch_status("scatter_cm_kernel");
}

int nsdp_group_points(const float *points, const int32_t *idx, int B, int C, int N, int NP, int NS,
                      float *out, void *stream) {
  const long long total = static_cast<long long>(B) * C * NP * NS;
  if (total <= 0) return 0;
  NSDP_REQUIRE(points && idx && out && N > 0, "group_points: bad argument");
  NSDP_REQUIRE(static_cast<long long>(NP) * NS < (1LL << 31), "group_points: npoint * nsample too large");
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kGatherRows, st, 0.0,
                          4.0 * (static_cast<double>(B) * NP * NS * (1 + C) + static_cast<double>(B) * C * N));
  if (launch_gather_cm(points, idx, B, C, N, NP * NS, out, st)) return 1;
  return nsdp::launch_status("gather_cm_kernel");
}

int nsdp_group_points_grad(const float *grad_out, const int32_t *idx, int B, int C, int N, int NP,
                           int NS, float *grad_points, void *stream) {
  const long long total = static_cast<long long>(B) * C * NP * NS;
  NSDP_REQUIRE(grad_points || static_cast<long long>(B) * C * N == 0, "group_points_grad: null output");
  hipStream_t st = nsdp::as_stream(stream);
  if (total <= 0) {
    if (static_cast<long long>(B) * C * N > 0)
      NSDP_HIP_TRY(hipMemsetAsync(grad_points, 0, sizeof(float) * static_cast<size_t>(B) * C * N, st));
    return 0;
  }
  NSDP_REQUIRE(grad_out && idx, "group_points_grad: null pointer");
  NSDP_REQUIRE(static_cast<long long>(NP) * NS < (1LL << 31), "group_points_grad: npoint * nsample too large");
  nsdp::prof::Scope scope(nsdp::prof::kScatterRows, st, 0.0,
                          4.0 * (static_cast<double>(B) * NP * NS * (1 + C) + static_cast<double>(B) * C * N));
  if (launch_scatter_cm(grad_out, idx, B, C, N, NP * NS, grad_points, st)) return 1;
  return nsdp::launch_status("scatter_cm_kernel");
}

int nsdp_ball_query(const float *new_xyz, const float *xyz, int B, int N, int M, float radius,
                    int nsample, int32_t *idx_out, void *stream) {
  const long long total = static_cast<long long>(B) * M * nsample;
  if (total <= 0) return 0;
  NSDP_REQUIRE(new_xyz && xyz && idx_out, "ball_query: null pointer");
  NSDP_REQUIRE(B <= 65535, "ball_query: batch %d too large", B);
  hipStream_t st = nsdp::as_stream(stream);
  const float radius2 = radius * radius;
  if (g_search_quad && nsample <= kBallMaxSample) {      // (writes every slot itself: no memset in front)
    NSDP_TRACE("ball_query_quad");
    hipLaunchKernelGGL(ball_query_quad_kernel, dim3(nsdp::ceil_div(M, kQuadQueries), B), dim3(kThreads), 0, st, new_xyz, xyz, N,
                       M, radius2, nsample, idx_out);
    return nsdp::launch_status("ball_query_quad_kernel");
  }
  NSDP_HIP_TRY(hipMemsetAsync(idx_out, 0, sizeof(int32_t) * static_cast<size_t>(total), st));
  NSDP_TRACE("ball_query");
  hipLaunchKernelGGL(ball_query_kernel, dim3(nsdp::ceil_div(M, kThreads), B), dim3(kThreads), 0, st,
                     new_xyz, xyz, N, M, radius2, nsample, idx_out);
  return nsdp::launch_status("ball_query_kernel");
}

int nsdp_three_nn(const float *unknown, const float *known, int B, int n, int m, float *dist2,
                  int32_t *idx, void *stream) {
  if (static_cast<long long>(B) * n <= 0) return 0;
  NSDP_REQUIRE(unknown && dist2 && idx && (known || m == 0), "three_nn: null pointer");
  NSDP_REQUIRE(B <= 65535, "three_nn: batch %d too large", B);
  if (g_search_quad) {
    NSDP_TRACE("three_nn_quad");
    hipLaunchKernelGGL(three_nn_quad_kernel, dim3(nsdp::ceil_div(n, kQuadQueries), B), dim3(kThreads), 0,
                       nsdp::as_stream(stream), unknown, known, n, m, dist2, idx);
    return nsdp::launch_status("three_nn_quad_kernel");
  }
  NSDP_TRACE("three_nn");
  hipLaunchKernelGGL(three_nn_kernel, dim3(nsdp::ceil_div(n, kThreads), B), dim3(kThreads), 0,
                     nsdp::as_stream(stream), unknown, known, n, m, dist2, idx);
  return nsdp::launch_status("three_nn_kernel");
}

int nsdp_three_interpolate(const float *points, const int32_t *idx, const float *weight, int B, int c,
                           int m, int n, float *out, void *stream) {
  const long long total = static_cast<long long>(B) * c * n;
  if (total <= 0) return 0;
  NSDP_REQUIRE(points && idx && weight && out && m > 0, "three_interpolate: bad argument");
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kGatherRows, st, 0.0,
                          4.0 * (static_cast<double>(B) * n * (6 + c) + static_cast<double>(B) * c * m));
  if (4LL * m > kLdsTableMax || B > 65535) {
    hipLaunchKernelGGL(three_interpolate_flat_kernel, dim3(grid_for(total)), dim3(kThreads), 0, st, points, idx, weight, total,
                       c, m, n, out);
    return nsdp::launch_status("three_interpolate_flat_kernel");
  }
  const CmPlan pl = plan_cm(B, c, m, n);
  const dim3 grid((c + pl.ch - 1) / pl.ch, B, pl.zsplit);
  NSDP_TRACE("three_interpolate_lds<%d> z=%d", pl.ch, pl.zsplit);
#define NSDP_TI(CH)                                                                                                        \
  {                                                                                                                        \
    if (!allow_big_lds<&three_interpolate_lds_kernel<CH>>()) return NSDP_EINVAL;                                        \
    hipLaunchKernelGGL((three_interpolate_lds_kernel<CH>), grid, dim3(kBig), pl.lds, st, points, idx, weight, c, m, n,      \
                       pl.e_per_wg, out);                                                                                  \
  }
  switch (pl.ch) {
    case 8: NSDP_TI(8) break;
    case 4: NSDP_TI(4) break;
    case 2: NSDP_TI(2) break;
    default: NSDP_TI(1) break;
  }
#undef NSDP_TI
  return nsdp::launch_status("three_interpolate_lds_kernel");
}

int nsdp_three_interpolate_grad(const float *grad_out, const int32_t *idx, const float *weight, int B,
                                int c, int n, int m, float *grad_points, void *stream) {
  const long long total = static_cast<long long>(B) * c * n;
  NSDP_REQUIRE(grad_points || static_cast<long long>(B) * c * m == 0, "three_interpolate_grad: null output");
  hipStream_t st = nsdp::as_stream(stream);
  if (total <= 0) {
    if (static_cast<long long>(B) * c * m > 0)
      NSDP_HIP_TRY(hipMemsetAsync(grad_points, 0, sizeof(float) * static_cast<size_t>(B) * c * m, st));
    return 0;
  }
  NSDP_REQUIRE(grad_out && idx && weight, "three_interpolate_grad: null pointer");
  nsdp::prof::Scope scope(nsdp::prof::kScatterRows, st, 0.0,
                          4.0 * (static_cast<double>(B) * n * (6 + c) + static_cast<double>(B) * c * m));
  if (4LL * m <= kLdsTableMax && B <= 65535) {
    const CmPlan pl = plan_cm(B, c, m, n);
    if (pl.zsplit > 1) NSDP_HIP_TRY(hipMemsetAsync(grad_points, 0, sizeof(float) * static_cast<size_t>(B) * c * m, st));
    const dim3 grid((c + pl.ch - 1) / pl.ch, B, pl.zsplit);
    const int combine = pl.zsplit > 1;
    NSDP_TRACE("three_interpolate_grad_lds<%d> z=%d", pl.ch, pl.zsplit);
#define NSDP_TIG(CH)                                                                                                       \
  {                                                                                                                        \
    if (!allow_big_lds<&three_interpolate_grad_lds_kernel<CH>>()) return NSDP_EINVAL;                                        \
    hipLaunchKernelGGL((three_interpolate_grad_lds_kernel<CH>), grid, dim3(kBig), pl.lds, st, grad_out, idx, weight, c, n,  \
                       m, pl.e_per_wg, combine, grad_points);                                                              \
  }
    switch (pl.ch) {
      case 8: NSDP_TIG(8) break;
      case 4: NSDP_TIG(4) break;
      case 2: NSDP_TIG(2) break;
      default: NSDP_TIG(1) break;
    }
#undef NSDP_TIG
    return nsdp::launch_status("three_interpolate_grad_lds_kernel");
  }
  NSDP_HIP_TRY(hipMemsetAsync(grad_points, 0, sizeof(float) * static_cast<size_t>(B) * c * m, st));
  hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(grid_for(total)), dim3(kThreads), 0, st,
                     grad_out, idx, weight, total, c, n, m, grad_points);
  return nsdp::launch_status("three_interpolate_grad_kernel");
}

int nsdp_scatter_cm_lists_supported(int B, int C, int N, int E) {
  return B > 0 && B <= 65535 && C > 0 && N > 0 && N <= 32768 && E > 0;
}

int nsdp_scatter_cm_lists(const float *grad_out, const int32_t *offsets, const int32_t *entries, int B, int C, int N,
                          int E, float *grad_points, void *stream) {
  if (static_cast<long long>(B) * C * N <= 0) return 0;
  NSDP_REQUIRE(grad_out && offsets && entries && grad_points, "scatter_cm_lists: null pointer");
  NSDP_REQUIRE(nsdp_scatter_cm_lists_supported(B, C, N, E), "scatter_cm_lists: unsupported shape (B=%d C=%d N=%d E=%d)", B, C, N, E);
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kScatterRows, st, 0.0,
                          4.0 * (static_cast<double>(B) * E * (1 + C) + static_cast<double>(B) * C * N));
  const long long row_bytes = 4LL * E;
  if (row_bytes > kLdsTableMax) {
    // rows longer than LDS: slices of the row, cursors into the ascending lists (scatter_cm_lists_sliced_kernel).  Four
    // channels share one walk of the lists when that still leaves two workgroups per CU.
    const bool many = N > 8 * kBig;      // 16 targets per thread: two channels only (registers)
    const bool four = !many && static_cast<long long>(B) * ((C + 3) / 4) >= 2LL * nsdp::num_cus();
    const int ch = four ? 4 : 2;
    const int slice = kLdsTableMax / 4 / ch;
    const dim3 grid((C + ch - 1) / ch, B);
    NSDP_TRACE("scatter_cm_lists_sliced<%d,%d>", ch, many ? 16 : 8);
#define NSDP_SCS(CH, TP, PF)                                                                                               \
  {                                                                                                                        \
    if (!allow_big_lds<&scatter_cm_lists_sliced_kernel<CH, TP, PF>>()) return NSDP_EINVAL;                                 \
    hipLaunchKernelGGL((scatter_cm_lists_sliced_kernel<CH, TP, PF>), grid, dim3(kBig), static_cast<size_t>(kLdsTableMax),   \
                       st, grad_out, offsets, entries, C, N, E, slice, grad_points);                                       \
  }
    if (four) NSDP_SCS(4, 8, true)
    else if (many) NSDP_SCS(2, 16, false)
    else NSDP_SCS(2, 8, true)
#undef NSDP_SCS
    return nsdp::launch_status("scatter_cm_lists_sliced_kernel");
  }
  int ch = static_cast<int>((row_bytes <= kLdsTableBytes ? kLdsTableBytes : kLdsTableMax) / row_bytes);
  ch = ch >= 8 ? 8 : ch >= 4 ? 4 : ch >= 2 ? 2 : 1;
  while (ch > 1 && static_cast<long long>(B) * ((C + ch - 1) / ch) < 2LL * nsdp::num_cus()) ch >>= 1;
  const dim3 grid((C + ch - 1) / ch, B);
  const size_t lds = static_cast<size_t>(ch) * E * 4;
  NSDP_TRACE("scatter_cm_lists<%d>", ch);
#define NSDP_SCL(CH)                                                                                                       \
  {                                                                                                                        \
    if (!allow_big_lds<&scatter_cm_lists_kernel<CH>>()) return NSDP_EINVAL;                                        \
    hipLaunchKernelGGL((scatter_cm_lists_kernel<CH>), grid, dim3(kBig), lds, st, grad_out, offsets, entries, C, N, E,       \
                       grad_points);                                                                                       \
  }
  switch (ch) {
    case 8: NSDP_SCL(8) break;
    case 4: NSDP_SCL(4) break;
    case 2: NSDP_SCL(2) break;
    default: NSDP_SCL(1) break;
  }
#undef NSDP_SCL
  return nsdp::launch_status("scatter_cm_lists_kernel");
}

int nsdp_three_interpolate_grad_lists_supported(int B, int c, int n, int m) {
  return B > 0 && B <= 65535 && c > 0 && m > 0 && m <= 32768 && n > 0 && n <= (1 << 28);
}

int nsdp_three_interpolate_grad_lists(const float *grad_out, const float *weight, const int32_t *offsets,
                                      const int32_t *entries, int B, int c, int n, int m, float *grad_points, void *stream) {
  if (static_cast<long long>(B) * c * m <= 0) return 0;
  NSDP_REQUIRE(grad_out && weight && offsets && entries && grad_points, "three_interpolate_grad_lists: null pointer");
  NSDP_REQUIRE(nsdp_three_interpolate_grad_lists_supported(B, c, n, m), "three_interpolate_grad_lists: unsupported shape (m <= 32768)");
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kScatterRows, st, 0.0,
                          4.0 * (static_cast<double>(B) * n * (6 + c) + static_cast<double>(B) * c * m));
  const long long row_bytes = 4LL * n;
  // rows that leave room for < 4 channels per workgroup (n > 8192): the sliced form -- the list entries and weights of a walk are
  // shared by 4 (m <= 8192) or 2 channels, the rows pass through LDS in slices
  if (row_bytes > kLdsTableBytes / 2) {
    const bool many = m > 8 * kBig;
    const int ch = many ? 2 : 4;
    const int slice = kLdsTableMax / 4 / ch;
    const dim3 grid((c + ch - 1) / ch, B);
    NSDP_TRACE("three_interpolate_grad_sliced<%d,%d>", ch, many ? 16 : 8);
    if (many) {
      if (!allow_big_lds<&scatter_cm_lists_sliced_kernel<2, 16, false, true>>()) return NSDP_EINVAL;
      hipLaunchKernelGGL((scatter_cm_lists_sliced_kernel<2, 16, false, true>), grid, dim3(kBig), static_cast<size_t>(kLdsTableMax), st,
                         grad_out, offsets, entries, c, m, n, slice, grad_points, weight);
    } else {
      if (!allow_big_lds<&scatter_cm_lists_sliced_kernel<4, 8, true, true>>()) return NSDP_EINVAL;
      hipLaunchKernelGGL((scatter_cm_lists_sliced_kernel<4, 8, true, true>), grid, dim3(kBig), static_cast<size_t>(kLdsTableMax), st,
                         grad_out, offsets, entries, c, m, n, slice, grad_points, weight);
    }
    return nsdp::launch_status("scatter_cm_lists_sliced_kernel<w3>");
  }
  int ch = static_cast<int>((row_bytes <= kLdsTableBytes ? kLdsTableBytes : kLdsTableMax) / row_bytes);
  ch = ch >= 8 ? 8 : ch >= 4 ? 4 : ch >= 2 ? 2 : 1;
  while (ch > 1 && static_cast<long long>(B) * ((c + ch - 1) / ch) < 2LL * nsdp::num_cus()) ch >>= 1;
  const dim3 grid((c + ch - 1) / ch, B);
  const size_t lds = static_cast<size_t>(ch) * n * 4;
  NSDP_TRACE("three_interpolate_grad_lists<%d>", ch);
#define NSDP_TIL(CH)                                                                                                       \
  {                                                                                                                        \
    if (!allow_big_lds<&three_interp_lists_kernel<CH>>()) return NSDP_EINVAL;                                              \
    hipLaunchKernelGGL((three_interp_lists_kernel<CH>), grid, dim3(kBig), lds, st, grad_out, weight, offsets, entries, c,   \
                       n, m, grad_points);                                                                                 \
  }
  switch (ch) {
    case 8: NSDP_TIL(8) break;
    case 4: NSDP_TIL(4) break;
    case 2: NSDP_TIL(2) break;
    default: NSDP_TIL(1) break;
  }
#undef NSDP_TIL
  return nsdp::launch_status("three_interp_lists_kernel");
}

int nsdp_gather_rows(const float *points, const int32_t *idx, int B, int N, int C, int S, float *out,
                     void *stream) {
  const long long total = static_cast<long long>(B) * S * C;
  if (total <= 0) return 0;
  NSDP_REQUIRE(points && idx && out && N > 0, "gather_rows: bad argument");
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kGatherRows, st, 0.0,
                          4.0 * (static_cast<double>(B) * S * (1 + C) + static_cast<double>(B) * N * C));
  const bool vec4 = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(points) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
  if (vec4) {
    const long long tv = total / 4;
    hipLaunchKernelGGL((gather_rows_kernel<4>), dim3(grid_for(tv)), dim3(kThreads), 0, st, points, idx,
                       tv, N, C / 4, S, out);
  } else {
    hipLaunchKernelGGL((gather_rows_kernel<1>), dim3(grid_for(total)), dim3(kThreads), 0, st, points,
                       idx, total, N, C, S, out);
  }
  return nsdp::launch_status("gather_rows_kernel");
}

int nsdp_rel_coords4(const float *query, const float *source, const int32_t *idx, int B, int n, int m, int k, float sign,
                     float *out4, void *stream) {
  const long long total = static_cast<long long>(B) * n * k;
  if (total <= 0) return 0;
  NSDP_REQUIRE(query && source && idx && out4 && m > 0, "rel_coords4: bad argument");
  NSDP_REQUIRE(sign == 1.f || sign == -1.f, "rel_coords4: sign must be +1 or -1");
  NSDP_REQUIRE(reinterpret_cast<uintptr_t>(out4) % 16 == 0, "rel_coords4: output must be 16-byte aligned");
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kGatherRows, st, 0.0, 4.0 * (5.0 * total + 3.0 * B * (n + m)));
  hipLaunchKernelGGL(rel_coords4_kernel, dim3(grid_for(total)), dim3(kThreads), 0, st, query, source, idx, total, n, m, k,
                     sign < 0.f ? 1 : 0, reinterpret_cast<float4 *>(out4));
  return nsdp::launch_status("rel_coords4_kernel");
}

int nsdp_scatter_add_rows(const float *grad_out, const int32_t *idx, int B, int N, int C, int S,
                          float *grad_points, void *stream) {
  const long long total = static_cast<long long>(B) * S * C;
  NSDP_REQUIRE(grad_points || static_cast<long long>(B) * N * C == 0, "scatter_add_rows: null output");
  hipStream_t st = nsdp::as_stream(stream);
  if (static_cast<long long>(B) * N * C > 0)
    NSDP_HIP_TRY(hipMemsetAsync(grad_points, 0, sizeof(float) * static_cast<size_t>(B) * N * C, st));
  if (total <= 0) return 0;
  NSDP_REQUIRE(grad_out && idx, "scatter_add_rows: null pointer");
  nsdp::prof::Scope scope(nsdp::prof::kScatterRows, st, 0.0,
                          4.0 * (static_cast<double>(B) * S * (1 + C) + 2.0 * static_cast<double>(B) * N * C));
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid_for(total)), dim3(kThreads), 0, st, grad_out,
                     idx, total, N, C, S, grad_points);
  return nsdp::launch_status("scatter_add_rows_kernel");
}

}  // extern "C"
