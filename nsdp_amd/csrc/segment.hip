// Atomics-free scatter for the attention backward of the encoder levels whose source table does not fit into LDS or
// registers (2048 / 500 source points): INVERSE NEIGHBOUR LISTS + SEGMENT SUMS.
//
// The scatter targets of an attention block -- dvf[b][s] = sum of d(pos) rows over all (centre, slot) pairs that
// reference source point s, dkf likewise with d(u) -- are sums over the inverse of the kNN index map.  That inverse
// depends on coordinates only: it is built ONCE per index set and step (counting sort in LDS, one workgroup per
// shape, each list then sorted so that the summation order is fixed) and re-used by every scatter of the block
// (model/encoder/blocks.py:104-124, :290-308 backward; two attentions share the set abstraction's index set).
// A scatter is then a gather-reduce: one 8/16-byte row segment per lane and list entry, every source row read once,
// fp32 accumulation in list order -- deterministic, no atomics (the atomic kernels were bound by their fp32 atomics:
// 200 G atomics/s, 0.40 ms for the 2048-point block in either storage type).
#include "common.h"
#include "prof.h"

namespace {

constexpr int kInvThreads = 1024;
constexpr int kMaxSources = 32768;     // counters of one shape in (dynamic) LDS: 128 KiB + 4 KiB of scan scratch
constexpr int kSortMax = 1024;         // longest list that is put into ascending order (fixed summation order)

// offsets [B][N+1], entries [B][E]: entries[b][offsets[b][s] .. offsets[b][s+1]) = ascending list of e = i*k + j with
// idx[b][e] == s
// LDS_ENTRIES: the lists are filled in LDS (E more ints next to the counters) and ordered from there (the ordering pass in
// global memory was one thread per list walking dependent loads and stores through L2: the longest list of a 2048-point level,
// ~30 entries, took 130 us of a 136 us launch)
template <bool LDS_ENTRIES>
__global__ __launch_bounds__(kInvThreads) void knn_invert_kernel(const int32_t *__restrict__ idx_all, int E, int N,
                                                                 int32_t *__restrict__ offsets_all,
                                                                 int32_t *__restrict__ entries_all) {
  extern __shared__ int cnt[];           // [N + 1] (+ [E] list entries)
  __shared__ int part[kInvThreads];
  const int b = blockIdx.x;
  const int32_t *idx = idx_all + static_cast<long long>(b) * E;
  int32_t *offsets = offsets_all + static_cast<long long>(b) * (N + 1);
  int32_t *entries = entries_all + static_cast<long long>(b) * E;
  int *ent = LDS_ENTRIES ? cnt + (N + 1) : entries;
  for (int s = threadIdx.x; s <= N; s += kInvThreads) cnt[s] = 0;
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += kInvThreads) atomicAdd(&cnt[idx[e]], 1);
  __syncthreads();
  // exclusive scan of cnt[0..N): every thread owns a contiguous chunk
  const int chunk = (N + kInvThreads - 1) / kInvThreads;
  const int s0 = threadIdx.x * chunk, s1 = min(N, s0 + chunk);
  int local = 0;
  for (int s = s0; s < s1; ++s) local += cnt[s];
  part[threadIdx.x] = local;
  __syncthreads();
  for (int off = 1; off < kInvThreads; off <<= 1) {       // Hillis-Steele inclusive scan of the chunk sums
    const int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = threadIdx.x ? part[threadIdx.x - 1] : 0;
  for (int s = s0; s < s1; ++s) {
    const int c = cnt[s];
    cnt[s] = run;                    // becomes the fill cursor
    offsets[s] = run;
    run += c;
  }
  if (threadIdx.x == 0) offsets[N] = E;
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += kInvThreads) ent[atomicAdd(&cnt[idx[e]], 1)] = e;
  __syncthreads();                   // (same workgroup: the entries written above are visible below)
  // fixed summation order: ascending entry number inside every list (lists are short: E / N on average; lists beyond kSortMax
  // entries keep the order the atomics produced, i.e. their sum is correct but its rounding may differ from run to run; kNN
  // index sets stay far below the bound except the decoder's anchor lists, which use the register-table / one-hot forms instead)
  if (LDS_ENTRIES) {
    // one thread per ENTRY: its place in the list is the number of smaller entries -- independent LDS reads (a list of 30 is
    // 900 reads spread over 30 threads, not one thread's chain of 225 dependent moves), written straight to the output
    for (int i = threadIdx.x; i < E; i += kInvThreads) {
      const int v = ent[i], s = idx[v];
      const int lo = s ? cnt[s - 1] : 0, hi = cnt[s];          // every cursor ended at its list's end = the next list's start
      int rank = i - lo;
      if (hi - lo <= kSortMax) {
        rank = 0;
        for (int j = lo; j < hi; ++j) rank += ent[j] < v ? 1 : 0;
      }
      entries[lo + rank] = v;
    }
  } else {
    for (int s = threadIdx.x; s < N; s += kInvThreads) {
      const int lo = s ? cnt[s - 1] : 0, hi = cnt[s];
      if (hi - lo > kSortMax) continue;
      for (int i = lo + 1; i < hi; ++i) {      // (one thread, insertion sort in global memory: the form for E beyond the LDS)
        const int v = ent[i];
        int j = i - 1;
        while (j >= lo && ent[j] > v) {
          ent[j + 1] = ent[j];
          --j;
        }
        ent[j + 1] = v;
      }
    }
  }
}

struct bf16_t {
  unsigned short v;
};
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t *p) {
  const uint2 r = *reinterpret_cast<const uint2 *>(p);
  return make_float4(__builtin_bit_cast(float, r.x << 16), __builtin_bit_cast(float, r.x & 0xffff0000u),
                     __builtin_bit_cast(float, r.y << 16), __builtin_bit_cast(float, r.y & 0xffff0000u));
}

// out[b][s][c] = scale * sum over the list of s of src[b][entry][c]; d/4 lanes per source point
template <typename T>
__global__ __launch_bounds__(256) void segment_sum_rows_kernel(const T *__restrict__ src, const int32_t *__restrict__ offsets,
                                                               const int32_t *__restrict__ entries, int B, int E, int N,
                                                               int d, float scale, const float *__restrict__ addend,
                                                               float *__restrict__ out) {
  const int lpp = d >> 2, ppw = 64 / lpp;                  // lanes per point, points per wave
  const int lane = threadIdx.x & 63;
  const int sub = lane / lpp, cq = lane - sub * lpp;
  if (sub >= ppw) return;
  const long long wave = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const long long pt = wave * ppw + sub;                   // flattened (b, s)
  if (pt >= static_cast<long long>(B) * N) return;
  const int b = static_cast<int>(pt / N), s = static_cast<int>(pt - static_cast<long long>(b) * N);
  const int32_t *off = offsets + static_cast<long long>(b) * (N + 1);
  const int32_t *ent = entries + static_cast<long long>(b) * E;
  const T *base = src + static_cast<long long>(b) * E * d + 4 * cq;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int lo = off[s], hi = off[s + 1];
  int i = lo;
  for (; i + 4 <= hi; i += 4) {                            // four rows in flight (the decoder's lists are ~570 long)
    const float4 a = ld4(base + static_cast<long long>(ent[i]) * d);
    const float4 c = ld4(base + static_cast<long long>(ent[i + 1]) * d);
    const float4 e = ld4(base + static_cast<long long>(ent[i + 2]) * d);
    const float4 f = ld4(base + static_cast<long long>(ent[i + 3]) * d);
    acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += c.w;
    acc.x += e.x; acc.y += e.y; acc.z += e.z; acc.w += e.w;
    acc.x += f.x; acc.y += f.y; acc.z += f.z; acc.w += f.w;
  }
  for (; i < hi; ++i) {
    const float4 a = ld4(base + static_cast<long long>(ent[i]) * d);
    acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
  }
  float4 r = make_float4(scale * acc.x, scale * acc.y, scale * acc.z, scale * acc.w);
  if (addend) {      // (the caller's next statement was `out += addend`: one launch less per attention block's backward)
    const float4 a = *reinterpret_cast<const float4 *>(addend + pt * d + 4 * cq);
    r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w;
  }
  *reinterpret_cast<float4 *>(out + pt * d + 4 * cq) = r;
}

template <typename T>
int segment_sum_t(const T *src, const int32_t *offsets, const int32_t *entries, int B, int E, int N, int d, float scale,
                  float *out, void *stream, const float *addend = nullptr) {
  if (B <= 0 || N <= 0) return 0;
  NSDP_REQUIRE(src && offsets && entries && out, "segment_sum_rows: null pointer");
  NSDP_REQUIRE(d >= 4 && d % 4 == 0 && d <= 256, "segment_sum_rows: d=%d must be a multiple of 4 in [4, 256]", d);
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kAttnBwd, st, 0.0,
                          static_cast<double>(B) * (static_cast<double>(E) * (sizeof(T) * d + 4.0) + 4.0 * N * d));
  const int ppw = 64 / (d >> 2);
  const long long waves = (static_cast<long long>(B) * N + ppw - 1) / ppw;
  hipLaunchKernelGGL(segment_sum_rows_kernel<T>, dim3(static_cast<unsigned>((waves + 3) / 4)), dim3(256), 0, st, src, offsets,
                     entries, B, E, N, d, scale, addend, out);
  return nsdp::launch_status("segment_sum_rows_kernel");
}

}  // namespace

extern "C" {

int nsdp_knn_invert(const int32_t *idx, int B, int E, int N, int32_t *offsets, int32_t *entries, void *stream) {
  if (B <= 0 || E <= 0) return 0;
  NSDP_REQUIRE(idx && offsets && entries, "knn_invert: null pointer");
  NSDP_REQUIRE(N > 0 && N <= kMaxSources, "knn_invert: N=%d must be in [1, %d]", N, kMaxSources);
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kKnn, st, 0.0, static_cast<double>(B) * (12.0 * E + 4.0 * N));
  const size_t lds_cnt = (static_cast<size_t>(N) + 1) * sizeof(int), lds_all = lds_cnt + static_cast<size_t>(E) * sizeof(int);
  const bool in_lds = lds_all <= 148 * 1024;      // (+ 4 KiB of scan scratch: the CU's 160 KiB)
  const size_t lds = in_lds ? lds_all : lds_cnt;
  const void *fn = in_lds ? reinterpret_cast<const void *>(knn_invert_kernel<true>) : reinterpret_cast<const void *>(knn_invert_kernel<false>);
  if (lds > 60 * 1024) {      // (per call: the attribute belongs to the current device's function object)
    NSDP_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  }
  if (in_lds) hipLaunchKernelGGL(knn_invert_kernel<true>, dim3(B), dim3(kInvThreads), lds, st, idx, E, N, offsets, entries);
  else hipLaunchKernelGGL(knn_invert_kernel<false>, dim3(B), dim3(kInvThreads), lds, st, idx, E, N, offsets, entries);
  return nsdp::launch_status("knn_invert_kernel");
}

int nsdp_segment_sum_rows(const float *src, const int32_t *offsets, const int32_t *entries, int B, int E, int N, int d,
                          float scale, float *out, void *stream) {
  return segment_sum_t<float>(src, offsets, entries, B, E, N, d, scale, out, stream);
}

int nsdp_segment_sum_rows_add(const float *src, const int32_t *offsets, const int32_t *entries, int B, int E, int N, int d,
                              float scale, const float *addend, float *out, void *stream) {
  return segment_sum_t<float>(src, offsets, entries, B, E, N, d, scale, out, stream, addend);
}

int nsdp_segment_sum_rows_add_bf16(const void *src, const int32_t *offsets, const int32_t *entries, int B, int E, int N, int d,
                                   float scale, const float *addend, float *out, void *stream) {
  return segment_sum_t<bf16_t>(reinterpret_cast<const bf16_t *>(src), offsets, entries, B, E, N, d, scale, out, stream, addend);
}

int nsdp_segment_sum_rows_bf16(const void *src, const int32_t *offsets, const int32_t *entries, int B, int E, int N, int d,
                               float scale, float *out, void *stream) {
  return segment_sum_t<bf16_t>(reinterpret_cast<const bf16_t *>(src), offsets, entries, B, E, N, d, scale, out, stream);
}

}  // extern "C"
