// k-nearest-neighbour index search for gfx950 -- replaces the ATen sequence
// `square_distance(a, b).argsort()[:, :, :k]` of the reference
// (/root/reference/model/utils.py:39-55; call sites model/encoder/blocks.py:101-102, :287-288 and
// model/decoder/blocks.py:50-52), which materialises a [B,n,m,3] difference tensor, a [B,n,m] distance
// matrix and a full m-wide sort per row.
//
// MI355X design: one lane per query point; the source cloud streams through LDS in float4 tiles
// (coalesced global loads, broadcast ds_read_b128 per candidate); each lane keeps its k best
// (distance, index) pairs sorted in VGPRs (fully unrolled insertion, no scratch).  HBM traffic is the
// algorithmic minimum: (n+m)*12 bytes in, n*k*4 out.  Nothing of size n x m ever exists.
// Exactness: distance = ((dx*dx + dy*dy) + dz*dz), dx = query - source, one rounding per op
// (contraction off), bit-identical to the reference's torch.sum of squares; order = ascending
// (distance, index) (torch.argsort is unstable on CPU, so exact ties have no reference order).
#include <cfloat>

#include "common.h"
#include "prof.h"

#pragma clang fp contract(off)

namespace {

constexpr int kTile = 1024;  // source points per LDS tile (16 KiB)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K>
__global__ __launch_bounds__(256) void knn_kernel(const float *__restrict__ query_all,
                                                  const float *__restrict__ source_all, int n, int m,
                                                  int k, int32_t *__restrict__ idx_all,
                                                  float *__restrict__ dist_all) {
  __shared__ float4 tile[kTile];
  const int b = blockIdx.y;
  const float *query = query_all + static_cast<size_t>(b) * n * 3;
  const float *source = source_all + static_cast<size_t>(b) * m * 3;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool active = i < n;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (active) {
    qx = query[i * 3 + 0]; qy = query[i * 3 + 1]; qz = query[i * 3 + 2];
  }
  float bd[K];
  int bi[K];
#pragma unroll
  for (int t = 0; t < K; ++t) {
    bd[t] = FLT_MAX;
    bi[t] = 0;
  }
  // FLT_MAX sentinel: a real distance equal to FLT_MAX cannot displace it, inf/NaN inputs are out of
  // contract (the reference's argsort order for NaN is unspecified as well).
  for (int base = 0; base < m; base += kTile) {
    const int cnt = min(kTile, m - base);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += 256) {
      const float *p = source + static_cast<size_t>(base + t) * 3;
      tile[t] = make_float4(p[0], p[1], p[2], 0.f);
    }
    __syncthreads();
    for (int t = 0; t < cnt; ++t) {
      const float4 s = tile[t];
      const float d = nsdp::sq_dist3(qx, qy, qz, s.x, s.y, s.z);
      if (d < bd[K - 1]) {  // wave-level branch: skipped when no lane improves
        const int j = base + t;
#pragma unroll
        for (int u = K - 1; u > 0; --u) {
          const bool shift = d < bd[u - 1];
          const bool here = !shift && d < bd[u];
          const float nd = shift ? bd[u - 1] : (here ? d : bd[u]);
          const int ni = shift ? bi[u - 1] : (here ? j : bi[u]);
          bd[u] = nd;
          bi[u] = ni;
        }
        if (d < bd[0]) {
          bd[0] = d;
          bi[0] = j;
        }
      }
    }
  }
  if (active) {
    int32_t *io = idx_all + (static_cast<size_t>(b) * n + i) * k;
#pragma unroll
    for (int t = 0; t < K; ++t)
      if (t < k) io[t] = bi[t];
    if (dist_all) {
      float *dout = dist_all + (static_cast<size_t>(b) * n + i) * k;
#pragma unroll
      for (int t = 0; t < K; ++t)
        if (t < k) dout[t] = bd[t];
    }
  }
}

// Several lanes per query.  With one lane per query the kernel is one wave per SIMD at the encoder's sizes (2048
// queries x 32 shapes = 1024 waves) and the wave-level insertion branch fires in most iterations (any of 64 lanes
// improving runs the unrolled shift for all): 313 us per launch whatever the batch.  Here S lanes share a query, lane s
// scans every S-th part of each LDS tile with its own sorted top-K, and the S lists are merged through LDS by
// (distance, index) -- the same total order, S times the waves, 1/S of the iterations per wave.
template <int K, int S>
__global__ __launch_bounds__(256) void knn_split_kernel(const float *__restrict__ query_all,
                                                        const float *__restrict__ source_all, int n, int m, int k,
                                                        int32_t *__restrict__ idx_all, float *__restrict__ dist_all) {
  __shared__ float4 tile[kTile];
  __shared__ float md[256 * K];
  __shared__ int mi[256 * K];
  constexpr int kQ = 256 / S;                    // queries per workgroup
  const int b = blockIdx.y;
  const float *query = query_all + static_cast<size_t>(b) * n * 3;
  const float *source = source_all + static_cast<size_t>(b) * m * 3;
  const int ql = threadIdx.x / S, sub = threadIdx.x - ql * S;   // S consecutive lanes share a query
  const int i = blockIdx.x * kQ + ql;
  const bool active = i < n;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (active) {
    qx = query[i * 3 + 0]; qy = query[i * 3 + 1]; qz = query[i * 3 + 2];
  }
  float bd[K];
  int bi[K];
#pragma unroll
  for (int t = 0; t < K; ++t) {
    bd[t] = FLT_MAX;
    bi[t] = 0x7fffffff;
  }
  for (int base = 0; base < m; base += kTile) {
    const int cnt = min(kTile, m - base);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += 256) {
      const float *p = source + static_cast<size_t>(base + t) * 3;
      tile[t] = make_float4(p[0], p[1], p[2], 0.f);
    }
    __syncthreads();
    const int per = (cnt + S - 1) / S;           // lane `sub` scans tile entries [sub * per, (sub + 1) * per)
    const int t0 = sub * per, t1 = min(cnt, t0 + per);
    for (int t = t0; t < t1; ++t) {
      const float4 sp = tile[t];
      const float d = nsdp::sq_dist3(qx, qy, qz, sp.x, sp.y, sp.z);
      if (d < bd[K - 1]) {
        const int j = base + t;
#pragma unroll
        for (int u = K - 1; u > 0; --u) {
          const bool shift = d < bd[u - 1];
          const bool here = !shift && d < bd[u];
          const float nd = shift ? bd[u - 1] : (here ? d : bd[u]);
          const int ni = shift ? bi[u - 1] : (here ? j : bi[u]);
          bd[u] = nd;
          bi[u] = ni;
        }
        if (d < bd[0]) {
          bd[0] = d;
          bi[0] = j;
        }
      }
    }
  }
  // A lane's list is ascending in (distance, index): strict `<` keeps the earlier (lower) index first on ties, and a
  // lane meets its candidates in increasing index order.  S-way merge by (distance, index) on lane 0 of the query.
#pragma unroll
  for (int t = 0; t < K; ++t) {
    md[threadIdx.x * K + t] = bd[t];
    mi[threadIdx.x * K + t] = bi[t];
  }
  __syncthreads();
  if (active && sub == 0) {
    int head[S];
#pragma unroll
    for (int u = 0; u < S; ++u) head[u] = 0;
    int32_t *io = idx_all + (static_cast<size_t>(b) * n + i) * k;
    float *dout = dist_all ? dist_all + (static_cast<size_t>(b) * n + i) * k : nullptr;
    for (int t = 0; t < k; ++t) {
      float best_d = FLT_MAX;
      int best_i = 0x7fffffff, best_u = 0;
#pragma unroll
      for (int u = 0; u < S; ++u) {
        const int h = head[u];
        const float dd = h < K ? md[(threadIdx.x + u) * K + h] : FLT_MAX;
        const int ii = h < K ? mi[(threadIdx.x + u) * K + h] : 0x7fffffff;
        if (dd < best_d || (dd == best_d && ii < best_i)) {
          best_d = dd; best_i = ii; best_u = u;
        }
      }
#pragma unroll
      for (int u = 0; u < S; ++u) head[u] += (u == best_u) ? 1 : 0;
      io[t] = best_i;
      if (dout) dout[t] = best_d;
    }
  }
}

// The same search with the insertion DEFERRED.  In knn_split_kernel the wave-level branch around the unrolled insertion
// (~5 K VALU operations) is taken whenever ANY of the 64 lanes improves its list -- for 16 neighbours out of 512
// candidates per lane that is ~95 % of all iterations, although a single lane inserts only ~70 times.  Here a lane that
// sees a candidate below its (possibly stale, hence never too small) threshold only appends (distance, index) to a private
// FIFO of kQueue entries in LDS; the lists are updated when some lane's FIFO is nearly full and once at the end, by
// insertion rounds in which most lanes take part.  A lane still meets its candidates in increasing index order and the
// insertion is the same strict `<` chain, so the result -- ties included -- is the one of knn_split_kernel, bit for bit;
// per candidate the loop is now a distance, a compare and a predicated LDS append.  Measured (k = 16, four lanes per
// query): 500 x 2048 x 32 shapes 200 -> 111 us, 2048 x 8192 x 32 1227 -> 691 us, 4096 x 16384 x 16 1979 -> 1251 us; a
// 16-entry FIFO with the candidates taken four at a time (independent LDS reads and distances) against 8 entries one at a
// time: 111 against 125 us; eight lanes per query instead of four: slower (124 / 1096 / 2029 us, more list work).  With the
// tile as three planes + sentinels and the distances on packed fp32 operations (below): 92 / 458 / 767 us.
template <int K, int S, int kQueue>
__global__ __launch_bounds__(256) void knn_split_queue_kernel(const float *__restrict__ query_all,
                                                              const float *__restrict__ source_all, int n, int m, int k,
                                                              int32_t *__restrict__ idx_all, float *__restrict__ dist_all) {
  // the tile as three planes (four candidates = three 16-byte reads, two distances per packed operation) with 16 sentinel
  // entries behind the last point: x = FLT_MAX squares to +inf, which no threshold admits -- no range checks in the scan
  __shared__ __attribute__((aligned(16))) float tx[kTile + 16];
  __shared__ __attribute__((aligned(16))) float ty[kTile + 16];
  __shared__ __attribute__((aligned(16))) float tz[kTile + 16];
  __shared__ float md[256 * K];      // the FIFOs during the scan ([slot][thread]: conflict-free), the S lists afterwards
  __shared__ int mi[256 * K];
  static_assert(K >= kQueue, "the FIFOs live in the merge buffers");
  constexpr int kQ = 256 / S;
  const int b = blockIdx.y;
  const float *query = query_all + static_cast<size_t>(b) * n * 3;
  const float *source = source_all + static_cast<size_t>(b) * m * 3;
  const int ql = threadIdx.x / S, sub = threadIdx.x - ql * S;
  const int i = blockIdx.x * kQ + ql;
  const bool active = i < n;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (active) {
    qx = query[i * 3 + 0]; qy = query[i * 3 + 1]; qz = query[i * 3 + 2];
  }
  float bd[K];
  int bi[K];
#pragma unroll
  for (int t = 0; t < K; ++t) {
    bd[t] = FLT_MAX;
    bi[t] = 0x7fffffff;
  }
  int fill = 0;                      // entries in this lane's FIFO
  auto drain = [&]() __attribute__((always_inline)) {
    for (int sl = 0; sl < kQueue; ++sl) {
      const bool has = sl < fill;
      if (__builtin_amdgcn_ballot_w64(has) == 0) break;
      const float d = has ? md[sl * 256 + threadIdx.x] : FLT_MAX;
      const int j = mi[sl * 256 + threadIdx.x];
      if (d < bd[K - 1]) {
#pragma unroll
        for (int u = K - 1; u > 0; --u) {
          const bool shift = d < bd[u - 1];
          const bool here = !shift && d < bd[u];
          const float nd = shift ? bd[u - 1] : (here ? d : bd[u]);
          const int ni = shift ? bi[u - 1] : (here ? j : bi[u]);
          bd[u] = nd;
          bi[u] = ni;
        }
        if (d < bd[0]) {
          bd[0] = d;
          bi[0] = j;
        }
      }
    }
    fill = 0;
  };
  for (int base = 0; base < m; base += kTile) {
    const int cnt = min(kTile, m - base);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += 256) {
      const float *p = source + static_cast<size_t>(base + t) * 3;
      tx[t] = p[0]; ty[t] = p[1]; tz[t] = p[2];
    }
    if (threadIdx.x < 16) {
      tx[cnt + threadIdx.x] = FLT_MAX; ty[cnt + threadIdx.x] = 0.f; tz[cnt + threadIdx.x] = 0.f;
    }
    __syncthreads();
    // lane `sub` scans tile entries [sub * per, (sub + 1) * per), per a multiple of four (the same trip count in every
    // lane: the ballots see whole waves); S * per <= cnt + 4 S - 1: the sentinels cover what lies behind the tile.
    const int per = (((cnt + S - 1) / S) + 3) & ~3;
    const int t0 = sub * per;
    static_assert(S <= 4, "16 sentinel entries");
    for (int tt = 0; tt < per; tt += 4) {
      const int t = t0 + tt;
      const f32x4 X = *reinterpret_cast<const f32x4 *>(&tx[t]);
      const f32x4 Y = *reinterpret_cast<const f32x4 *>(&ty[t]);
      const f32x4 Z = *reinterpret_cast<const f32x4 *>(&tz[t]);
      // ((dx*dx + dy*dy) + dz*dz), dx = query - source, one rounding per operation (contraction is off in this file):
      // nsdp::sq_dist3 on two candidates per packed instruction
      const f32x2 q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
      f32x2 d01, d23;
      {
        const f32x2 dx = q2x - f32x2{X[0], X[1]}, dy = q2y - f32x2{Y[0], Y[1]}, dz = q2z - f32x2{Z[0], Z[1]};
        d01 = (dx * dx + dy * dy) + dz * dz;
      }
      {
        const f32x2 dx = q2x - f32x2{X[2], X[3]}, dy = q2y - f32x2{Y[2], Y[3]}, dz = q2z - f32x2{Z[2], Z[3]};
        d23 = (dx * dx + dy * dy) + dz * dz;
      }
      const float d[4] = {d01[0], d01[1], d23[0], d23[1]};
      const float thr = bd[K - 1];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (d[u] < thr) {
          md[fill * 256 + threadIdx.x] = d[u];
          mi[fill * 256 + threadIdx.x] = base + t + u;
          ++fill;
        }
      }
      if (__builtin_amdgcn_ballot_w64(fill > kQueue - 4) != 0) drain();
    }
  }
  drain();
  __syncthreads();                   // every wave is done with its FIFOs: the buffers become the merge lists
#pragma unroll
  for (int t = 0; t < K; ++t) {
    md[threadIdx.x * K + t] = bd[t];
    mi[threadIdx.x * K + t] = bi[t];
  }
  __syncthreads();
  if (active && sub == 0) {
    int head[S];
#pragma unroll
    for (int u = 0; u < S; ++u) head[u] = 0;
    int32_t *io = idx_all + (static_cast<size_t>(b) * n + i) * k;
    float *dout = dist_all ? dist_all + (static_cast<size_t>(b) * n + i) * k : nullptr;
    for (int t = 0; t < k; ++t) {
      float best_d = FLT_MAX;
      int best_i = 0x7fffffff, best_u = 0;
#pragma unroll
      for (int u = 0; u < S; ++u) {
        const int h = head[u];
        const float dd = h < K ? md[(threadIdx.x + u) * K + h] : FLT_MAX;
        const int ii = h < K ? mi[(threadIdx.x + u) * K + h] : 0x7fffffff;
        if (dd < best_d || (dd == best_d && ii < best_i)) {
          best_d = dd; best_i = ii; best_u = u;
        }
      }
#pragma unroll
      for (int u = 0; u < S; ++u) head[u] += (u == best_u) ? 1 : 0;
      io[t] = best_i;
      if (dout) dout[t] = best_d;
    }
  }
}

int g_knn_queue = 1;      // nsdp_debug_set(10, v), NSDP_KNN_QUEUE: 0 = the immediate-insertion kernel (A/B)

template <int K>
int launch(const float *q, const float *s, int B, int n, int m, int k, int32_t *idx, float *d2,
           hipStream_t st) {
  // enough queries to fill the chip with one lane each (the decoder: 8192 queries per shape), or a cloud too small to
  // split: the one-lane-per-query form
  const long long waves = (static_cast<long long>(n) + 255) / 256 * 4 * B;
  if constexpr (K <= 16) {
    if (m >= 256 && waves < 8LL * nsdp::num_cus()) {
      dim3 grid(nsdp::ceil_div(n, 64), B);
      if (g_knn_queue) {
        constexpr int kQueue = K >= 16 ? 16 : 8;
        NSDP_TRACE("knn_split_queue<%d,4,%d>", K, kQueue);
        hipLaunchKernelGGL((knn_split_queue_kernel<K, 4, kQueue>), grid, dim3(256), 0, st, q, s, n, m, k, idx, d2);
        return nsdp::launch_status("knn_split_queue_kernel");
      }
      NSDP_TRACE("knn_split<%d,4>", K);
      hipLaunchKernelGGL((knn_split_kernel<K, 4>), grid, dim3(256), 0, st, q, s, n, m, k, idx, d2);
      return nsdp::launch_status("knn_split_kernel");
    }
  }
  dim3 grid(nsdp::ceil_div(n, 256), B);
  NSDP_TRACE("knn<%d>", K);
  hipLaunchKernelGGL((knn_kernel<K>), grid, dim3(256), 0, st, q, s, n, m, k, idx, d2);
  return nsdp::launch_status("knn_kernel");
}

}  // namespace

namespace nsdp {
void debug_set_knn(int value) { g_knn_queue = value; }
}  // namespace nsdp

extern "C" int nsdp_knn(const float *query, const float *source, int B, int n, int m, int k,
                        int32_t *idx_out, float *dist2_out, void *stream) {
  if (B <= 0 || n <= 0 || k <= 0) return 0;
  NSDP_REQUIRE(query && source && idx_out, "knn: null pointer");
  NSDP_REQUIRE(k <= m, "knn: k=%d exceeds the number of source points m=%d", k, m);
  NSDP_REQUIRE(k <= 64, "knn: k=%d > 64 is not supported", k);
  NSDP_REQUIRE(B <= 65535, "knn: batch %d too large for one launch", B);
  hipStream_t st = nsdp::as_stream(stream);
  nsdp::prof::Scope scope(nsdp::prof::kKnn, st, 0.0,
                          static_cast<double>(B) * (12.0 * (n + m) + 4.0 * n * k * (dist2_out ? 2 : 1)));
  if (k <= 8) return launch<8>(query, source, B, n, m, k, idx_out, dist2_out, st);
  if (k <= 16) return launch<16>(query, source, B, n, m, k, idx_out, dist2_out, st);
  if (k <= 32) return launch<32>(query, source, B, n, m, k, idx_out, dist2_out, st);
  return launch<64>(query, source, B, n, m, k, idx_out, dist2_out, st);
}
