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
#include "x3_kernel.h"

namespace nsdp {
int g_x3_dbg = 0;
thread_local int g_x3_side_reserve = 0;
}  // namespace nsdp

namespace {

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
  constexpr int kNT8 = NT > 8 && NT <= 13 ? NT : 13;      // (the 8-wave form below is reached with 9 ... 13 n tiles only: no 16-tile instances)
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
    } else if (pre == 0) launch_x3_pre<2, kNT8, 0, 8, false, 2, GATHER>(p, st);
    else launch_x3_pre<2, kNT8, 2, 8>(p, st);
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
