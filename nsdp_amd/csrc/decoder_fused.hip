// Fused cross-attention decoder forward for gfx950 (inference / no-grad path).
//
// Reference: CrossTransformerDecoder.forward (model/decoder/crosstransformer_decoder.py:45-70) =
// CrossTransformerBlock (model/decoder/blocks.py:48-95) + init_enc + 5 x (fc_c, ResnetBlockFC) + fc_out,
// i.e. 18 dense layers and ~10 materialised [B,NQ,8,200] tensors per call in ATen, 11 kernel launches and
// 5 [R,200] HBM round trips in this repository's unfused path.
//
// Here one wave owns 16 query points and walks the whole decoder for them without leaving the register
// file.  Every dense layer is evaluated in the TRANSPOSED form  Y^T = W * X^T  on
// v_mfma_f32_16x16x4_f32: the weights are the A operand (16 output channels x 4 k per instruction), the
// activations the B operand (4 k x 16 query rows).  With the k-permuted fragment convention (lane group
// g = lane>>4 feeds k = 16*kb + 4g + s at step s) lane (row j = lane&15, g) holds, for every 16-channel
// tile t, exactly the channels 16t + 4g + {0..3} of its row -- which is ALSO the C/D layout of that MFMA
// (D row = 4g + reg, D col = lane&15).  So the accumulators of layer L are, register for register, the
// B fragments of layer L+1: no LDS round trip, no shuffles, no barriers between layers.
//
// Per neighbour slot (7 nearest anchors + the global token): rel -> delta0 (K = 3 + bias, one MFMA per
// tile) -> ReLU -> delta2 -> u = (q - k_anchor) + pos (table gather) -> gamma0 -> ReLU -> gamma2 -> online
// softmax update of the per-channel running (max, sum, weighted value) -- the softmax over the 8 tokens
// is per lane and per channel, purely in registers.  Then lat -> init_enc -> 5 x (fc_c + ResNet block) ->
// fc_out.  HBM traffic per query: 12 B in + 28 B of indices + 12 B out; weights (1.8 MB, padded) and the
// per-shape anchor tables come from L2.
#include <type_traits>
#include "common.h"
#include "prof.h"

namespace {

#include "chain_f32.h"

constexpr int DT = 13;   // 16-channel tiles of the attention width (200 -> 208)
constexpr int HT = 8;    // tiles of the MLP width (128)
constexpr int DP = DT * 16;
constexpr int HP = HT * 16;

struct DecParams {
  const float *xyz_q;      // [B,NQ,3]
  const float *anchors;    // [B,A,3]
  const int32_t *idx;      // [B,NQ,KN]
  const float *qk;         // [B,A,DP]   q - w_ks(anchor_feats), zero padded
  const float *vtab;       // [B,A,DP]   w_vs(anchor_feats)
  const float *a_g;        // [B,DP]     logits of the global token
  const float *v_g;        // [B,DP]
  const float *wd0;        // [DP,4]     fc_delta.0 weight | bias
  const float *wd2, *bd2;  // [DP,DP], [DP]
  const float *wg0, *bg0, *wg2, *bg2;
  const float *winit, *binit;   // [HP,DP], [HP]
  const float *wc, *bc;         // [5][HP,DP], [5][HP]
  const float *w0, *b0, *w1, *b1;  // [5][HP,HP], [5][HP]
  const float *wout, *bout;     // [16,HP], [16]
  float *out;              // [B,NQ,3]
  int B, NQ, A, KN;
};

struct Vec {                    // one activation vector per row: NT tiles x 4 channels per lane
  f32x4 t[DT];
};

constexpr int kWaves = 2;   // waves per workgroup: 2 x 39 KiB of private softmax state -> two workgroups per CU

__global__ __launch_bounds__(kWaves * 64) void decoder_fused_fwd_kernel(DecParams p) {
  // Architectural VGPRs stop at 256 per lane (the other 256 registers of the file are AGPRs, usable only
  // as MFMA operands), and the chain already keeps three 52-register activation vectors live.  The
  // per-channel online-softmax state (running max / sum / weighted value: 156 registers) therefore lives in
  // a wave-private LDS slab, laid out [quantity][tile][lane] as float4 = conflict-free ds_read/write_b128,
  // touched once per neighbour slot (78 LDS instructions against ~2100 MFMAs).  No barriers anywhere.
  __shared__ float4 state[kWaves][3][DT][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int b = blockIdx.y;
  const int q0 = (blockIdx.x * kWaves + wave) * 16;
  if (q0 >= p.NQ) return;                          // no barriers in this kernel
  int q = q0 + li;
  const bool qvalid = q < p.NQ;
  q = qvalid ? q : (p.NQ - 1);
  const size_t qrow = static_cast<size_t>(b) * p.NQ + q;

  const float qx = p.xyz_q[qrow * 3 + 0], qy = p.xyz_q[qrow * 3 + 1], qz = p.xyz_q[qrow * 3 + 2];
  const float *anch = p.anchors + static_cast<size_t>(b) * p.A * 3;
  const float *qkb = p.qk + static_cast<size_t>(b) * p.A * DP;
  const float *vtb = p.vtab + static_cast<size_t>(b) * p.A * DP;

  // online-softmax state per (tile, channel): running max, running sum, running weighted value.
  // It starts from the global token (logits a_g, value v_g, position encoding 0), model/decoder/blocks.py:73-86
  float4 (*S)[DT][64] = state[wave];
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    S[0][t][lane] = *reinterpret_cast<const float4 *>(p.a_g + static_cast<size_t>(b) * DP + t * 16 + 4 * g);
    S[1][t][lane] = make_float4(1.f, 1.f, 1.f, 1.f);
    S[2][t][lane] = *reinterpret_cast<const float4 *>(p.v_g + static_cast<size_t>(b) * DP + t * 16 + 4 * g);
  }

  for (int slot = 0; slot < p.KN; ++slot) {
    // The weights are loop-invariant, and LICM would hoist every one of the ~1000 weight-fragment loads of
    // an iteration out of the slot loop (thousands of live registers -> scratch spills).  Laundering the
    // base pointers through an opaque offset once per iteration makes the loads iteration-dependent again.
    // (an opaque zero offset, not the pointers themselves: those must keep their global address space)
    int opaque0 = 0;
    asm volatile("" : "+s"(opaque0));
    const float *wd0 = p.wd0 + opaque0, *wd2 = p.wd2 + opaque0, *bd2 = p.bd2 + opaque0, *wg0 = p.wg0 + opaque0,
                *bg0 = p.bg0 + opaque0, *wg2 = p.wg2 + opaque0, *bg2 = p.bg2 + opaque0;
    const int a = p.idx[qrow * p.KN + slot];
    // relative coordinate, augmented with 1 for the bias column: lane group g carries component g
    const float rel = g == 0 ? qx - anch[a * 3 + 0]
                    : g == 1 ? qy - anch[a * 3 + 1]
                    : g == 2 ? qz - anch[a * 3 + 2] : 1.0f;
    Vec va, vb, pos;
    // delta0: [DP x 4] * [4 x 16 rows], ReLU
#pragma unroll
    for (int ot = 0; ot < DT; ++ot) {
      const float w = wd0[(ot * 16 + li) * 4 + g];
      f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w, rel, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      acc[0] = fmaxf(acc[0], 0.f); acc[1] = fmaxf(acc[1], 0.f); acc[2] = fmaxf(acc[2], 0.f); acc[3] = fmaxf(acc[3], 0.f);
      va.t[ot] = acc;
    }
    dense<DT, DT, false, false, false>(wd2, bd2, va.t, pos.t, li, g);           // pos = delta2(h1)
    const float *qka = qkb + static_cast<size_t>(a) * DP + 4 * g;
#pragma unroll
    for (int t = 0; t < DT; ++t) {                                                   // u = (q - k_a) + pos
      const float4 k4 = *reinterpret_cast<const float4 *>(qka + t * 16);
      va.t[t] = f32x4{k4.x + pos.t[t][0], k4.y + pos.t[t][1], k4.z + pos.t[t][2], k4.w + pos.t[t][3]};
    }
    dense<DT, DT, false, true, false>(wg0, bg0, va.t, vb.t, li, g);             // h2 = relu(gamma0(u))
    dense<DT, DT, false, false, false>(wg2, bg2, vb.t, va.t, li, g);            // logits = gamma2(h2)
    const float *vta = vtb + static_cast<size_t>(a) * DP + 4 * g;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const float4 v4 = *reinterpret_cast<const float4 *>(vta + t * 16);
      const float sv[4] = {v4.x + pos.t[t][0], v4.y + pos.t[t][1], v4.z + pos.t[t][2], v4.w + pos.t[t][3]};
      const float4 m4 = S[0][t][lane], l4 = S[1][t][lane], y4 = S[2][t][lane];
      float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ll[4] = {l4.x, l4.y, l4.z, l4.w}, yy[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float av = va.t[t][c];
        const float mn = fmaxf(mm[c], av);
        const float sc = __expf(mm[c] - mn);
        const float w = __expf(av - mn);
        ll[c] = ll[c] * sc + w;
        yy[c] = yy[c] * sc + w * sv[c];
        mm[c] = mn;
      }
      S[0][t][lane] = make_float4(mm[0], mm[1], mm[2], mm[3]);
      S[1][t][lane] = make_float4(ll[0], ll[1], ll[2], ll[3]);
      S[2][t][lane] = make_float4(yy[0], yy[1], yy[2], yy[3]);
    }
  }
  Vec y;
#pragma unroll
  for (int t = 0; t < DT; ++t) {                                                     // lat = y / l
    const float4 l4 = S[1][t][lane], y4 = S[2][t][lane];
    y.t[t] = f32x4{y4.x / l4.x, y4.y / l4.y, y4.z / l4.z, y4.w / l4.w};
  }

  // MLP tail on [HP]-wide vectors (crosstransformer_decoder.py:63-69)
  f32x4 net[HT], h[HT];
  dense<HT, DT, false, false, false>(p.winit, p.binit, y.t, net, li, g);
#pragma unroll 1
  for (int i = 0; i < 5; ++i) {
    dense<HT, DT, false, false, true>(p.wc + static_cast<size_t>(i) * HP * DP, p.bc + i * HP, y.t, net, li, g);
    dense<HT, HT, true, false, false>(p.w0 + static_cast<size_t>(i) * HP * HP, p.b0 + i * HP, net, h, li, g);
    dense<HT, HT, true, false, true>(p.w1 + static_cast<size_t>(i) * HP * HP, p.b1 + i * HP, h, net, li, g);
  }
  f32x4 o[1];
  dense<1, HT, true, false, false>(p.wout, p.bout, net, o, li, g);
  if (g == 0 && qvalid) {   // output channels 0..2 live in lane group 0, registers 0..2
    float *dst = p.out + qrow * 3;
    dst[0] = o[0][0]; dst[1] = o[0][1]; dst[2] = o[0][2];
  }
}

}  // namespace

extern "C" int nsdp_decoder_fused_fwd(const float *xyz_q, const float *anchors, const int32_t *idx,
                                      const float *qk, const float *vtab, const float *a_g, const float *v_g,
                                      const float *const *weights, int n_weights, int B, int NQ, int A,
                                      int KN, int D, int H, float *out, void *stream) {
  if (static_cast<long long>(B) * NQ <= 0) return 0;
  NSDP_REQUIRE(D == 200 && H == 128, "decoder_fused_fwd: built for dim=200, hidden_dim=128 (got %d, %d)", D, H);
  NSDP_REQUIRE(n_weights == 17, "decoder_fused_fwd: expected 17 packed weight pointers, got %d", n_weights);
  NSDP_REQUIRE(xyz_q && anchors && idx && qk && vtab && a_g && v_g && weights && out, "decoder_fused_fwd: null pointer");
  NSDP_REQUIRE(B <= 65535, "decoder_fused_fwd: batch too large");
  DecParams p;
  p.xyz_q = xyz_q; p.anchors = anchors; p.idx = idx; p.qk = qk; p.vtab = vtab; p.a_g = a_g; p.v_g = v_g;
  p.wd0 = weights[0];
  p.wd2 = weights[1]; p.bd2 = weights[2];
  p.wg0 = weights[3]; p.bg0 = weights[4];
  p.wg2 = weights[5]; p.bg2 = weights[6];
  p.winit = weights[7]; p.binit = weights[8];
  p.wc = weights[9]; p.bc = weights[10];
  p.w0 = weights[11]; p.b0 = weights[12];
  p.w1 = weights[13]; p.b1 = weights[14];
  p.wout = weights[15]; p.bout = weights[16];
  p.out = out;
  p.B = B; p.NQ = NQ; p.A = A; p.KN = KN;
  hipStream_t st = nsdp::as_stream(stream);
  // algorithmic work: 2.484 MFLOP per query; bytes: coordinates, indices, output
  nsdp::prof::Scope scope(nsdp::prof::kDecoderFwd, st, 2.484e6 * static_cast<double>(B) * NQ,
                          static_cast<double>(B) * NQ * (24.0 + 4.0 * KN));
  hipLaunchKernelGGL(decoder_fused_fwd_kernel, dim3(nsdp::ceil_div(NQ, 16 * kWaves), B), dim3(kWaves * 64), 0, st, p);
  return nsdp::launch_status("decoder_fused_fwd_kernel");
}
