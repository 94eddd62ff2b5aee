// The bf16x3 dense-layer kernel (x3_kernel.h, design notes in gemm_bf16x3.hip) over tensors in the G16 layout
//
//     T[M, C]  ->  [M / 16][C / 4][16 rows][4 floats]          (M % 16 == 0, C % 4 == 0; same bytes, permuted)
//
// -- the layout of the wide intermediates that only the dense-layer kernels themselves touch: the hidden layer of every
// Linear -> ReLU -> Linear pair over [rows, d] per-(centre, neighbour) tensors (reference model/encoder/blocks.py:86-124,
// model/decoder/blocks.py:30-91: fc_gamma; ResnetBlockFC :99-142) and its gradient.  Each is written by one GEMM, read by the
// next GEMM, by one weight gradient (wgrad_bf16x3.hip takes the same layout) and once more as a ReLU mask.  In this layout
// every wave-wide activation load and every accumulator tile's store is ONE contiguous KiB (row-major: 16 runs of 64 B at an
// 800- or 1024-byte pitch), the output needs no staging through LDS, and a 200-wide tensor needs no padding (50 quads).
// Same arithmetic, element for element, as the row-major kernels: results are bit-identical after un-permuting
// (tests/test_g16_gpu.py).
#include "x3_kernel.h"

namespace {

// form selection: launch_x3's defaults (gemm_bf16x3.hip), minus the experiment knobs
template <int NT, int PRE, int LAY>
void launch_g16(const X3Params &p, hipStream_t st) {
  constexpr int MT1 = NT >= 16 ? 2 : NT >= 13 ? 3 : 4;
  if constexpr (PRE == 1 || PRE == 4) {
    if constexpr (NT <= 8) launch_x3_pre<2, NT, PRE, 4, true, 2, 0, LAY>(p, st, 2);
    else if constexpr (PRE == 4 && NT == 13) {
      // ReLU bits leave the register path room for 13 n tiles (the fp32 mask's raw float4 did not).  Two 4-wave workgroups per
      // CU are 7 % faster ALONE (1421 -> 1323 us at 1.8 M rows with a residual, bit-identical) and 0.3 ms SLOWER in the B = 32
      // step, where this launch runs beside the side stream's weight gradient (three interleaved rounds: 37.53 / 37.60 / 37.65
      // against 37.25 / 37.33 / 37.33 ms): the one-workgroup form stays; nsdp_debug_set(6, 8192) selects the other (A/B)
      if (nsdp::g_x3_dbg & 8192) launch_x3_pre<2, 13, 4, 4, true, 2, 0, LAY>(p, st, 2);
      else launch_x3_pre<MT1, NT, PRE, 4, false, 2, 0, LAY>(p, st);
    } else launch_x3_pre<MT1, NT, PRE, 4, false, 2, 0, LAY>(p, st);
  } else if constexpr (NT <= 8) {
    if (p.K <= 128) launch_x3_pre<2, NT, PRE, 8, false, 4, 0, LAY>(p, st);      // weight planes resident in LDS
    else launch_x3_pre<2, NT, PRE, 4, false, 2, 0, LAY>(p, st, 2);
  } else if constexpr (NT == 13) {
    // (launch_x3 takes the register-path two-workgroup form up to 0.5 M rows; its PLAIN-prologue G16 instances are not built:
    // the compiler parks hand-issued weight-fragment loads in AGPRs there -- tests/test_no_inflight_spills.py -- so those
    // launches take the 8-wave form, 8-19 % slower at these sizes, which only a B <= 8 decoder reaches)
    if constexpr (PRE == 2) {
      if (p.M <= (1 << 19)) launch_x3_pre<2, 13, PRE, 4, true, 2, 0, LAY>(p, st, 2);
      else launch_x3_pre<2, 13, PRE, 8, false, 2, 0, LAY>(p, st);
    } else {
      launch_x3_pre<2, 13, PRE, 8, false, 2, 0, LAY>(p, st);
    }
  } else {
    const long long cus = nsdp::num_cus();
    const long long r3 = ((p.M + 191) / 192 + cus - 1) / cus * 192, r2 = ((p.M + 127) / 128 + cus - 1) / cus * 128;
    if (r2 * 108 < r3 * 100) launch_x3_pre<2, NT, PRE, 4, false, 2, 0, LAY>(p, st);
    else launch_x3_pre<3, NT, PRE, 4, false, 2, 0, LAY>(p, st);
  }
}

template <int NT>
int launch_g16_nt(const X3Params &p, int layout, hipStream_t st) {
  const int pre = p.bits ? 4 : p.mask ? 1 : (p.relu_in ? 2 : 0);
  nsdp::prof::Scope scope(nsdp::prof::kLinearX3, st, 2.0 * p.M * p.N * p.K,
                          4.0 * (static_cast<double>(p.M) * (p.K + p.N) + static_cast<double>(p.N) * p.K));
  // instantiated: X in G16 with the plain / masked prologue (second layer forward, first layer dX);
  //               Y in G16 with the plain / ReLU prologue (first layer forward, second layer dX);
  //               X in G16 masked by ReLU BITS (first layer dX: one byte per row tile and k block instead of a float4 stream)
  if (layout == kLayX && pre == 0) launch_g16<NT, 0, kLayX>(p, st);
  else if (layout == kLayX && pre == 1) launch_g16<NT, 1, kLayX>(p, st);
  else if (layout == kLayX && pre == 4) launch_g16<NT, 4, kLayX>(p, st);
  else if (layout == kLayY && pre == 0) launch_g16<NT, 0, kLayY>(p, st);
  else if (layout == kLayY && pre == 2) launch_g16<NT, 2, kLayY>(p, st);
  else {
    nsdp::set_error("linear_bf16x3_g16: layout=%d with %s is not an instantiated form", layout,
                    pre == 1 ? "a mask" : pre == 2 ? "an input ReLU" : "the plain prologue");
    return NSDP_EINVAL;
  }
  return nsdp::launch_status("linear_bf16x3_kernel (g16)");
}

// row-major [M, C] <-> G16, one float4 per thread (G16 float4 index: group * 4C + quad * 16 + row in group)
__global__ __launch_bounds__(256) void layout_g16_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, long long n4, int c4,
                                                         int to_g16) {
  const long long f = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (f >= n4) return;
  const long long per_group = 16LL * c4, grp = f / per_group;
  const int rem = static_cast<int>(f - grp * per_group), quad = rem >> 4, row = rem & 15;
  const long long rm = (grp * 16 + row) * c4 + quad;
  if (to_g16) dst[f] = src[rm];
  else dst[rm] = src[f];
}

}  // namespace

extern "C" {

// dst = src re-laid out: to_g16 != 0 row-major [M, C] -> G16, else back.  M % 16 == 0, C % 4 == 0; out of place.
int nsdp_layout_g16_f32(const float *src, float *dst, long long M, int C, int to_g16, void *stream) {
  if (M <= 0 || C <= 0) return 0;
  NSDP_REQUIRE(src && dst && src != dst, "layout_g16: null or aliased pointers");
  NSDP_REQUIRE(M % 16 == 0 && C % 4 == 0, "layout_g16: M=%lld must be a multiple of 16 and C=%d of 4", M, C);
  NSDP_REQUIRE(((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0, "layout_g16: operands must be 16-byte aligned");
  const long long n4 = M * (C / 4);
  hipLaunchKernelGGL(layout_g16_kernel, dim3(static_cast<unsigned>((n4 + 255) / 256)), dim3(256), 0, nsdp::as_stream(stream),
                     reinterpret_cast<const f32x4 *>(src), reinterpret_cast<f32x4 *>(dst), n4, C / 4, to_g16);
  return nsdp::launch_status("layout_g16_kernel");
}

int nsdp_linear_bf16x3_g16_supported(long long M, int N, int K, int layout, int has_mask, int relu_in) {
  if (!(M > 0 && M % 16 == 0 && K > 32 && K % 4 == 0 && N % 4 == 0 && N > 64 && N <= 256)) return 0;
  if (layout == kLayX) return !relu_in;
  if (layout == kLayY) return !has_mask;
  return 0;
}

size_t nsdp_relu_bits_bytes(long long M, int C) {
  if (M <= 0 || C <= 0) return 0;
  return static_cast<size_t>((M + 15) / 16) * static_cast<size_t>((C + 31) / 32) * 64;
}

int nsdp_linear_bf16x3_g16_f32(const float *X, const void *Wp, const float *bias, const float *residual, const float *mask,
                               const float *out_mask, const float *addend, float *Y, long long M, int N, int K, int relu_in,
                               int relu_out, int layout, const unsigned char *mask_bits, unsigned char *bits_out, void *stream) {
  if (M <= 0 || N <= 0) return 0;
  NSDP_REQUIRE(X && Wp && Y, "linear_bf16x3_g16: null pointer");
  NSDP_REQUIRE(nsdp_linear_bf16x3_g16_supported(M, N, K, layout, mask != nullptr || mask_bits != nullptr, relu_in),
               "linear_bf16x3_g16: unsupported call M=%lld N=%d K=%d layout=%d mask=%d relu_in=%d", M, N, K, layout,
               mask != nullptr || mask_bits != nullptr, relu_in);
  NSDP_REQUIRE(!(layout & kLayY) || (!out_mask && !addend && !residual),
               "linear_bf16x3_g16: a G16 output takes no out_mask / addend / row-major residual");
  NSDP_REQUIRE(!addend || ((mask || mask_bits) && out_mask && !relu_in), "linear_bf16x3_g16: an addend needs mask and out_mask, no input ReLU");
  NSDP_REQUIRE(!mask_bits || (layout == kLayX && !mask && !relu_in), "linear_bf16x3_g16: ReLU bits mask a G16 input (layout 1), instead of `mask`");
  NSDP_REQUIRE(!bits_out || (layout == kLayY && relu_out), "linear_bf16x3_g16: ReLU bits are written for a G16 output with an output ReLU");
  NSDP_REQUIRE(((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Wp) | reinterpret_cast<uintptr_t>(Y) |
                 reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(mask) |
                 reinterpret_cast<uintptr_t>(out_mask) | reinterpret_cast<uintptr_t>(addend)) & 15) == 0,
               "linear_bf16x3_g16: all operands must be 16-byte aligned");
  X3Params p{X, Wp, bias, residual, mask, out_mask, Y, M, N, K, relu_in, relu_out, nsdp::g_x3_dbg & ~(128 | 1024 | 2048)};
  p.addend = addend;
  p.bits = mask_bits;
  p.bits_out = bits_out;
  hipStream_t st = nsdp::as_stream(stream);
  const int nt = (N + 15) / 16;
  if (nt <= 8) return launch_g16_nt<8>(p, layout, st);
  if (nt <= 13) return launch_g16_nt<13>(p, layout, st);
  return launch_g16_nt<16>(p, layout, st);
}

}  // extern "C"
