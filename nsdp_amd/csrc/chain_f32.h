// The transposed fp32 MFMA layer of the register-resident decoder chains (decoder_fused.hip: inference; decoder_train.hip:
// the training-mode forward of the cross attention): one wave, 16 rows, Y^T = W X^T on v_mfma_f32_16x16x4_f32 with the
// accumulator layout of layer L being the B-operand layout of layer L + 1, fragment-major weights prefetched through a
// register ring by hand-issued loads.  Included into the anonymous namespace of both translation units.
// (no include guard and no #include here: the including translation unit pulls in <type_traits> at global scope first)

using f32x4 = __attribute__((ext_vector_type(4))) float;

// v_out[ot] = act( W[ot*16 + ., :] * v_in + bias )  for NTOUT output tiles, NTIN input tiles.
// W row-major [NTOUT*16, NTIN*16] (zero padded); ACC: v_out is also the start value (fused residual add).
//
// The layer is a flat sequence of STEPS; a step feeds two independent accumulators with 4 MFMAs each
// (consecutive MFMAs on one accumulator would be separated by the 40-cycle dependent latency, longer than
// the 32-cycle issue interval; two chains keep the matrix pipe back to back).  For a pair of output tiles
// the two chains are the two tiles at the same k block; for the odd last tile they are the even and the odd
// k blocks of that tile, summed at the end.
//
// With the whole chain state in registers there is ONE wave per SIMD, so nothing but this wave hides the
// L2 latency of its weight fragments: they are fetched kPrefetch steps (kPrefetch x 256 matrix-pipe cycles)
// ahead into a register ring, and a scheduling barrier per step keeps the compiler from sinking the loads
// back down to their uses.
#ifndef NSDP_DEC_PREFETCH
#define NSDP_DEC_PREFETCH 6
#endif
constexpr int kPrefetch = NSDP_DEC_PREFETCH;
constexpr int kRing = 8;

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ float4 ldg4(const float *uniform_base, unsigned lane_byte_off) {
  return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(uniform_base) + lane_byte_off);
}

// Weight fragment loads are issued and awaited by hand: hipcc neither keeps a register-ring prefetch in
// place (it sinks the loads to their uses or hoists the MFMAs over them) nor can it wait for "all but the
// last N" loads across such a ring.  global_load with a uniform SGPR base + one 32-bit lane offset; the
// matching s_waitcnt takes the fragment registers as in/out operands so that every MFMA using them is
// data-dependent on the wait.  (vmcnt retires in order, so waiting until at most N younger loads are in
// flight is exact for N = the number of asm loads issued since; compiler-issued loads in between only make
// the wait more conservative.)
template <int IMM>
__device__ __forceinline__ void wload(f32x4 &dst, const float *uniform_base, unsigned lane_byte_off) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(lane_byte_off), "s"(uniform_base), "n"(IMM));
}
template <int N>
__device__ __forceinline__ void wwait(f32x4 &a, f32x4 &b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N>
__device__ __forceinline__ void wwait(f32x4 &a) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N));
}

template <int NTOUT, int NTIN>
struct Steps {
  static constexpr int kPairs = NTOUT / 2;
  static constexpr bool kOdd = (NTOUT & 1) != 0;
  static constexpr int kPairSteps = kPairs * NTIN;
  static constexpr int kTailSteps = kOdd ? (NTIN + 1) / 2 : 0;
  static constexpr int kSteps = kPairSteps + kTailSteps;
  static constexpr bool tail(int s) { return s >= kPairSteps; }
  static constexpr int tile_a(int s) { return tail(s) ? NTOUT - 1 : 2 * (s / NTIN); }
  static constexpr int tile_b(int s) { return tail(s) ? NTOUT - 1 : 2 * (s / NTIN) + 1; }
  static constexpr int kb_a(int s) { return tail(s) ? 2 * (s - kPairSteps) : s % NTIN; }
  static constexpr int kb_b(int s) { return tail(s) ? 2 * (s - kPairSteps) + 1 : s % NTIN; }
  static constexpr bool has_b(int s) { return kb_b(s) < NTIN; }
  static constexpr bool first(int s) { return tail(s) ? s == kPairSteps : s % NTIN == 0; }
  static constexpr bool last(int s) { return tail(s) ? s == kSteps - 1 : s % NTIN == NTIN - 1; }
  static constexpr int next_first(int s) {      // first step of the next accumulator group (kSteps: none)
    int sn = s + 1;
    while (sn < kSteps && !first(sn)) ++sn;
    return sn;
  }
  static constexpr int loads_after(int s, int depth) {   // asm loads issued for steps s+1 .. s+depth
    int n = 0;
    for (int t = s + 1; t <= s + depth && t < kSteps; ++t) n += has_b(t) ? 2 : 1;
    return n;
  }
};

__device__ __forceinline__ void pin(f32x4 &a, f32x4 &b) { asm volatile("" : "+a"(a), "+a"(b)); }

// BIASN > 0: `bias` holds only BIASN entries (a multiple of 4); fragments past its end read as zero.
template <int NTOUT, int NTIN, bool RELU_IN, bool RELU_OUT, bool ACC, int BIASN = 0>
__device__ __forceinline__ void dense(const float *__restrict__ W, const float *__restrict__ bias,
                                      const f32x4 *v_in, f32x4 *v_out, int li, int g) {
  auto ldbias = [&](int tile, unsigned bl) -> float4 {
    if constexpr (BIASN > 0) {
      if (tile * 16 + 4 * g + 4 > BIASN) return make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return ldg4(bias + tile * 16, bl);
  };
  using S = Steps<NTOUT, NTIN>;
  f32x4 x[NTIN];
#pragma unroll
  for (int kb = 0; kb < NTIN; ++kb) {
    x[kb] = v_in[kb];
    if (RELU_IN) {
      x[kb][0] = fmaxf(x[kb][0], 0.f); x[kb][1] = fmaxf(x[kb][1], 0.f);
      x[kb][2] = fmaxf(x[kb][2], 0.f); x[kb][3] = fmaxf(x[kb][3], 0.f);
    }
  }
  // W is fragment-major: [out tile][k block][lane = 16 g + li][4] -- each wave-wide load is one contiguous KiB
  // (row-major rows would make every 16-lane group touch 16 different cache lines for 16 B each)
  const unsigned wl = (16u * g + li) * 16u;
  const unsigned bl = 16u * g;
  f32x4 ra[kRing], rb[kRing];
  auto issue = [&](auto I) {
    constexpr int s = decltype(I)::value;
    wload<0>(ra[s % kRing], W + (S::tile_a(s) * NTIN + S::kb_a(s)) * 256, wl);
    if constexpr (S::has_b(s)) wload<0>(rb[s % kRing], W + (S::tile_b(s) * NTIN + S::kb_b(s)) * 256, wl);
  };
  constexpr int kPro = kPrefetch < S::kSteps ? kPrefetch : S::kSteps;
  static_for<0, kPro>(issue);
  float4 ba = ldbias(S::tile_a(0), bl);
  float4 bb = ldbias(S::tile_b(0), bl);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  static_for<0, S::kSteps>([&](auto I) {
    constexpr int s = decltype(I)::value;
    if constexpr (s + kPrefetch < S::kSteps) issue(std::integral_constant<int, s + kPrefetch>{});
    if constexpr (S::first(s)) {
      acc0 = ACC ? v_out[S::tile_a(s)] : f32x4{0.f, 0.f, 0.f, 0.f};
      acc0[0] += ba.x; acc0[1] += ba.y; acc0[2] += ba.z; acc0[3] += ba.w;
      if constexpr (S::tail(s)) {
        acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        acc1 = ACC ? v_out[S::tile_b(s)] : f32x4{0.f, 0.f, 0.f, 0.f};
        acc1[0] += bb.x; acc1[1] += bb.y; acc1[2] += bb.z; acc1[3] += bb.w;
      }
      constexpr int sn = S::next_first(s);          // bias of the next accumulator group, one group ahead
      if constexpr (sn < S::kSteps) {
        ba = ldbias(S::tile_a(sn), bl);
        bb = ldbias(S::tile_b(sn), bl);
      }
    }
    constexpr int kYounger = S::loads_after(s, kPrefetch);
    const f32x4 xa = x[S::kb_a(s)];
    if constexpr (S::has_b(s)) {
      wwait<kYounger>(ra[s % kRing], rb[s % kRing]);
      const f32x4 a4 = ra[s % kRing], b4 = rb[s % kRing];
      const f32x4 xb = x[S::kb_b(s)];
      // MFMAs are pure values to the compiler: left alone it regroups the eight by accumulator (two dependent
      // runs of four) and even sinks a whole layer's MFMAs past every scheduling barrier down to the first
      // use of the result, leaving the prefetched fragments to be spilled.  An empty volatile asm over the two
      // accumulators after each independent pair pins them (volatile asms keep their order).
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[0], xa[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b4[0], xb[0], acc1, 0, 0, 0);
      pin(acc0, acc1);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[1], xa[1], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b4[1], xb[1], acc1, 0, 0, 0);
      pin(acc0, acc1);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[2], xa[2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b4[2], xb[2], acc1, 0, 0, 0);
      pin(acc0, acc1);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[3], xa[3], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b4[3], xb[3], acc1, 0, 0, 0);
    } else {
      wwait<kYounger>(ra[s % kRing]);
      const f32x4 a4 = ra[s % kRing];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[0], xa[0], acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[1], xa[1], acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[2], xa[2], acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[3], xa[3], acc0, 0, 0, 0);
    }
    if constexpr (S::last(s)) {
      if constexpr (S::tail(s)) { acc0[0] += acc1[0]; acc0[1] += acc1[1]; acc0[2] += acc1[2]; acc0[3] += acc1[3]; }
      if (RELU_OUT) {
        acc0[0] = fmaxf(acc0[0], 0.f); acc0[1] = fmaxf(acc0[1], 0.f); acc0[2] = fmaxf(acc0[2], 0.f); acc0[3] = fmaxf(acc0[3], 0.f);
        acc1[0] = fmaxf(acc1[0], 0.f); acc1[1] = fmaxf(acc1[1], 0.f); acc1[2] = fmaxf(acc1[2], 0.f); acc1[3] = fmaxf(acc1[3], 0.f);
      }
      v_out[S::tile_a(s)] = acc0;
      if constexpr (!S::tail(s)) v_out[S::tile_b(s)] = acc1;
    }
    pin(acc0, acc1);
  });
}

