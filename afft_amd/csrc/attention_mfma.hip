// bf16 MFMA attention for the short sequences of the AFFT path (forward and backward).
//
//   SA-Fuser : L = M+1 modality tokens per frame -> G = 16/L frames are PACKED into one 16-row MFMA tile and a
//              block-diagonal mask (same-frame test) is applied in-register together with the reference's mask;
//   GPT-2    : L = T = 16 frames = exactly one tile, causal mask in-register;  T = 32 -> two tiles (NT = 2);
//   CA-Fuser : causal self / cross attention over T frames, q and k/v from different tensors.
//
// One workgroup (4 waves) per (packed sequence group, head).  Q, K, V (and dO) tiles [16*NT rows][hd] are staged
// once into LDS with 16-byte loads; every wave computes the small score tile S^T = K Q^T with
// v_mfma_f32_16x16x32_bf16 (keys land on registers, queries on lanes, so the softmax reduction over keys is
// in-lane + two cross-lane steps), masks and normalises in fp32 registers, and the four waves split the head
// dimension for the P*V product on v_mfma_f32_16x16x16_bf16, whose B operand is exactly the register layout the
// softmax left behind and whose A operand (V^T) comes from the hardware transposed LDS read ds_read_b64_tr_b16.
// Backward uses the same two layouts (P^T / P) so no tile is ever transposed through memory.
//
// HBM-bound by construction (attention is < 0.2 % of the path's FLOPs): q, k, v are read once and
// out written once per head; what this kernel buys over the generic VALU kernel (attention.hip) is the removal of
// ~100 us of latency-bound per-pair dot products per launch.
//
// Reference semantics: models/transformerblock.py:24-33,64-73 ; HF GPT-2 eager attention (causal, masked
// probabilities exactly 0).  probs output holds the PRE-dropout probabilities (see include/afft_hip.h).
#include "common.h"

namespace {

struct AttnArgs {
  const bf16_t *q, *k, *v, *dout;
  int64_t ldq, ldk, ldv, lddo;
  bf16_t *out, *dq, *dk, *dv;
  int64_t ldo, lddq, lddk, lddv;
  float* probs;        // fwd: written; bwd: read
  int nseq, L, H, hd, G;
  int hc;              // head-dimension chunk staged in LDS at a time (hd % hc == 0, hc % 64 == 0)
  float scale;
  int mask, period;
  unsigned dthresh, dkey;
  float dinv;
  const unsigned* salt;   // device word XOR-ed into dkey (afft_set_dropout_salt) or NULL
  // fp16x2 forward (PL instantiations): q / k / v / out are the HI planes of two-plane fp16 splits
  int64_t in_lo, out_lo;  // elements from a hi plane to its lo plane (inputs; output)
  bf16_t* out_b;          // optional bf16 copy of the output (what the bf16 backward reads)
  int64_t ldob;
  unsigned char* out_lo8; // optional: the output's lo part as an e4m3 byte plane (row pitch ldo bytes) instead of the fp16 plane
};

__device__ __forceinline__ bool pair_valid(int mask, int period, int L, int rows_valid, int qi, int kj) {
  if (qi >= rows_valid || kj >= rows_valid) return false;
  const int sq = qi / L, sk = kj / L;
  if (sq != sk) return false;                    // block-diagonal: tokens of different packed sequences never mix
  const int i = qi - sq * L, j = kj - sk * L;
  if (mask == AFFT_MASK_DIAG && i == j) return false;
  if (mask == AFFT_MASK_CAUSAL && j > i) return false;
  if (mask == AFFT_MASK_BLOCKCAUSAL && (j % period) > (i % period)) return false;   // T-SA-Fuser: causal T x T tiled
  return true;
}

// LDS tile [R][hd] bf16; 32-byte unit u of row r is stored at unit u ^ (r & 7): conflict-free transposed reads,
// 2-way (harmless here) ds_read_b128 row reads.
__device__ __forceinline__ int swz(int row, int row_bytes) {   // XOR stays inside the row: rows hold row_bytes/32 units
  return row & 7 & ((row_bytes >> 5) - 1);
}
__device__ __forceinline__ int tile_off(int row, int chunk16, int row_bytes) {
  return row * row_bytes + ((chunk16 ^ (swz(row, row_bytes) << 1)) << 4);
}

// stages columns [0, hd) of rows row0.. of src (the caller offsets src to the head and head-dimension chunk)
__device__ __forceinline__ void load_tile(const bf16_t* __restrict__ src, int64_t ld, int64_t row0, int rows_valid,
                                          int R, int hd, char* lds) {
  const int cpr = hd >> 3;  // 16-byte chunks per row
  for (int idx = threadIdx.x; idx < R * cpr; idx += 256) {
    const int row = idx / cpr, ch = idx - row * cpr;
    uint4 val = make_uint4(0u, 0u, 0u, 0u);
    if (row < rows_valid) val = *(const uint4*)(src + (row0 + row) * ld + ch * 8);
    *(uint4*)(lds + tile_off(row, ch, hd * 2)) = val;
  }
}

__device__ __forceinline__ bf16x8 row_frag(const char* lds, int row, int chunk16, int row_bytes) {
  return *(const bf16x8*)(lds + tile_off(row, chunk16, row_bytes));
}
// A operand of 16x16x16 for X^T: lane (g, i) gets tile[row0 + 4g + j][16*cb + i], j = 0..3
__device__ __forceinline__ bf16x4 tr_frag(const char* lds, int row0, int cb, int lane, int row_bytes) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int r = row0 + 4 * g + q;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (AFFT_LDS bf16x4*)(lds + r * row_bytes + ((cb ^ swz(r, row_bytes)) << 5) + p * 8));
}
__device__ __forceinline__ bf16x4 pack4(const float (&v)[4]) {
  bf16x4 r;
  r[0] = (short)f2bf(v[0]); r[1] = (short)f2bf(v[1]); r[2] = (short)f2bf(v[2]); r[3] = (short)f2bf(v[3]);
  return r;
}
__device__ __forceinline__ void store_o4(bf16_t* dst, const f32x4& a) {
  uint2 u;
  u.x = (unsigned)f2bf(a[0]) | ((unsigned)f2bf(a[1]) << 16);
  u.y = (unsigned)f2bf(a[2]) | ((unsigned)f2bf(a[3]) << 16);
  *(uint2*)dst = u;
}

typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) _Float16 h4;
__device__ __forceinline__ f32x4 mfma32_h(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16_h(bf16x4 a, h4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h4, a), b, c, 0, 0, 0);
}

// PL ("planes", the fp16x2 forward): q, k, v arrive as two-plane fp16 splits x = hi + lo and every product is accumulated from
// hi*hi + lo*hi + hi*lo on the fp16 MFMAs (exact to ~2^-21: the attention core adds no operand rounding of its own to a forward
// pass whose GEMMs carry their activations as hi + lo); the probabilities are split the same way in registers, and the output is
// written as planes again (+ the bf16 copy the backward pass reads).  PL = 2: q, k, v are ONE fp16 plane each (a.in_lo = 0: the sub-layer's
// AFFT_F16X2_ONE_PASS_ATTN site -- the operands carry one fp16 rounding like a weight does); the probabilities stay hi + lo, the output planes too.
template <int NT, int PL>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned dkey = a.dkey ^ (a.salt ? *a.salt : 0u);
  constexpr int R = 16 * NT;
  constexpr int NP = PL == 1 ? 2 : 1;  // planes per operand tile
  const int hd = a.hd, hc = a.hc, L = a.L, H = a.H;
  const int rb = hc * 2;               // bytes per LDS row: one head-dimension chunk
  const int nch = hd / hc;
  const int tb = R * rb;               // bytes of one tile
  char* Qs = smem;                     // [NP planes][R][hc]
  char* Ks = Qs + NP * tb;
  char* Vs = nch == 1 ? Ks + NP * tb : smem;     // chunked: V chunks reuse the Q/K space after the scores are done
  const int grp = blockIdx.x / H, h = blockIdx.x % H;
  const int seq0 = grp * a.G;
  const int nsq = min(a.G, a.nseq - seq0);
  const int rows_valid = nsq * L;
  const int64_t row0 = (int64_t)seq0 * L;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  // S^T[key][query] : keys on (lane>>4, reg), queries on lane&15; accumulated over the head-dimension chunks
  f32x4 s[NT][NT];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) s[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < nch; ++c) {
    if (c) __syncthreads();            // the previous chunk has been consumed by every wave
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      load_tile(a.q + pl * a.in_lo + (int64_t)h * hd + c * hc, a.ldq, row0, rows_valid, R, hc, Qs + pl * tb);
      load_tile(a.k + pl * a.in_lo + (int64_t)h * hd + c * hc, a.ldk, row0, rows_valid, R, hc, Ks + pl * tb);
      if (nch == 1) load_tile(a.v + pl * a.in_lo + (int64_t)h * hd, a.ldv, row0, rows_valid, R, hc, Vs + pl * tb);
    }
    __syncthreads();
    for (int ks = 0; ks < hc / 32; ++ks) {
      const int ch = ks * 4 + (lane >> 4);
      bf16x8 kf[NT], qf[NT], kl[NT], ql[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        kf[t] = row_frag(Ks, t * 16 + (lane & 15), ch, rb);
        qf[t] = row_frag(Qs, t * 16 + (lane & 15), ch, rb);
        if constexpr (PL == 1) {
          kl[t] = row_frag(Ks + tb, t * 16 + (lane & 15), ch, rb);
          ql[t] = row_frag(Qs + tb, t * 16 + (lane & 15), ch, rb);
        }
      }
#pragma unroll
      for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
          if constexpr (PL == 1) {
            s[kt][qt] = mfma32_h(kl[kt], qf[qt], s[kt][qt]);       // small terms first
            s[kt][qt] = mfma32_h(kf[kt], ql[qt], s[kt][qt]);
            s[kt][qt] = mfma32_h(kf[kt], qf[qt], s[kt][qt]);
          } else if constexpr (PL == 2) {
            s[kt][qt] = mfma32_h(kf[kt], qf[qt], s[kt][qt]);
          } else {
            s[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt], qf[qt], s[kt][qt], 0, 0, 0);
          }
        }
    }
  }
  // masked softmax over keys, per query column
  bf16x4 pb[NT][NT];
  h4 ph[NT][NT], pq[NT][NT];       // PL: the probabilities as hi + lo fp16
  (void)pb; (void)ph; (void)pq;
#pragma unroll
  for (int qt = 0; qt < NT; ++qt) {
    const int qi = qt * 16 + (lane & 15);
    float m = -INFINITY;
    bool ok[NT][4];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kj = kt * 16 + 4 * (lane >> 4) + r;
        ok[kt][r] = pair_valid(a.mask, a.period, L, rows_valid, qi, kj);
        s[kt][qt][r] *= a.scale;
        if (ok[kt][r]) m = fmaxf(m, s[kt][qt][r]);
      }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = ok[kt][r] ? __expf(s[kt][qt][r] - m) : 0.f;
        s[kt][qt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    const int sq = qi / L, i = qi - sq * L;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      float pv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = s[kt][qt][r] * inv;
        pv[r] = p;
        if (ok[kt][r]) {
          const int kj = kt * 16 + 4 * (lane >> 4) + r;
          const int j = kj - sq * L;
          const int64_t pidx = ((((int64_t)(seq0 + sq)) * H + h) * L + i) * L + j;
          if (wave == 0 && a.probs) a.probs[pidx] = p;
          if (a.dthresh) pv[r] = drop_keep(dkey, (unsigned)pidx, a.dthresh) ? p * a.dinv : 0.f;
        }
      }
      if constexpr (PL) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { ph[kt][qt][r] = (_Float16)pv[r]; pq[kt][qt][r] = (_Float16)(pv[r] - (float)ph[kt][qt][r]); }
      } else {
        pb[kt][qt] = pack4(pv);
      }
    }
    // masked (but same-sequence) pairs must read as exactly 0 in probs: write them too
    if (wave == 0 && a.probs && qi < rows_valid) {
#pragma unroll
      for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kj = kt * 16 + 4 * (lane >> 4) + r;
          if (!ok[kt][r] && kj < rows_valid && kj / L == sq)
            a.probs[((((int64_t)(seq0 + sq)) * H + h) * L + i) * L + (kj - sq * L)] = 0.f;
        }
    }
  }
  // O^T[c][query] = sum_key V^T[c][key] P^T[key][query]; the 4 waves split the head dimension in 16-channel blocks
  for (int c = 0; c < nch; ++c) {
    if (nch > 1) {
      __syncthreads();                 // scores / previous V chunk done by every wave
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
        load_tile(a.v + pl * a.in_lo + (int64_t)h * hd + c * hc, a.ldv, row0, rows_valid, R, hc, Vs + pl * tb);
      __syncthreads();
    }
    for (int cb = wave; cb < hc / 16; cb += 4) {
      f32x4 o[NT];
#pragma unroll
      for (int qt = 0; qt < NT; ++qt) o[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        const bf16x4 vt = tr_frag(Vs, kt * 16, cb, lane, rb);
        if constexpr (PL == 1) {
          const bf16x4 vl = tr_frag(Vs + tb, kt * 16, cb, lane, rb);
#pragma unroll
          for (int qt = 0; qt < NT; ++qt) {
            o[qt] = mfma16_h(vl, ph[kt][qt], o[qt]);
            o[qt] = mfma16_h(vt, pq[kt][qt], o[qt]);
            o[qt] = mfma16_h(vt, ph[kt][qt], o[qt]);
          }
        } else if constexpr (PL == 2) {
#pragma unroll
          for (int qt = 0; qt < NT; ++qt) {
            o[qt] = mfma16_h(vt, pq[kt][qt], o[qt]);
            o[qt] = mfma16_h(vt, ph[kt][qt], o[qt]);
          }
        } else {
#pragma unroll
          for (int qt = 0; qt < NT; ++qt) o[qt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vt, pb[kt][qt], o[qt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int qt = 0; qt < NT; ++qt) {
        const int qi = qt * 16 + (lane & 15);
        if (qi < rows_valid) {
          const int64_t col = (int64_t)h * hd + c * hc + cb * 16 + 4 * (lane >> 4);
          if constexpr (PL) {
            const float ov[4] = {o[qt][0], o[qt][1], o[qt][2], o[qt][3]};
            if (a.out_lo8) store_split8<4>(a.out, a.out_lo8, (row0 + qi) * a.ldo + col, ov);
            else if (a.out_lo) store_split<4>(a.out, (row0 + qi) * a.ldo + col, a.out_lo, ov);
            else store4(a.out, (row0 + qi) * a.ldo + col, AFFT_F16, ov);
            if (a.out_b) store_o4(a.out_b + (row0 + qi) * a.ldob + col, o[qt]);
          } else {
            store_o4(a.out + (row0 + qi) * a.ldo + col, o[qt]);
          }
        }
      }
    }
  }
}

template <int NT>
__global__ __launch_bounds__(256) void attn_bwd_mfma_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned dkey = a.dkey ^ (a.salt ? *a.salt : 0u);
  constexpr int R = 16 * NT;
  const int hd = a.hd, hc = a.hc, L = a.L, H = a.H;
  const int rb = hc * 2;
  const int nch = hd / hc;
  // one chunk: Q, K, V, dO all resident; chunked: phase 1 uses (V, dO), phase 3 re-stages (Q, K, dO) per chunk
  char* T0 = smem;
  char* T1 = T0 + R * rb;
  char* T2 = T1 + R * rb;
  char* T3 = T2 + R * rb;
  char* Qs = T0;
  char* Ks = T1;
  char* Vs = nch == 1 ? T2 : T0;
  char* Ds = nch == 1 ? T3 : T1;       // phase 1 (chunked: phase 3 uses T2 for dO)
  const int grp = blockIdx.x / H, h = blockIdx.x % H;
  const int seq0 = grp * a.G;
  const int nsq = min(a.G, a.nseq - seq0);
  const int rows_valid = nsq * L;
  const int64_t row0 = (int64_t)seq0 * L;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c15 = lane & 15;

  // The saved probabilities in both register layouts (A: query on the lane, keys on registers; B: the transpose), loaded FIRST:
  // fp32 from global, ~2 us of latency that used to sit between the two MFMA phases of a workgroup and now runs under the staging
  // of the operand tiles (masked / out-of-range pairs read as 0 and carry dropout scale 0).
  float pA[NT][NT][4], pB[NT][NT][4];
#pragma unroll
  for (int qt = 0; qt < NT; ++qt)
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        {
          const int qi = qt * 16 + c15, kj = kt * 16 + 4 * g + r;
          const int sq = qi / L, i = qi - sq * L;
          pA[qt][kt][r] = pair_valid(a.mask, a.period, L, rows_valid, qi, kj)
                              ? a.probs[((((int64_t)(seq0 + sq)) * H + h) * L + i) * L + (kj - sq * L)] : 0.f;
        }
        {
          const int qi = qt * 16 + 4 * g + r, kj = kt * 16 + c15;
          const int sq = qi / L, i = qi - sq * L;
          pB[qt][kt][r] = pair_valid(a.mask, a.period, L, rows_valid, qi, kj)
                              ? a.probs[((((int64_t)(seq0 + sq)) * H + h) * L + i) * L + (kj - sq * L)] : 0.f;
        }
      }
  // dP in both layouts from the same fragments:  A: dP^T[key][query] (keys on regs) ; B: dP[query][key] (queries on regs)
  f32x4 dpa[NT][NT], dpb[NT][NT];
#pragma unroll
  for (int x = 0; x < NT; ++x)
#pragma unroll
    for (int y = 0; y < NT; ++y) { dpa[x][y] = f32x4{0.f, 0.f, 0.f, 0.f}; dpb[x][y] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  for (int c = 0; c < nch; ++c) {
    if (c) __syncthreads();
    if (nch == 1) {
      load_tile(a.q + (int64_t)h * hd, a.ldq, row0, rows_valid, R, hc, Qs);
      load_tile(a.k + (int64_t)h * hd, a.ldk, row0, rows_valid, R, hc, Ks);
    }
    load_tile(a.v + (int64_t)h * hd + c * hc, a.ldv, row0, rows_valid, R, hc, Vs);
    load_tile(a.dout + (int64_t)h * hd + c * hc, a.lddo, row0, rows_valid, R, hc, Ds);
    __syncthreads();
    for (int ks = 0; ks < hc / 32; ++ks) {
      const int ch = ks * 4 + g;
      bf16x8 vf[NT], df[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        vf[t] = row_frag(Vs, t * 16 + c15, ch, rb);
        df[t] = row_frag(Ds, t * 16 + c15, ch, rb);
      }
#pragma unroll
      for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
          dpa[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[kt], df[qt], dpa[kt][qt], 0, 0, 0);
          dpb[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df[qt], vf[kt], dpb[qt][kt], 0, 0, 0);
        }
    }
  }
  // layout A: query = qt*16 + c15 (lane), key = kt*16 + 4g + r (regs)
  bf16x4 dsa[NT][NT];   // dS^T * scale  (B operand for dQ)
#pragma unroll
  for (int qt = 0; qt < NT; ++qt) {
    const int qi = qt * 16 + c15;
    const int sq = qi / L, i = qi - sq * L;
    float p[NT][4], dp[NT][4];
    float dot = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kj = kt * 16 + 4 * g + r;
        float pv = 0.f, m = 0.f;
        if (pair_valid(a.mask, a.period, L, rows_valid, qi, kj)) {
          const int64_t pidx = ((((int64_t)(seq0 + sq)) * H + h) * L + i) * L + (kj - sq * L);
          pv = pA[qt][kt][r];
          m = (a.dthresh && !drop_keep(dkey, (unsigned)pidx, a.dthresh)) ? 0.f : a.dinv;
        }
        p[kt][r] = pv;
        dp[kt][r] = dpa[kt][qt][r] * m;
        dot += pv * dp[kt][r];
      }
    dot += __shfl_xor(dot, 16, 64);
    dot += __shfl_xor(dot, 32, 64);
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      float v4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v4[r] = p[kt][r] * (dp[kt][r] - dot) * a.scale;
      dsa[kt][qt] = pack4(v4);
    }
  }
  // layout B: query = qt*16 + 4g + r (regs), key = kt*16 + c15 (lane); one query tile at a time (registers)
  bf16x4 dsb[NT][NT];   // dS * scale  (B operand for dK)
  bf16x4 ppb[NT][NT];   // dropped-out P (B operand for dV)
#pragma unroll
  for (int qt = 0; qt < NT; ++qt) {
    float p[NT][4], dp[NT][4], pm[NT][4], dot[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = qt * 16 + 4 * g + r;
      const int sq = qi / L, i = qi - sq * L;
      float d = 0.f;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        const int kj = kt * 16 + c15;
        float pv = 0.f, m = 0.f;
        if (pair_valid(a.mask, a.period, L, rows_valid, qi, kj)) {
          const int64_t pidx = ((((int64_t)(seq0 + sq)) * H + h) * L + i) * L + (kj - sq * L);
          pv = pB[qt][kt][r];
          m = (a.dthresh && !drop_keep(dkey, (unsigned)pidx, a.dthresh)) ? 0.f : a.dinv;
        }
        p[kt][r] = pv;
        pm[kt][r] = pv * m;                   // dropped-out probability P' (for dV)
        dp[kt][r] = dpb[qt][kt][r] * m;       // dP = dP' * m/(1-p)
        d += pv * dp[kt][r];
      }
      // sum over keys = over the 16 lanes of this lane group
      d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64); d += __shfl_xor(d, 8, 64);
      dot[r] = d;
    }
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      float v4[4], pd[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v4[r] = p[kt][r] * (dp[kt][r] - dot[r]) * a.scale;
        pd[r] = pm[kt][r];
      }
      dsb[qt][kt] = pack4(v4);
      ppb[qt][kt] = pack4(pd);
    }
  }
  // the three [hc x 16] products per head-dimension chunk, 16 channels per step, waves split the chunk
  for (int c = 0; c < nch; ++c) {
    if (nch > 1) {
      Ds = T2;
      __syncthreads();                 // phase 1 / the previous chunk has been consumed by every wave
      load_tile(a.q + (int64_t)h * hd + c * hc, a.ldq, row0, rows_valid, R, hc, Qs);
      load_tile(a.k + (int64_t)h * hd + c * hc, a.ldk, row0, rows_valid, R, hc, Ks);
      load_tile(a.dout + (int64_t)h * hd + c * hc, a.lddo, row0, rows_valid, R, hc, Ds);
      __syncthreads();
    }
    for (int cb = wave; cb < hc / 16; cb += 4) {
      f32x4 odq[NT], odk[NT], odv[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) { odq[t] = f32x4{0.f, 0.f, 0.f, 0.f}; odk[t] = odq[t]; odv[t] = odq[t]; }
#pragma unroll
      for (int t = 0; t < NT; ++t) {          // reduction tile (keys for dQ, queries for dK / dV)
        const bf16x4 kT = tr_frag(Ks, t * 16, cb, lane, rb);
        const bf16x4 qT = tr_frag(Qs, t * 16, cb, lane, rb);
        const bf16x4 dT = tr_frag(Ds, t * 16, cb, lane, rb);
#pragma unroll
        for (int u = 0; u < NT; ++u) {        // output tile
          odq[u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kT, dsa[t][u], odq[u], 0, 0, 0);  // dQ^T[c][query u]
          odk[u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qT, dsb[t][u], odk[u], 0, 0, 0);  // dK^T[c][key u]
          odv[u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dT, ppb[t][u], odv[u], 0, 0, 0);  // dV^T[c][key u]
        }
      }
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int ri = u * 16 + c15;
        if (ri < rows_valid) {
          const int64_t col = (int64_t)h * hd + c * hc + cb * 16 + 4 * g;
          store_o4(a.dq + (row0 + ri) * a.lddq + col, odq[u]);
          store_o4(a.dk + (row0 + ri) * a.lddk + col, odk[u]);
          store_o4(a.dv + (row0 + ri) * a.lddv + col, odv[u]);
        }
      }
    }
  }
}

bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---------------------------------------------------------------------------------------------------------------------
// Round 5 backward: every wave owns a COLUMN SLICE of the head (SL = hd / 4 channels) from start to end.
//   * its slices of V, dO, Q, K are fetched with fully coalesced 16-byte loads issued back to back at kernel start (64 registers
//     of reads in flight per lane) and pass through two WAVE-PRIVATE LDS buffers -- no workgroup barrier for staging;
//   * dP = dO V^T is summed over the head dimension: each wave multiplies its own slice (a quarter of the MFMAs the redundant
//     form spent) and the four partial 16 x 16 tiles meet through 2 KiB of LDS -- the kernel's only workgroup barrier;
//   * the three output products run on the wave's slice (A operands by ds_read_tr from its buffers), and each result goes back
//     through the buffer that has just been consumed so that it is STORED as whole 256-byte row segments (the old kernel stored
//     8 bytes per lane in 32-byte runs);
//   * 8 KiB of LDS per wave: 40 KiB per workgroup, four workgroups per CU instead of two.
// hd % 128 == 0, hd <= 512, L <= 32 (NT <= 2); everything else takes attn_bwd_mfma_kernel.
__device__ __forceinline__ void wave_lds_sync() {      // LDS operations of ONE wave execute in order: drain, and keep the compiler from reordering across
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

template <int NT>
__global__ __launch_bounds__(256) void attn_bwd_sliced_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned dkey = a.dkey ^ (a.salt ? *a.salt : 0u);
  constexpr int R = 16 * NT;
  const int hd = a.hd, L = a.L, H = a.H;
  const int SL = hd >> 2;                // channels per wave
  const int rb = SL * 2;                 // bytes per LDS row
  const int cpr = SL >> 3;               // 16-byte chunks per row
  const int grp = blockIdx.x / H, h = blockIdx.x % H;
  const int seq0 = grp * a.G;
  const int nsq = min(a.G, a.nseq - seq0);
  const int rows_valid = nsq * L;
  const int64_t row0 = (int64_t)seq0 * L;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c15 = lane & 15;
  const int64_t c0 = (int64_t)h * hd + wave * SL;        // first channel of this wave's slice
  char* bufA = smem + wave * (2 * R * rb);               // wave-private: V, later K, later output staging
  char* bufB = bufA + R * rb;                            //               dO, later Q, later output staging
  float* red = (float*)(smem + 4 * (2 * R * rb));        // [4 waves][2 NT^2][64 lanes] f32x4: the partial dP tiles
  constexpr int NP = (R * 16 + 63) / 64;                 // 16-byte pieces per lane of a slice at SL = 128 (fewer lanes / pieces are live below)
  const int npieces = R * cpr;

  // ---- all reads of the kernel, back to back: probabilities (fp32, both register layouts) and the four operand slices
  float pA[NT][NT][4], pB[NT][NT][4];
#pragma unroll
  for (int qt = 0; qt < NT; ++qt)
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        {
          const int qi = qt * 16 + c15, kj = kt * 16 + 4 * g + r;
          const int sq = qi / L, i = qi - sq * L;
          pA[qt][kt][r] = pair_valid(a.mask, a.period, L, rows_valid, qi, kj)
                              ? a.probs[((((int64_t)(seq0 + sq)) * H + h) * L + i) * L + (kj - sq * L)] : 0.f;
        }
        {
          const int qi = qt * 16 + 4 * g + r, kj = kt * 16 + c15;
          const int sq = qi / L, i = qi - sq * L;
          pB[qt][kt][r] = pair_valid(a.mask, a.period, L, rows_valid, qi, kj)
                              ? a.probs[((((int64_t)(seq0 + sq)) * H + h) * L + i) * L + (kj - sq * L)] : 0.f;
        }
      }
  uint4 rv[NP], rd[NP], rk[NP], rq[NP];
  auto fetch = [&](const bf16_t* __restrict__ src, int64_t ld, uint4 (&dst)[NP]) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int p = lane + 64 * j;
      const int row = p / cpr, ch = p - row * cpr;
      dst[j] = make_uint4(0u, 0u, 0u, 0u);
      if (p < npieces && row < rows_valid) dst[j] = *(const uint4*)(src + (row0 + row) * ld + c0 + ch * 8);
    }
  };
  auto put = [&](char* buf, const uint4 (&srcr)[NP]) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int p = lane + 64 * j;
      const int row = p / cpr, ch = p - row * cpr;
      if (p < npieces) *(uint4*)(buf + tile_off(row, ch, rb)) = srcr[j];
    }
  };
  fetch(a.v, a.ldv, rv);
  fetch(a.dout, a.lddo, rd);
  fetch(a.k, a.ldk, rk);
  fetch(a.q, a.ldq, rq);

  // ---- phase 1: this wave's share of dP (both layouts) from its slices of V and dO
  put(bufA, rv);
  put(bufB, rd);
  wave_lds_sync();
  f32x4 dpa[NT][NT], dpb[NT][NT];
#pragma unroll
  for (int x = 0; x < NT; ++x)
#pragma unroll
    for (int y = 0; y < NT; ++y) { dpa[x][y] = f32x4{0.f, 0.f, 0.f, 0.f}; dpb[x][y] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  for (int ks = 0; ks < SL / 32; ++ks) {
    const int ch = ks * 4 + g;
    bf16x8 vf[NT], df[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      vf[t] = row_frag(bufA, t * 16 + c15, ch, rb);
      df[t] = row_frag(bufB, t * 16 + c15, ch, rb);
    }
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int qt = 0; qt < NT; ++qt) {
        dpa[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[kt], df[qt], dpa[kt][qt], 0, 0, 0);
        dpb[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df[qt], vf[kt], dpb[qt][kt], 0, 0, 0);
      }
  }
  // the four partial tiles meet (fixed order: wave 0 + 1 + 2 + 3, the same bits on every run)
  {
    f32x4* mine = (f32x4*)red + (wave * 2 * NT * NT) * 64 + lane;
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
      for (int y = 0; y < NT; ++y) {
        mine[((x * NT + y) * 2 + 0) * 64] = dpa[x][y];
        mine[((x * NT + y) * 2 + 1) * 64] = dpb[x][y];
      }
    __syncthreads();
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
      for (int y = 0; y < NT; ++y) {
        f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, sb = sa;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const f32x4* o = (const f32x4*)red + (w * 2 * NT * NT) * 64 + lane;
          sa += o[((x * NT + y) * 2 + 0) * 64];
          sb += o[((x * NT + y) * 2 + 1) * 64];
        }
        dpa[x][y] = sa;
        dpb[x][y] = sb;
      }
  }

  // ---- softmax backward in both layouts (as attn_bwd_mfma_kernel)
  bf16x4 dsa[NT][NT];   // dS^T * scale  (B operand for dQ)
#pragma unroll
  for (int qt = 0; qt < NT; ++qt) {
    const int qi = qt * 16 + c15;
    const int sq = qi / L, i = qi - sq * L;
    float p[NT][4], dp[NT][4];
    float dot = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kj = kt * 16 + 4 * g + r;
        float pv = 0.f, m = 0.f;
        if (pair_valid(a.mask, a.period, L, rows_valid, qi, kj)) {
          const int64_t pidx = ((((int64_t)(seq0 + sq)) * H + h) * L + i) * L + (kj - sq * L);
          pv = pA[qt][kt][r];
          m = (a.dthresh && !drop_keep(dkey, (unsigned)pidx, a.dthresh)) ? 0.f : a.dinv;
        }
        p[kt][r] = pv;
        dp[kt][r] = dpa[kt][qt][r] * m;
        dot += pv * dp[kt][r];
      }
    dot += __shfl_xor(dot, 16, 64);
    dot += __shfl_xor(dot, 32, 64);
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      float v4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v4[r] = p[kt][r] * (dp[kt][r] - dot) * a.scale;
      dsa[kt][qt] = pack4(v4);
    }
  }
  bf16x4 dsb[NT][NT];   // dS * scale  (B operand for dK)
  bf16x4 ppb[NT][NT];   // dropped-out P (B operand for dV)
#pragma unroll
  for (int qt = 0; qt < NT; ++qt) {
    float p[NT][4], dp[NT][4], pm[NT][4], dot[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = qt * 16 + 4 * g + r;
      const int sq = qi / L, i = qi - sq * L;
      float d = 0.f;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        const int kj = kt * 16 + c15;
        float pv = 0.f, m = 0.f;
        if (pair_valid(a.mask, a.period, L, rows_valid, qi, kj)) {
          const int64_t pidx = ((((int64_t)(seq0 + sq)) * H + h) * L + i) * L + (kj - sq * L);
          pv = pB[qt][kt][r];
          m = (a.dthresh && !drop_keep(dkey, (unsigned)pidx, a.dthresh)) ? 0.f : a.dinv;
        }
        p[kt][r] = pv;
        pm[kt][r] = pv * m;
        dp[kt][r] = dpb[qt][kt][r] * m;
        d += pv * dp[kt][r];
      }
      d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64); d += __shfl_xor(d, 8, 64);
      dot[r] = d;
    }
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      float v4[4], pd[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v4[r] = p[kt][r] * (dp[kt][r] - dot[r]) * a.scale;
        pd[r] = pm[kt][r];
      }
      dsb[qt][kt] = pack4(v4);
      ppb[qt][kt] = pack4(pd);
    }
  }

  // ---- the three products on this wave's slice; results leave through the buffer that was just consumed
  // out^T[c][u-th row tile] = sum_t X^T[c][rows of tile t] * Bop[t][u]
  auto product = [&](const char* xbuf, const bf16x4 (&bop)[NT][NT], char* stage, bf16_t* __restrict__ dst, int64_t ld) {
    const int ncb = SL >> 4;
    for (int cb = 0; cb < ncb; ++cb) {
      f32x4 o[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) o[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bf16x4 xT = tr_frag(xbuf, t * 16, cb, lane, rb);
#pragma unroll
        for (int u = 0; u < NT; ++u) o[u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xT, bop[t][u], o[u], 0, 0, 0);
      }
      // D[row = channel cb*16 + 4g + r][col = row index u*16 + c15] -> stage[row u*16 + c15][channels cb*16 + 4g .. +3] (8 bytes)
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        uint2 w2;
        w2.x = (unsigned)f2bf(o[u][0]) | ((unsigned)f2bf(o[u][1]) << 16);
        w2.y = (unsigned)f2bf(o[u][2]) | ((unsigned)f2bf(o[u][3]) << 16);
        *(uint2*)(stage + tile_off(u * 16 + c15, cb * 2 + (g >> 1), rb) + (g & 1) * 8) = w2;
      }
    }
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int p = lane + 64 * j;
      const int row = p / cpr, ch = p - row * cpr;
      if (p < npieces && row < rows_valid) *(uint4*)(dst + (row0 + row) * ld + c0 + ch * 8) = *(const uint4*)(stage + tile_off(row, ch, rb));
    }
    wave_lds_sync();       // the staging buffer is free again
  };
  // dV^T = dO^T P'      : reads bufB (dO), leaves through bufA (V is dead since phase 1)
  product(bufB, ppb, bufA, a.dv, a.lddv);
  // dQ^T = K^T dS^T     : K -> bufA, leaves through bufB (dO is dead now)
  put(bufA, rk);
  wave_lds_sync();
  product(bufA, dsa, bufB, a.dq, a.lddq);
  // dK^T = Q^T dS       : Q -> bufB, leaves through bufA
  put(bufB, rq);
  wave_lds_sync();
  product(bufB, dsb, bufA, a.dk, a.lddk);
}

}  // namespace

#ifndef AFFT_ATTN_PL_LDS_KB
#define AFFT_ATTN_PL_LDS_KB 48
#endif
// Returns 0 when launched, -1 when the shape is not handled by the MFMA path (caller falls back), >0 on error.
int afft_attention_mfma(bool backward, const void* dout, int64_t lddo, const void* q, int64_t ldq, const void* k,
                        int64_t ldk, const void* v, int64_t ldv, float* probs, int nseq, int L, int H, int hd,
                        float scale, int mask, float drop_p, unsigned drop_key, void* out, int64_t ldo, void* dq,
                        int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, hipStream_t stream,
                        int planes, int64_t in_lo, int64_t out_lo, void* out_b, int64_t ldob, void* out_lo8) {
  if (L > 64 || hd % 64 != 0 || hd > 1024) return -1;
  if (planes && (backward || in_lo % 8 || in_lo < 0 || out_lo % 4 || ldob % 4 || (((uintptr_t)out_b) & 7))) return -1;
  if (ldq % 8 || ldk % 8 || ldv % 8 || !al16(q) || !al16(k) || !al16(v)) return -1;
  if (!backward && (ldo % 4 || (((uintptr_t)out) & 7))) return -1;
  if (backward && (lddo % 8 || !al16(dout) || lddq % 4 || lddk % 4 || lddv % 4 || (((uintptr_t)dq) & 7) ||
                   (((uintptr_t)dk) & 7) || (((uintptr_t)dv) & 7) || !probs)) return -1;
  const int NT = L > 32 ? 4 : L > 16 ? 2 : 1;
  // the whole head dimension in LDS when it fits (3 tiles forward, 4 backward); else chunks of the head dimension,
  // the scores / dP accumulate over the chunks and the operand tiles are re-staged (2 tiles forward, 3 backward)
  int hc = hd;
  const int np = (planes && in_lo) ? 2 : 1;       // fp16x2 forward: every operand tile is two planes (in_lo = 0: the hi plane alone)
  size_t lds = (size_t)(backward ? 4 : 3) * np * 16 * NT * hd * 2;
  // planes (fp16x2 forward): the two-plane tiles of a whole head (96 KiB at hd = 512) leave ONE workgroup per CU, whose load -> barrier ->
  // compute -> store runs with nothing beside it (2.75 TB/s); chunks that fit 48 KiB keep three workgroups per CU in flight
  const size_t budget = (planes && !backward) ? (size_t)(AFFT_ATTN_PL_LDS_KB) * 1024 : (size_t)160 * 1024;
  if (lds > budget) {
    hc = 0;
    for (int cand = hd / 2; cand >= 64; cand /= 2)
      if (hd % cand == 0 && cand % 64 == 0 && (size_t)(backward ? 3 : 2) * np * 16 * NT * cand * 2 <= budget) { hc = cand; break; }
    if (!hc) return -1;
    lds = (size_t)(backward ? 3 : 2) * np * 16 * NT * hc * 2;
  }
  AttnArgs a;
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.dout = (const bf16_t*)dout;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.lddo = lddo;
  a.out = (bf16_t*)out; a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv;
  a.ldo = ldo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  a.probs = probs;
  a.in_lo = in_lo; a.out_lo = out_lo; a.out_b = (bf16_t*)out_b; a.ldob = ldob; a.out_lo8 = (unsigned char*)out_lo8;
  a.nseq = nseq; a.L = L; a.H = H; a.hd = hd; a.G = (16 * NT) / L; a.hc = hc;
  a.scale = scale; a.mask = mask & 0xff; a.period = mask >> 8;   // block-causal period rides in the upper bits
  afft_dropout_t dd = {drop_p, drop_key, 0.f, 0u, 1};
  const DropParams dp = make_drop(&dd);
  a.dthresh = dp.thresh; a.dkey = dp.key; a.dinv = dp.inv_keep; a.salt = dp.salt;
  const int groups = (nseq + a.G - 1) / a.G;
  const dim3 grid(groups * H), block(256);
#define AFFT_ATTN_LAUNCH(KERN)                                                                              \
  do {                                                                                                      \
    static std::atomic<uint64_t> attr_done{0};                                                              \
    if (afft_ensure_dynamic_lds(reinterpret_cast<const void*>(KERN), 160 * 1024, &attr_done)) return -1;    \
    hipLaunchKernelGGL(KERN, grid, block, lds, stream, a);                                                  \
  } while (0)
  if (!backward && planes && in_lo) {
    if (NT == 1) AFFT_ATTN_LAUNCH((attn_fwd_mfma_kernel<1, 1>));
    else if (NT == 2) AFFT_ATTN_LAUNCH((attn_fwd_mfma_kernel<2, 1>));
    else AFFT_ATTN_LAUNCH((attn_fwd_mfma_kernel<4, 1>));
  } else if (!backward && planes) {
    if (NT == 1) AFFT_ATTN_LAUNCH((attn_fwd_mfma_kernel<1, 2>));
    else if (NT == 2) AFFT_ATTN_LAUNCH((attn_fwd_mfma_kernel<2, 2>));
    else AFFT_ATTN_LAUNCH((attn_fwd_mfma_kernel<4, 2>));
  } else if (!backward) {
    if (NT == 1) AFFT_ATTN_LAUNCH((attn_fwd_mfma_kernel<1, 0>));
    else if (NT == 2) AFFT_ATTN_LAUNCH((attn_fwd_mfma_kernel<2, 0>));
    else AFFT_ATTN_LAUNCH((attn_fwd_mfma_kernel<4, 0>));
  } else {
    if (NT <= 2 && hd % 128 == 0 && hd <= 512 && lddq % 8 == 0 && lddk % 8 == 0 && lddv % 8 == 0 && al16(dq) && al16(dk) && al16(dv)) {
      // column-sliced backward (attn_bwd_sliced_kernel): 2 wave-private buffers of [16 NT][hd / 4] bf16 per wave + the partial dP tiles
      lds = (size_t)4 * 2 * 16 * NT * (hd / 4) * 2 + (size_t)4 * 2 * NT * NT * 64 * 16;
      if (NT == 1) AFFT_ATTN_LAUNCH(attn_bwd_sliced_kernel<1>);
      else AFFT_ATTN_LAUNCH(attn_bwd_sliced_kernel<2>);
    }
    else if (NT == 1) AFFT_ATTN_LAUNCH(attn_bwd_mfma_kernel<1>);
    else if (NT == 2) AFFT_ATTN_LAUNCH(attn_bwd_mfma_kernel<2>);
    else AFFT_ATTN_LAUNCH(attn_bwd_mfma_kernel<4>);
  }
#undef AFFT_ATTN_LAUNCH
  AFFT_LAUNCH_CHECK();
  return 0;
}
