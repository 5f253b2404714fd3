// bf16 flash-attention FORWARD for gfx950, 32 queries per wave on v_mfma_f32_32x32x16_bf16 (included by attn_bf16.hip).
//
// Replaces the xformers memory_efficient_attention call sites modeling_cogvlm.py:113-128 (BlockDiagonalCausalMask, head_dim 128)
// and visual.py:91-99 (BlockDiagonalMask, head_dim 112).
//
// Structure (guides: cdna_hip_programming.md §B "Fused attention prefill", T10 / T12 / T13 / T15; MI355X_MICROARCH.md "Two waves per SIMD"):
//  * one workgroup = NW waves (8: one workgroup per CU, two waves per SIMD; 4: two workgroups per CU) x 32 queries; the queries of a
//    sequence are split EVENLY over its workgroups in multiples of 32 (785 ViT-E tokens = 4 x 224, not 3 x 256 + 17);
//  * scores are computed transposed, S^T[key][q] = K Q^T, so the query sits on the MFMA column (= lane & 31): a lane's 32 accumulator
//    registers of the two 32-key tiles are 32 keys of ONE query — row max and row sum are in-lane plus one v_permlane32_swap — and
//    registers 8s..8s+7, packed to bf16, ARE the B operand of O^T[d][q] += V^T P^T (§3 "An accumulator tile as the next MFMA's operand");
//    V^T comes from the row-major V tile through ds_read_b64_tr_b16. O^T keeps the query on the lane: the online-softmax rescale is a
//    per-lane scalar;
//  * the head width is walked in 16-wide k-steps of the 32x32x16 instruction: 7 steps for head_dim 112 (the 16x16x32 kernels padded
//    it to 4 x 32); only the P V product pads (4 d-tiles of 32);
//  * software pipeline inside a wave (T15), two phases per 64-key tile t:
//      X: O^T += V(t-1)^T P(t-1)^T  (16 MFMAs, operands: the PREVIOUS tile's packed probabilities and V tile)  beside
//         P(t) = exp2(S(t) sc - m), row sums, bf16 packing (112 VALU): independent instruction streams of equal length, so ONE wave
//         keeps the matrix pipe and the VALU busy at the same time;
//      Y: S(t+1) = K(t+1) Q^T (14 MFMAs at head_dim 112) into the registers S(t) just left, then its row maximum.
//    K is consumed one tile ahead of the exponentials, V one tile behind: K ring of 2 slots, V ring of 3. Tiles arrive by LDS-DMA
//    (buffer_load ... lds, the image's XOR on the SOURCE chunk), issued a whole iteration before they are read, so the one
//    vmcnt(0) + barrier per tile finds them landed;
//  * the O rescale is deferred while the running maximum grows by less than 2^RESCALE_THR (T13): exponentials then exceed 1 by at
//    most that factor — bf16's relative precision does not depend on the magnitude, the row sum and O are fp32. When it does fire,
//    the pending product P(t) V(t) is flushed FIRST (and P(t) zeroed), so everything accumulated is at the old maximum exactly once
//    (T13's hazard: never rescale between a tile's exponentials and its P V);
//  * epilogue through per-wave LDS slabs: whole 16-byte-chunk rows leave the CU instead of 8-byte pieces of 32 rows per instruction.
#pragma once

namespace a32 {

constexpr int TILE = 64 * ROWB;                 // one 64-row operand tile (16 KiB)
constexpr int SLAB_PITCH = 272;                 // epilogue slab: [32 q][head_dim] bf16 rows, 16-byte aligned, 2-way on the 8-byte writes
constexpr int SLAB = 32 * SLAB_PITCH;
constexpr float RESCALE_THR = 6.0f;             // log2 units: P <= 64

template <int NW> constexpr int fwd_lds() { return (5 * TILE > NW * SLAB) ? 5 * TILE : NW * SLAB; }      // K[2] | V[3]; the epilogue slabs reuse it

// every wave's LDS-DMA has landed (its own vmcnt(0)) and every wave has finished reading the slots about to be restaged.
// s_barrier itself is no memory fence for the compiler (IntrNoMem): without the empty asm behind it LDS reads of the next step
// are hoisted above the barrier and see slots other waves' DMA has not filled yet.
__device__ __forceinline__ void dma_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// lanes 32..63 of `a` <-> lanes 0..31 of `b`. Inline assembly on purpose: given the same VALUE for both operands, hipcc 7.2 folds the
// builtin's two results into ONE register (max(r[0], r[1]) became r[0]: every half kept only the OTHER half's maximum — 5 % errors).
// The s_nop covers the VALU-write -> permlane-read wait states nobody inserts for inline code.
__device__ __forceinline__ void swap_halves(float& a, float& b) {
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

__device__ __forceinline__ f32x16_t mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// LDS image of a 64-row x 256-byte operand tile (guide T10, image (a)): 8-row x 64-byte subtiles of 512 B,
//   off(row, ch) = 2048 (row >> 3) + 512 (ch >> 2) + 64 (row & 7) + 16 ((ch & 3) ^ ((row >> 2) & 3))        ch = 16-byte chunk of the row
// Both kinds of read are conflict-free on it AND need only two lane-constant base addresses each — everything that varies inside the
// tile loop (ring slot, 32-key half, k-step, d-tile) is an instruction immediate. (On the plain 256-byte-row image of vm_tile.hpp the
// XOR term depends on the k-step / d-tile: 15 address registers, which at 250 live registers were spilled INTO the tile loop.)
//  * row fragment of the 32x32x16 A operand, lane (r = lane & 31, h = lane >> 5), rows 32 kt + r, chunk 2 ks + h:
//      base[ks & 1] + 8192 kt + 512 (ks >> 1),   base[e] = 2048 (r >> 3) + 64 (r & 7) + 16 ((2 e + h) ^ ((r >> 2) & 3))
//  * transposed fragment (ds_read_b64_tr_b16 x 2), lane (i = lane & 15, g = (lane >> 4) & 1, h), rows 16 m + 4 h + (i >> 2) (+ 8), d-tile b:
//      baseA + 4096 m + 512 b,  baseB + 4096 m + 512 b   (baseB = the +8 rows: + 2048 and the XOR term of (h + 2) & 3)
__device__ __forceinline__ int row_base(int lane, int e) {
  const int r = lane & 31, h = lane >> 5;
  return 2048 * (r >> 3) + 64 * (r & 7) + 16 * ((2 * e + h) ^ ((r >> 2) & 3));
}
__device__ __forceinline__ int tr_base(int lane, int second) {
  const int i = lane & 15, g = (lane >> 4) & 1, h = lane >> 5;
  const int q4 = i >> 2, pp = i & 3;
  return 2048 * second + 64 * (4 * h + q4) + 16 * ((2 * g + (pp >> 1)) ^ ((h + 2 * second) & 3)) + 8 * (pp & 1);
}
__device__ __forceinline__ bf16x8_t ld_row(const char* base, int imm) {
  return *reinterpret_cast<const bf16x8_t*>(base + imm);
}
__device__ __forceinline__ bf16x8_t ld_tr(const char* baseA, const char* baseB, int imm) {
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(baseA + imm));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(baseB + imm));
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// LDS-DMA of 64-row operand tiles by NW waves: a tile is 16 wave-instructions of 1 KiB = one 8-row group x two 64-byte column groups
// (lane -> subtile lane >> 5, row (lane >> 2) & 7, slot lane & 3: lane-linear in LDS, the image's XOR goes onto the SOURCE chunk);
// PW = 16 / NW instructions per wave. The lane's tile row and source column never change; per tile only the rows' physical indices do.
template <int HD, int NW>
struct Stager {
  static constexpr int PW = 16 / NW;
  int row[PW];       // tile row of wave-instruction i
  int col[PW];       // byte offset of the lane's SOURCE chunk inside an operand row, or -1 past the head dimension
  int pr[PW];        // physical rows of the next tile to stage, fetched a whole step before they are used (the packed layout's
  int pr2[PW];       // indirection is a global load: used right away it would drain the DMA queue in front of it) and of the one after
  __device__ __forceinline__ void init(int head, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int j = wave * PW + i;
      row[i] = 8 * (j >> 1) + ((lane >> 2) & 7);
      const int chunk = 4 * (2 * (j & 1) + (lane >> 5)) + ((lane & 3) ^ ((row[i] >> 2) & 3));
      col[i] = chunk * 8 < HD ? (head * HD + chunk * 8) * 2 : -1;
    }
  }
  __device__ __forceinline__ void rows(const AttnP& p, int seq0, int seqlen, int pos0, int (&dst)[PW]) const {
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int pos = pos0 + row[i];
      dst[i] = pos < seqlen ? phys_row(p, seq0 + pos) : 0;
    }
  }
  // stage a tile of `rows_left` valid rows whose physical rows are `src`
  __device__ __forceinline__ void stage(__amdgpu_buffer_rsrc_t rs, int ld_b, int rows_left, const int (&src)[PW], char* tile, int wave) const {
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const bool valid = row[i] < rows_left && col[i] >= 0;
      const int voff = valid ? (int)__umul24(src[i], ld_b) + col[i] : OOB_OFF;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr_t)(tile + (wave * PW + i) * 1024), 16, voff, 0, 0, 0);
    }
  }
  __device__ __forceinline__ void shift() {
#pragma unroll
    for (int i = 0; i < PW; ++i) pr[i] = pr2[i];
  }
};

// S^T tiles of 64 keys x this wave's 32 queries: s[kt][r] = score(key = 32 kt + acc_row(r, lane >> 5), q = lane & 31).
// kb0 / kb1: the lane's row-fragment bases for even / odd k-steps inside the K ring; `imm` = ring slot offset (compile-time)
template <int KS>
__device__ __forceinline__ void scores(const char* kb0, const char* kb1, int imm, const bf16x8_t (&qf)[KS], f32x16_t (&s)[2]) {
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    f32x16_t acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = mfma32(ld_row((ks & 1) ? kb1 : kb0, imm + 8192 * kt + 512 * (ks >> 1)), qf[ks], acc);
    s[kt] = acc;
  }
}

template <int HD, int NW, bool CAUSAL>
__global__ __launch_bounds__(NW * 64, 2) void fwd_k(const AttnP p) {
  constexpr int KS = HD / 16;                 // k-steps of the score product
  constexpr int ND = (HD + 31) / 32;          // 32-row d-tiles of O^T
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5;
  int tile_, head, seq;
  if (!a16_block(p, tile_, head, seq)) return;
  const int seq0 = p.cu[seq];
  const int seqlen = p.cu[seq + 1] - seq0;
  const int q0 = tile_ * p.q_block;
  if (q0 >= seqlen) return;
  const int qw = q0 + wave * 32;                                   // this wave's first query
  const bool wave_live = wave * 32 < p.q_block && qw < seqlen;     // (dead waves still stage tiles and meet the barriers)
  const int qpos = qw + (lane & 31);
  const bool qvalid = wave_live && qpos < seqlen;
  const int64_t qrow = qvalid ? phys_row(p, seq0 + qpos) : 0;

  bf16x8_t qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    i32x4_t v = {0, 0, 0, 0};
    if (qvalid) v = *reinterpret_cast<const i32x4_t*>(p.q + qrow * p.ldq + head * HD + 16 * s + 8 * h);
    qf[s] = __builtin_bit_cast(bf16x8_t, v);
  }
  f32x16_t o[ND];
#pragma unroll
  for (int b = 0; b < ND; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[b][r] = 0.f;
  const float sc = p.scale * LOG2E;

  const int kv_end = CAUSAL ? min(seqlen, q0 + p.q_block) : seqlen;
  const int nt = (kv_end + 63) / 64;
  // tiles this wave computes: causal waves stop at their own diagonal
  const int nt_w = !wave_live ? 0 : (CAUSAL ? min(nt, (qw + 31) / 64 + 1) : nt);
  const __amdgpu_buffer_rsrc_t rK = whole_rsrc(p.k), rV = whole_rsrc(p.v);
  const int ldk_b = (int)p.ldk * 2, ldv_b = (int)p.ldv * 2;
  char* const kring = smem;                      // K tile t in slot t & 1
  char* const vring = smem + 2 * TILE;           // V tile t in slot t % 3

  Stager<HD, NW> st;
  st.init(head, wave, lane);
  st.rows(p, seq0, seqlen, 0, st.pr);
  st.rows(p, seq0, seqlen, 64, st.pr2);
  st.stage(rK, ldk_b, seqlen, st.pr, kring, wave);
  st.stage(rV, ldv_b, seqlen, st.pr, vring, wave);
  if (nt > 1) st.stage(rK, ldk_b, seqlen - 64, st.pr2, kring + TILE, wave);
  // pr = rows of tile 1, pr2 = rows of tile 2 from here on (iteration t stages V(t+1) from pr and K(t+2) from pr2)
  st.shift();
  st.rows(p, seq0, seqlen, 128, st.pr2);
  dma_barrier();

  const char* const kb0 = kring + row_base(lane, 0);
  const char* const kb1 = kring + row_base(lane, 1);
  const char* const vbA = vring + tr_base(lane, 0);
  const char* const vbB = vring + tr_base(lane, 1);

  // the only tile of a wave that can touch the sequence end or the causal diagonal is its LAST one (earlier tiles end at or below
  // 64 floor((qw + 31) / 64) - 1 <= qw < seqlen)
  auto mask_last = [&](f32x16_t (&s)[2], int t) {
    const int kv0 = t * 64;
    if (kv0 + 64 > seqlen || (CAUSAL && kv0 + 63 > qw)) {
      const int lim = CAUSAL ? min(qpos, seqlen - 1) : seqlen - 1;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = (kv0 + 32 * kt + acc_row(r, h) <= lim) ? s[kt][r] : NEG_BIG;
    }
  };
  auto row_max_scaled = [&](const f32x16_t (&s)[2]) {
    float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s[0][r], s[1][r]));
    float a = mx, b = mx;
    swap_halves(a, b);                           // a = the lower half's maximum, b = the upper half's, on every lane
    return fmaxf(a, b) * sc;
  };
  // P = exp2(S sc - msc) -> four packed B-operand fragments; returns this half's row sum
  auto exp_pack = [&](const f32x16_t (&s)[2], float msc, bf16x8_t (&pf)[4]) {
    float rs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        u16x8_t w;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float e = fast_exp2(__builtin_fmaf(s[kt][8 * s2 + j], sc, -msc));
          rs[j & 3] += e;
          w[j] = f2bf(e);
        }
        pf[2 * kt + s2] = __builtin_bit_cast(bf16x8_t, w);
      }
    return (rs[0] + rs[1]) + (rs[2] + rs[3]);
  };
  // O^T[d][q] += V^T P^T from the V tile in ring slot `voff` (byte offset)
  auto pv = [&](int voff, const bf16x8_t (&pf)[4]) {
    const char* a = vbA + voff;
    const char* b2 = vbB + voff;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int b = 0; b < ND; ++b) o[b] = mfma32(ld_tr(a, b2, 4096 * m + 512 * b), pf[m], o[b]);
  };

  f32x16_t S[2];
  bf16x8_t pprev[4];
  float msc = NEG_BIG, l_run = 0.f;              // running maximum (scaled log2 domain), this half's running sum
  if (nt_w > 0) {
    scores<KS>(kb0, kb1, 0, qf, S);
    if (nt_w == 1) mask_last(S, 0);
    msc = row_max_scaled(S);                     // (O and l are still zero: nothing to rescale)
  }
  int kslot = TILE;                              // byte offset of K(t+1)'s slot inside the K ring
  int vprev = 0, vcur = 0, vnext = TILE;         // byte offsets of the V slots of tiles t-1, t, t+1
  for (int t = 0; t < nt; ++t) {
    // ---- stage ahead: V(t+1) into the slot V(t-2) left, K(t+2) into the slot K(t) left in the previous iteration
    if (t + 1 < nt) {
      st.stage(rV, ldv_b, seqlen - (t + 1) * 64, st.pr, vring + vnext, wave);
      if (t + 2 < nt) st.stage(rK, ldk_b, seqlen - (t + 2) * 64, st.pr2, kring + (kslot ^ TILE), wave);
      st.shift();
      if (t + 3 < nt) st.rows(p, seq0, seqlen, (t + 3) * 64, st.pr2);
    }
    if (t < nt_w) {
      // ---- X: exponentials of tile t (VALU) beside the P V product of tile t-1 (MFMA)
      bf16x8_t pcur[4];
      if (t > 0) {
        pv(vprev, pprev);
        l_run += exp_pack(S, msc, pcur);
      } else {
        l_run += exp_pack(S, msc, pcur);
      }
      // ---- Y: scores of tile t+1 into the registers S(t) just left, its row maximum, and the (rare) rescale
      if (t + 1 < nt_w) {
        scores<KS>(kb0 + kslot, kb1 + kslot, 0, qf, S);
        if (t + 2 == nt_w) mask_last(S, t + 1);
        const float mxs = row_max_scaled(S);
        if (__builtin_amdgcn_ballot_w64(mxs > msc + RESCALE_THR) != 0) {      // wave-uniform, rare
          pv(vcur, pcur);                        // flush the pending product: everything accumulated is at the old maximum
#pragma unroll
          for (int m = 0; m < 4; ++m) pcur[m] = __builtin_bit_cast(bf16x8_t, (i32x4_t){0, 0, 0, 0});
          const float nm = fmaxf(msc, mxs);
          const float alpha = fast_exp2(msc - nm);
          msc = nm;
          l_run *= alpha;
#pragma unroll
          for (int b = 0; b < ND; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) pprev[m] = pcur[m];
    } else if (t == nt_w && nt_w > 0) {
      pv(vprev, pprev);                          // a causal wave past its diagonal: its last product, then it only stages
    }
    kslot ^= TILE;
    vprev = vcur; vcur = vnext; vnext = vnext == 2 * TILE ? 0 : vnext + TILE;
    dma_barrier();
  }
  if (nt_w == nt && nt_w > 0) pv(vprev, pprev);

  if (!wave_live) return;
  {
    float a = l_run, b = l_run;
    swap_halves(a, b);
    l_run = a + b;
  }
  const float inv_l = l_run > 0.f ? 1.0f / l_run : 0.f;
  if (qvalid && h == 0 && p.lse) p.lse[(int64_t)head * p.total_pos_max + seq0 + qpos] = (msc + log2f(l_run)) * LN2;
  // O^T registers -> this wave's slab [32 q][HD] (every wave passed the last barrier: the rings are free), then whole rows out
  __builtin_amdgcn_s_barrier();                  // (the last product above still read the V ring)
  asm volatile("" ::: "memory");
  char* slab = smem + wave * SLAB;
#pragma unroll
  for (int b = 0; b < ND; ++b)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int d0 = 32 * b + 8 * g4 + 4 * h;
      if (d0 < HD) {
        const u16x4_t w = {f2bf(o[b][4 * g4] * inv_l), f2bf(o[b][4 * g4 + 1] * inv_l), f2bf(o[b][4 * g4 + 2] * inv_l), f2bf(o[b][4 * g4 + 3] * inv_l)};
        *reinterpret_cast<u16x4_t*>(slab + (lane & 31) * SLAB_PITCH + d0 * 2) = w;
      }
    }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  constexpr int CPR = HD / 8;                    // 16-byte chunks per row
#pragma unroll
  for (int i = lane; i < 32 * CPR; i += 64) {
    const int r = i / CPR, c = i - r * CPR;
    const int qp = qw + r;
    if (qp < seqlen) {
      const int64_t row = phys_row(p, seq0 + qp);
      *reinterpret_cast<i32x4_t*>(p.out + row * p.ldo + head * HD + c * 8) = *reinterpret_cast<const i32x4_t*>(slab + r * SLAB_PITCH + c * 16);
    }
  }
}

}  // namespace a32
