// bf16 flash-attention FORWARD for gfx950, 32 queries per wave on v_mfma_f32_32x32x16_bf16 (included by attn_bf16.hip).
//
// Replaces the xformers memory_efficient_attention call sites modeling_cogvlm.py:113-128 (BlockDiagonalCausalMask, head_dim 128)
// and visual.py:91-99 (BlockDiagonalMask, head_dim 112).
//
// Structure (guides: cdna_hip_programming.md §B "Fused attention prefill", T10 / T13 / T16; MI355X_MICROARCH.md "Two waves per SIMD"):
//  * one workgroup = 8 waves x 32 queries = one workgroup per CU, two waves per SIMD;
//  * scores are computed transposed, S^T[key][q] = K Q^T, so the query sits on the MFMA column (= lane & 31): a lane's 32 accumulator
//    registers of the two 32-key tiles are 32 keys of ONE query — row max and row sum are in-lane plus one v_permlane32_swap — and
//    registers 8s..8s+7, packed to bf16, ARE the B operand of O^T[d][q] += V^T P^T (§3 "An accumulator tile as the next MFMA's operand");
//    V^T comes from the row-major V tile through ds_read_b64_tr_b16. O^T keeps the query on the lane: the online-softmax rescale is a
//    per-lane scalar;
//  * the head width is walked in 16-wide k-steps of the 32x32x16 instruction: 7 steps for head_dim 112 (the 16x16x32 kernels padded
//    it to 4 x 32); only the P V product pads (4 d-tiles of 32);
//  * four clusters per 64-key tile (guide T16), separated by raw s_barrier:
//      c0  the tile's K fragments LDS -> registers (one burst of ds_read_b128, ONE wait)        | DMA of V(t+2) issued
//      c1  S^T = K Q^T: 14 MFMAs (head_dim 112) back to back from registers                     | DMA of K(t+2) issued
//      c2  the tile's V fragments LDS -> the SAME registers (ds_read_b64_tr_b16), beside the row maximum and the (rare) rescale
//      c3  O^T += V^T P^T: 16 MFMAs from registers, each group of four beside the exponentials, row sums and bf16 packing of the
//          NEXT 16 keys on the VALU
//    and the second half of the workgroup (the SIMD partners of the first half's waves) runs ONE cluster behind: at any moment one
//    wave of a SIMD is in a matrix cluster and the other in an LDS / VALU cluster — with two waves per SIMD an LDS round trip cannot
//    hide behind thread-level parallelism, a wave has to batch its reads and let the partner's MFMAs cover them (the first 32-query
//    form of this kernel waited on an LDS read in front of every second MFMA: 46 % of its wave cycles parked, matrix pipe 0.26 busy);
//  * K and V tiles arrive by LDS-DMA (buffer_load ... lds, the image's XOR on the SOURCE chunk) into three-slot rings, each issued
//    1.5 - 2.5 tiles before it is read and retired by a COUNTED vmcnt (two or three younger groups stay in flight across the barriers);
//  * online softmax with the rescale decision in c2, between one tile's P V product and the next one's, so everything accumulated is
//    at the old maximum exactly once; deferring the rescale (T13) is built in and switched off (RESCALE_THR below: +6 % speed, +5 % error);
//  * epilogue through per-wave LDS slabs: whole 16-byte-chunk rows leave the CU instead of 8-byte pieces of 32 rows per instruction.
#pragma once

namespace a32 {

constexpr int TILE = 64 * ROWB;                 // one 64-row operand tile (16 KiB)
constexpr int SLAB_PITCH = 272;                 // epilogue slab: [32 q][head_dim] bf16 rows, 16-byte aligned, 2-way on the 8-byte writes
constexpr int SLAB = 32 * SLAB_PITCH;
// Deferred rescale (guide T13): with A32_THR = t the running maximum is only raised when a tile exceeds it by more than t (log2 units),
// so most tiles skip the O *= alpha pass. Measured (tools/ubench/attn_bench.hip, -DA32_THR=6.0f against 0): 8 x 785 ViT-E 67.4 vs 71.6 us,
// decoder 46.0 vs 47.9 us — and the error against an fp32 reference grows from 2.21e-3 to 2.33e-3 relative L2 (max abs 2.05e-3 -> 3.44e-3):
// with the TRUE maximum the dominant probability of a row is exactly 1.0 and rounds to bf16 without error, with a stale one it is an
// arbitrary value that does not. 0 (the shipped value) rounds exactly where round 3's kernel and the reference round; the model-level
// parity bars (tests/test_config0_gpu.py) were calibrated on that.
#ifndef A32_THR
#define A32_THR 0.0f
#endif
constexpr float RESCALE_THR = A32_THR;
constexpr int NW = 8;                           // waves per workgroup (measured: 4-wave workgroups, two per CU, tie on 8 x 785 and lose 6 % on 4 x 4609)
constexpr int NS = 3;                           // ring slots per operand: tile t+2 is staged while the late half of the workgroup still reads tile t
constexpr int RING_LDS = (2 * NS * TILE > NW * SLAB) ? 2 * NS * TILE : NW * SLAB;      // K ring | V ring; the epilogue slabs overlay them

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
// tile loop (32-key half, k-step, d-tile) is an instruction immediate. (On the plain 256-byte-row image of vm_tile.hpp the XOR term
// depends on the k-step / d-tile: 15 address registers, which at 250 live registers were spilled INTO the tile loop.)
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
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
// The same read as ld_tr below through inline assembly: hipcc cannot see what a ds_read_b64_tr_b16 BUILTIN reads and puts a vmcnt(0)
// between any LDS-DMA issue and the next such read, i.e. it drains the hand-counted DMA queue once per tile. The assembly form is invisible
// to it; the data is complete only after the cluster's own "s_waitcnt lgkmcnt(0)" (cluster_end), which every consumer sits behind
// (guide §5.7 form (iii): loads, a wait-only statement, sched_barrier(0)).
template <int IMM>
__device__ __forceinline__ bf16x8_t ld_tr_asm(unsigned addrA, unsigned addrB) {
  s16x4_t lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addrA), "i"(IMM));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addrB), "i"(IMM));
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ bf16x8_t ld_tr(const char* baseA, const char* baseB, int imm) {
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(baseA + imm));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(baseB + imm));
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// LDS-DMA of 64-row operand tiles by the 8 waves: a tile is 16 wave-instructions of 1 KiB = one 8-row group x two 64-byte column
// groups (lane -> subtile lane >> 5, row (lane >> 2) & 7, slot lane & 3: lane-linear in LDS, the image's XOR goes onto the SOURCE
// chunk); PW = 2 instructions per wave. The lane's tile row and source column never change; per tile only the rows' physical indices
// do: identity + seq0, or — packed expert-sorted layout — the sequence's slice of row_of_pos, copied into LDS once per workgroup so
// that no global load ever sits in the vmcnt queue between the hand-counted DMA groups.
template <int HD>
struct Stager {
  static constexpr int PW = 16 / NW;
  int row[PW];       // tile row of wave-instruction i
  int col[PW];       // byte offset of the lane's SOURCE chunk inside an operand row, or -1 past the head dimension
  __device__ __forceinline__ void init(int head, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int j = wave * PW + i;
      row[i] = 8 * (j >> 1) + ((lane >> 2) & 7);
      const int chunk = 4 * (2 * (j & 1) + (lane >> 5)) + ((lane & 3) ^ ((row[i] >> 2) & 3));
      col[i] = chunk * 8 < HD ? (head * HD + chunk * 8) * 2 : -1;
    }
  }
  // stage tile `t` (positions 64 t .. 64 t + 63 of the sequence) into `tile`
  __device__ __forceinline__ void stage(__amdgpu_buffer_rsrc_t rs, int ld_b, int t, int seq0, int seqlen, const int* rowtab, char* tile, int wave) const {
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int pos = 64 * t + row[i];
      const bool valid = pos < seqlen && col[i] >= 0;
      const int prow = rowtab ? rowtab[valid ? pos : 0] : seq0 + pos;
      const int voff = valid ? (int)__umul24(prow, ld_b) + col[i] : OOB_OFF;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr_t)(tile + (wave * PW + i) * 1024), 16, voff, 0, 0, 0);
    }
  }
};

// end of a cluster: at most VM VMEM operations stay in flight (the younger DMA groups), the LDS reads of this cluster are done (their
// slot may be restaged once every wave has passed), then the workgroup barrier — fenced for the compiler on both sides: s_barrier
// itself is no memory fence (IntrNoMem) and register-only MFMAs are not ordered by a "memory" clobber (guide rule 18).
template <int VM>
__device__ __forceinline__ void cluster_end() {
  if constexpr (VM == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  else if constexpr (VM == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
  else if constexpr (VM == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
  else if constexpr (VM == -2) asm volatile("" ::: "memory");     // the caller has waited already
  else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // VM == -1: no DMA group to retire here
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

#ifdef A32_STAMPS        // diagnostic build (tools/ubench/attn_bench.hip -DA32_STAMPS): where do a tile's cycles go? (guide §7 "In-kernel stamps")
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define A32_STAMP_WORK(i) { const unsigned long long t_ = stamp(); acc_[2 * (i)] += t_ - last_; last_ = t_; }
#define A32_STAMP_WAIT(i) { const unsigned long long t_ = stamp(); acc_[2 * (i) + 1] += t_ - last_; last_ = t_; }
#define A32_STAMP_SUB(i) { const unsigned long long t_ = stamp(); acc_[8 + (i)] += t_ - last_; }
#else
#define A32_STAMP_WORK(i)
#define A32_STAMP_WAIT(i)
#define A32_STAMP_SUB(i)
#endif

template <int HD, bool CAUSAL>
__global__ __launch_bounds__(NW * 64, 2) void fwd_k(const AttnP p) {
  constexpr int KS = HD / 16;                 // k-steps of the score product
  constexpr int ND = (HD + 31) / 32;          // 32-row d-tiles of O^T
  constexpr int PW = 16 / NW;                 // DMA instructions per wave and tile
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
  // the second half of the workgroup runs ONE CLUSTER behind the first (the two waves of a SIMD are waves w and w + 4): while one of
  // them issues MFMAs from registers the other one reads its next operands from LDS
  const bool late = wave >= NW / 2;

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
  char* const kring = smem;                      // K tile t in slot t % NS
  char* const vring = smem + NS * TILE;          // V tile t in slot t % NS
  int* rowtab = nullptr;
  if (p.row_of_pos) {                            // the sequence's physical rows -> LDS (behind rings and slabs)
    rowtab = reinterpret_cast<int*>(smem + RING_LDS);
    for (int i = tid; i < min(seqlen, nt * 64); i += NW * 64) rowtab[i] = p.row_of_pos[seq0 + i];      // every position a staged tile can hold
    __syncthreads();
  }

  Stager<HD> st;
  st.init(head, wave, lane);
  // DMA schedule (per wave PW = 2 instructions per group, in this order in the vmcnt queue):
  //   prologue K(0) V(0) V(1) K(1);  c0(t) issues V(t+2), c1(t) issues K(t+2)
  //   end of c0(t): V(t) must have landed — the younger groups V(t+1) K(t+1) V(t+2) may stay in flight (vmcnt(3 PW))
  //   end of c2(t): K(t+1) must have landed — V(t+2) K(t+2) may stay in flight (vmcnt(2 PW)); vmcnt(0) once the sequence runs out.
  // Where the issues sit was measured with in-kernel stamps (tools/ubench/attn_bench.hip -DA32_STAMPS): an LDS-DMA piece costs the
  // issuing wave ~130 cycles wherever it is put and does NOT overlap with that wave's own MFMAs, so the four pieces per tile go where
  // the cluster lengths stay balanced (c0 and c1 are the short ones), not "among the MFMAs".
  // Ring slots t % 3: a tile is restaged two tiles after its last read, far behind the half of the workgroup that runs a cluster late.
  st.stage(rK, ldk_b, 0, seq0, seqlen, rowtab, kring, wave);
  st.stage(rV, ldv_b, 0, seq0, seqlen, rowtab, vring, wave);
  if (nt > 1) {
    st.stage(rV, ldv_b, 1, seq0, seqlen, rowtab, vring + TILE, wave);
    st.stage(rK, ldk_b, 1, seq0, seqlen, rowtab, kring + TILE, wave);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                  // K(0) landed on this wave's side (three younger groups)
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  cluster_end<-2>();
  if (late) cluster_end<-1>();                                        // the stagger: one barrier interval behind from here on

  const char* const kb0 = kring + row_base(lane, 0);
  const char* const kb1 = kring + row_base(lane, 1);
  const char* const vbA = vring + tr_base(lane, 0);
  const char* const vbB = vring + tr_base(lane, 1);

#ifdef A32_STAMPS
  unsigned long long acc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_ = stamp();
  const unsigned long long t_begin_ = last_;
#endif
  f32x16_t S[2];
  bf16x8_t kv[16];                               // the K fragments of a tile, then its V fragments (time-shared)
  float msc = NEG_BIG, l_run = 0.f;              // running maximum (scaled log2 domain), this half's running sum
  int s0 = 0, s1 = TILE, s2 = 2 * TILE;          // ring slot offsets of tiles t, t+1, t+2
  for (int t = 0; t < nt; ++t) {
    const bool live = t < nt_w;
    const bool more1 = t + 1 < nt, more2 = t + 2 < nt;
    // ---- c0: K(t) fragments -> registers; the DMA of V(t+2)
    if (live) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kv[kt * KS + ks] = ld_row(((ks & 1) ? kb1 : kb0) + s0, 8192 * kt + 512 * (ks >> 1));
    }
    if (more2) st.stage(rV, ldv_b, t + 2, seq0, seqlen, rowtab, vring + s2, wave);
    A32_STAMP_WORK(0)
    if (more2) cluster_end<3 * PW>(); else if (more1) cluster_end<2 * PW>(); else cluster_end<0>();       // V(t) landed
    A32_STAMP_WAIT(0)
    // ---- c1: S^T = K Q^T, registers only; the DMA of K(t+2)
    if (live) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        f32x16_t acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = mfma32(kv[kt * KS + ks], qf[ks], acc);
        S[kt] = acc;
      }
    }
    if (more2) st.stage(rK, ldk_b, t + 2, seq0, seqlen, rowtab, kring + s2, wave);
    A32_STAMP_WORK(1)
    cluster_end<-1>();
    A32_STAMP_WAIT(1)
    // ---- c2: V(t) fragments -> the same registers, beside the row maximum and the (rare) rescale
    if (live) {
      const unsigned va = (unsigned)(size_t)(vbA + s0), vb2 = (unsigned)(size_t)(vbB + s0);
      static_for<4 * ND>([&](auto i_c) {
        constexpr int I = decltype(i_c)::value;
        kv[I] = ld_tr_asm<4096 * (I / ND) + 512 * (I % ND)>(va, vb2);
      });
      A32_STAMP_SUB(0)
      // only a wave's LAST tile can touch the sequence end or the causal diagonal (earlier tiles end at or below
      // 64 floor((qw + 31) / 64) - 1 <= qw < seqlen)
      if (t + 1 == nt_w) {
        const int kv0 = t * 64;
        if (kv0 + 64 > seqlen || (CAUSAL && kv0 + 63 > qw)) {
          const int lim = CAUSAL ? min(qpos, seqlen - 1) : seqlen - 1;
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) S[kt][r] = (kv0 + 32 * kt + acc_row(r, h) <= lim) ? S[kt][r] : NEG_BIG;
        }
      }
      // v_max3_f32 through assembly: fmaxf() on MFMA results is compiled as IEEE maxNum — every operand first canonicalised by a
      // v_max_f32 x, x — 48 instructions instead of 16 (the scores are never NaN unless the inputs are, and then the output is NaN
      // either way). Four independent chains (one chain of 16 dependent instructions is latency-bound).
      float m4[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m4[c]) : "v"(S[0][4 * c]), "v"(S[1][4 * c]), "v"(S[0][4 * c + 1]));
        asm("v_max3_f32 %0, %0, %1, %2" : "+v"(m4[c]) : "v"(S[1][4 * c + 1]), "v"(S[0][4 * c + 2]));
        asm("v_max3_f32 %0, %0, %1, %2" : "+v"(m4[c]) : "v"(S[1][4 * c + 2]), "v"(S[0][4 * c + 3]));
      }
      float mx;
      asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(m4[0]), "v"(m4[1]), "v"(S[1][3]));
      asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(m4[2]), "v"(S[1][7]));
      asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(m4[3]), "v"(S[1][11]));
      asm("v_max_f32 %0, %0, %1" : "+v"(mx) : "v"(S[1][15]));
      float ma = mx, mb = mx;
      swap_halves(ma, mb);                       // ma = the lower half's maximum, mb = the upper half's, on every lane
      float mab;
      asm("v_max_f32 %0, %1, %2" : "=v"(mab) : "v"(ma), "v"(mb));
      const float mxs = mab * sc;
      if (__builtin_amdgcn_ballot_w64(mxs > msc + RESCALE_THR) != 0) {      // wave-uniform, rare after the first tile (T13)
        const float nm = fmaxf(msc, mxs);
        const float alpha = fast_exp2(msc - nm);
        msc = nm;
        l_run *= alpha;
#pragma unroll
        for (int b = 0; b < ND; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
      }
      A32_STAMP_SUB(1)
    }
    A32_STAMP_WORK(2)
    if (more2) cluster_end<2 * PW>(); else cluster_end<0>();          // K(t+1) landed
    A32_STAMP_WAIT(2)
    // ---- c3: O^T += V^T P^T from registers, beside the exponentials, row sums and bf16 packing of the NEXT 16 keys (VALU)
    if (live) {
      float rs0 = 0.f, rs1 = 0.f;
      const f32x2_t sc2 = {sc, sc}, nm2 = {-msc, -msc};
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        u16x8_t w;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const int r = 8 * (m & 1) + j;
          const f32x2_t x = __builtin_elementwise_fma((f32x2_t){S[m >> 1][r], S[m >> 1][r + 1]}, sc2, nm2);
          float e0 = fast_exp2(x[0]), e1 = fast_exp2(x[1]);
          w[j] = f2bf(e0);
          w[j + 1] = f2bf(e1);
          // plain single adds (assembly: left to itself hipcc packs them into v_pk_add_f32 fed by ~45 v_mov per tile)
          asm("v_add_f32 %0, %1, %0" : "+v"(rs0) : "v"(e0));
          asm("v_add_f32 %0, %1, %0" : "+v"(rs1) : "v"(e1));
        }
        const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, w);
#pragma unroll
        for (int b = 0; b < ND; ++b) o[b] = mfma32(kv[m * ND + b], pf, o[b]);
      }
      l_run += rs0 + rs1;
    }
    A32_STAMP_WORK(3)
    cluster_end<-1>();
    A32_STAMP_WAIT(3)
    { const int x = s0; s0 = s1; s1 = s2; s2 = x; }
  }
#ifdef A32_STAMPS
  if (p.dbg && lane == 0) {
    unsigned long long* d = p.dbg + ((size_t)blockIdx.x * NW + wave) * 12;
    for (int i = 0; i < 8; ++i) d[i] = acc_[i];
    d[8] = stamp() - t_begin_;
    d[9] = (unsigned long long)nt_w;
    d[10] = acc_[8]; d[11] = acc_[9];
  }
#endif
  if (!late) cluster_end<-1>();                  // the first half waits for the second: every LDS read is over, the slabs may overlay the rings

  if (!wave_live) return;
  {
    float a = l_run, b = l_run;
    swap_halves(a, b);
    l_run = a + b;
  }
  const float inv_l = l_run > 0.f ? 1.0f / l_run : 0.f;
  if (qvalid && h == 0 && p.lse) p.lse[(int64_t)head * p.total_pos_max + seq0 + qpos] = (msc + log2f(l_run)) * LN2;
  // O^T registers -> this wave's slab [32 q][HD], then whole rows out
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
