// Four-wave form of the 256-column bf16 NT GEMM (included by gemm256.hip inside its anonymous namespace, and by tools/ubench/w4_core_bench.hip).
#pragma once
#include "vm_common.hpp"
#include "gemm_common.hpp"
#include <type_traits>

// -DVM_W4_EXPERIMENT=<bits> (tools/ubench/w4_core_bench.hip only; the library builds with 0): knock-outs for timing — 1 no LDS-DMA inside the K loop,
// 2 no fragment reads inside the K loop, 4 no barrier / waits inside the K loop, 8 no epilogue at all, 16 no slab flush, 32 no slab fill (results wrong by construction);
// scheduling variants that keep the results: 256 no scheduling fences between the MFMA rows
#ifndef VM_W4_EXPERIMENT
#define VM_W4_EXPERIMENT 0
#endif
constexpr int W4X = VM_W4_EXPERIMENT;
constexpr int W4X_SCHED_MASK = 0xFF00;

// One MFMA of the four-wave body. The accumulator operand is constrained to the ACCUMULATOR half of the register file ("+a"): with the
// builtin, hipcc's allocator — 256 accumulator registers live across the whole K loop, no spare one — shuffled accumulator tiles between
// the two halves inside the loop (per K-tile and wave 70-200 v_accvgpr_read / _write / _mov and 10-30 s_nop next to the 128 MFMAs: the
// MFMA-only knock-out of the loop ran at 1 600 TFLOP/s where the bare instruction stream sustains 2 100). As inline assembly the tiles
// never move. hipcc does not know what the statement is (guide section 5.7): the operands' waits are still the compiler's (they are
// ordinary "v" inputs), the wait states between the LAST MFMA and the first non-MFMA reader of an accumulator are ours (w4_mfma_drain).
__device__ __forceinline__ void w4_mfma(f32x4_t& acc, const bf16x8_t& b, const bf16x8_t& a) {
  if constexpr (W4X & 512) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc, 0, 0, 0);
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(b), "v"(a));
}
__device__ __forceinline__ void w4_mfma_drain() {
  if constexpr (!(W4X & 512)) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
}

// ====================================================================================================================
// Four-wave form (round 6): 256 threads, ONE wave per SIMD, wave tile (16 TA) x 128 — 128 x 128 for the 256-row tile — so that
// a K-tile costs the CU 32 fragment reads per 128 MFMAs and wave (0.25 per MFMA; the eight-wave form above: 24 per 64 = 0.375) and
// the LDS array carries 128 KiB instead of 192 KiB of fragment traffic per K-tile. This is the geometry the vendor library's
// fastest kernels use on this chip (MT256x256x64, 4 waves, wave tile 8 x 8 MFMA tiles; it is 5-10 % ahead of the eight-wave
// kernel on the step's shapes: profiles/r6_gemm_vs_library.txt). A single wave per SIMD has no partner to hide behind: the
// K loop is software-pipelined INSIDE the wave —
//   * a K-tile is two 32-deep substeps; the 64 (TA = 8) MFMAs of a substep run from one fragment set (8 A + 8 B fragments, 64
//     VGPRs) while the 16 ds_read_b128 of the NEXT substep fill the other set, two reads behind every eight MFMAs;
//   * the accumulators (64 tiles x 4 = 256 registers) live in the accumulator half of the unified register file;
//   * LDS: two stages of (A 256 x 128 B | B 256 x 128 B), row-major with the 16-byte chunk XOR (row & 7) applied to the SOURCE
//     address of the LDS-DMA (guide rule 21) — the image of gemm_nt_k;
//   * ONE barrier per K-tile, between its substeps: in front of it every wave has received its second-substep fragments (the last
//     reads of the current stage) and has waited for its own LDS-DMA pieces of tile t + 1 (issued a whole substep earlier, behind
//     the MFMAs of tile t - 1's second substep); behind it the reads of tile t + 1 and the DMA of tile t + 2 (into the stage just
//     vacated) are legal. 14-16 DMA pieces per wave and K-tile, two behind every eight MFMAs of a second substep.
// Same K order per output element as the eight-wave form (extension tiles, then the main tiles, k ascending): bit-identical results.
// Flush of a full ROWS x 128 bf16 slab (interior tile, 16-byte aligned rows of C and of the residual): no per-row predicates, every LDS read
// and every residual load of a batch of 8 rows-of-4 issued before the first store — one wave alone on its SIMD has nobody to hide a
// read -> wait -> store chain behind (epi_flush's predicated form cost this kernel 17 us per tile).
template <int ROWS>
__device__ __forceinline__ void w4_flush_full(const char* slab, const GemmParams& p, int64_t m0, int n0, int lane) {
  constexpr int PITCH = EpiSlab<ROWS, 128>::PITCH;
  const int ch = lane & 15, rr = lane >> 4;                 // 16 chunks of 16 B per row, 4 rows per wave-instruction
  unsigned short* cbase = (unsigned short*)p.C + (m0 + rr) * p.ldc + n0 + ch * 8;
  const char* sbase = slab + rr * PITCH + ch * 16;
  const unsigned short* rp = (const unsigned short*)p.residual;
  constexpr int NIT = ROWS / 4, BATCH = 8;
#pragma unroll
  for (int b0 = 0; b0 < NIT; b0 += BATCH) {
    u16x8_t v[BATCH], rv[BATCH];
#pragma unroll
    for (int k = 0; k < BATCH; ++k)
      if (b0 + k < NIT) v[k] = *reinterpret_cast<const u16x8_t*>(sbase + (b0 + k) * 4 * PITCH);
    if (rp) {
      const unsigned short* rbase = rp + (m0 + rr) * p.ldr + n0 + ch * 8;
#pragma unroll
      for (int k = 0; k < BATCH; ++k)
        if (b0 + k < NIT) rv[k] = *reinterpret_cast<const u16x8_t*>(rbase + (int64_t)(b0 + k) * 4 * p.ldr);
#pragma unroll
      for (int k = 0; k < BATCH; ++k)
        if (b0 + k < NIT) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[k][e] = f2bf(bf2f(v[k][e]) + bf2f(rv[k][e]));
        }
    }
#pragma unroll
    for (int k = 0; k < BATCH; ++k)
      if (b0 + k < NIT) *reinterpret_cast<u16x8_t*>(cbase + (int64_t)(b0 + k) * 4 * p.ldc) = v[k];
  }
}

template <bool OUT_F32, int TA>
__device__ __forceinline__ void gemm256w_tile(const GemmParams& p, char* smem, const int tid, int tm, int tn) {
  constexpr int BMT = 32 * TA;           // tile rows: 256 (TA = 8) or 192 (TA = 6)
  constexpr int WR = 16 * TA;            // rows of one wave row
  constexpr int OPB = 256 * 128;         // LDS bytes of one operand tile (A occupies its first BMT rows)
  constexpr int STG = 2 * OPB;
  constexpr int NP = TA + 8;             // LDS-DMA pieces per wave and K-tile
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  int row0, nrows, seg;
  gemm_tile_rows<BMT>(p, tm, row0, nrows, seg);
  if (nrows <= 0) return;
  const int n0 = tn * 256;
  const int ncols = min(256, p.N - n0);
  const char* Bw = seg ? p.B1 : p.B0;
  const char* B2w = seg ? p.B2_1 : p.B2_0;
  const int lda_b = (int)p.lda * 2, ldb_b = (int)p.ldb * 2;
  const __amdgpu_buffer_rsrc_t rA = make_rsrc(p.A, (int64_t)row0 * lda_b, nrows * lda_b);
  const __amdgpu_buffer_rsrc_t rB = make_rsrc(Bw, (int64_t)n0 * ldb_b, ncols * ldb_b);
  const int kt_ext = p.K2 / 64, kt_total = kt_ext + p.K / 64;
  __amdgpu_buffer_rsrc_t rA2 = rA, rB2 = rB;
  int lda2_b = 0, ldb2_b = 0;
  if (kt_ext > 0) {
    lda2_b = (int)p.lda2 * 2; ldb2_b = (int)p.ldb2 * 2;
    rA2 = make_rsrc(p.A2, (int64_t)row0 * lda2_b, nrows * lda2_b);
    rB2 = make_rsrc(B2w, (int64_t)n0 * ldb2_b, ncols * ldb2_b);
  }
  // piece q of this wave: rows 8 (wave + 4 q) .. + 7 of the operand tile, 128 B each; lane -> (row r8, LDS slot): source chunk slot ^ r8
  const int r8 = lane >> 3, c16 = ((lane & 7) ^ r8) * 16;
  const int prow = wave * 8 + r8;
  i32x4_t sink[(W4X & 1024) ? NP : 1];
  // per-lane byte offsets of this wave's pieces inside the MAIN operands (loop invariants: the K-tile travels in the scalar offset)
  int voA[TA], voB[8];
#pragma unroll
  for (int q = 0; q < TA; ++q) voA[q] = (prow + 32 * q) * lda_b + c16;
#pragma unroll
  for (int q = 0; q < 8; ++q) voB[q] = (prow + 32 * q) * ldb_b + c16;
  // idx < TA: A piece idx, else B piece idx - TA (compile-time after unrolling). MAIN: tile t is known to be a main tile (no selects)
  auto piece = [&](int t, int idx, auto main_tag) {
    constexpr bool MAIN = decltype(main_tag)::value;
    const bool ext = !MAIN && t < kt_ext;
    const int koff = (ext ? t : t - kt_ext) * 128;
    const bool isb = idx >= TA;
    const int q = isb ? idx - TA : idx;
    char* dst = smem + (t & 1) * STG + (isb ? OPB : 0) + (wave + 4 * q) * 1024;
    const int vo = MAIN ? (isb ? voB[q] : voA[q]) : (prow + 32 * q) * (isb ? (ext ? ldb2_b : ldb_b) : (ext ? lda2_b : lda_b)) + c16;
    const __amdgpu_buffer_rsrc_t rs = MAIN ? (isb ? rB : rA) : (isb ? (ext ? rB2 : rB) : (ext ? rA2 : rA));
    if constexpr (W4X & 1024) {      // timing experiment: the same bytes as a plain load into registers (nothing reaches the LDS: results wrong)
      sink[idx] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, koff, 0);
      (void)dst;
    } else
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, vo, koff, 0, 0);
  };

  f32x4_t acc[TA][8];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t fa[2][TA], fb[2][8];      // fragment sets: [set][tile]; set 0 = first substep (k 0..31), set 1 = second

  const int frow = lane & 15, fq = lane >> 4;
  const int slot0 = fq ^ (frow & 7);
  const int offk[2] = {frow * 128 + slot0 * 16, frow * 128 + (slot0 ^ 4) * 16};
  const int a_base = wm * WR * 128, b_base = OPB + wn * 128 * 128;
  // fragment idx of substep ks out of stage `st` into set `set`: idx < 8 a B tile, else an A tile — in the order the next substep consumes
  // them (its first row of MFMAs takes all eight B fragments and A tile 0: read last, they would be waited for at every substep start)
  auto fread = [&](const char* st, int ks, int set, int idx) {
    if (idx < 8) fb[set][idx] = *reinterpret_cast<const bf16x8_t*>(st + b_base + idx * 2048 + offk[ks]);
    else fa[set][idx - 8] = *reinterpret_cast<const bf16x8_t*>(st + a_base + (idx - 8) * 2048 + offk[ks]);
  };
  constexpr int NF = TA + 8;               // fragments per substep
  constexpr int NS = 8 * TA;               // MFMA slots per substep (one MFMA each)
  // A single wave on its SIMD issues in order: an MFMA keeps the matrix pipe busy for 16 cycles and the issue port for 8 of them, every
  // other instruction of the wave costs >= 4 issue cycles — so the companions (fragment reads, DMA pieces with their M0 write) are dealt
  // out ONE PER MFMA SLOT and the order is pinned slot by slot. Clustered behind a row of eight MFMAs (2 reads + 2 pieces + their scalar
  // set-up, ~40 issue cycles) they left the matrix pipe idle for ~30 cycles per row, a quarter of the loop.
  // Slot k of a substep = MFMA (i = k / 8, j = k % 8). Reads go to slots 4 r + 1 (r-th fragment), pieces to slots 4 q + 3.
  constexpr int RSTEP = NS / NF >= 4 ? 4 : NS / NF, PSTEP = NS / NP >= 4 ? 4 : NS / NP;

  // one K-tile. H1: tile t + 1 exists (its first-substep fragments are read behind the second substep); H2: tile t + 2 exists (its DMA
  // is issued behind the second substep, into the stage this tile leaves); MAIN2: tile t + 2 is a main tile for certain
  auto ktile = [&](int t, auto h1_tag, auto h2_tag, auto main2_tag) {
    constexpr bool H1 = decltype(h1_tag)::value, H2 = decltype(h2_tag)::value;
    const char* cur = smem + (t & 1) * STG;
    const char* nxt = smem + ((t + 1) & 1) * STG;
    // ---- substep 0: MFMAs from set 0, second-substep fragments into set 1 (the reads end well in front of the barrier's lgkmcnt(0))
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      w4_mfma(acc[k / 8][k % 8], fb[0][k % 8], fa[0][k / 8]);
      if (k % RSTEP == 1 && k / RSTEP < NF && !(W4X & 2)) fread(cur, 1, 1, k / RSTEP);
      if (!(W4X & 256)) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (H1 && !(W4X & 4)) {
      // every read of the current stage has returned; this wave's pieces of tile t + 1 have landed; then everybody's
      if constexpr (W4X & 2048) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // timing experiment: the DMA is never waited for (races)
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if constexpr (W4X & 1024) {
#pragma unroll
        for (int q = 0; q < NP; ++q) asm volatile("" :: "v"(sink[q]));
      }
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- substep 1: MFMAs from set 1, tile t + 1's first fragments into set 0, tile t + 2's DMA into the vacated stage
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      w4_mfma(acc[k / 8][k % 8], fb[1][k % 8], fa[1][k / 8]);
      if constexpr (H1) { if (k % RSTEP == 1 && k / RSTEP < NF && !(W4X & 2)) fread(nxt, 0, 0, k / RSTEP); }
      if constexpr (H2) { if (k % PSTEP == PSTEP - 1 && k / PSTEP < NP && !(W4X & 1)) piece(t + 2, k / PSTEP, main2_tag); }
      if (!(W4X & 256)) __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto ext_scale = [&]() {
    if (p.drop_p > 0.f) {
#pragma unroll
      for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          gemm_ext_scale4<true>(p, row0 + wm * WR + i * 16 + frow, n0 + wn * 128 + j * 16 + fq * 4, acc[i][j]);
    } else if (p.alpha2 != 1.f) {
#pragma unroll
      for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] *= p.alpha2;
    }
  };

  // prologue: tiles 0 and 1 on their way, tile 0 landed, its first fragments read
#pragma unroll
  for (int q = 0; q < NP; ++q) piece(0, q, std::false_type{});
#pragma unroll
  for (int q = 0; q < NP; ++q) piece(1, q, std::false_type{});
  if constexpr (NP == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int f = 0; f < NF; ++f) fread(smem, 0, 0, f);

  int t = 0;
  // (the scale of the LoRA extension sits BETWEEN the loops, never inside one: see the eight-wave form. The launcher sends a call here only
  // with >= 2 main K-tiles behind the extension, so every extension tile has two successors and runs the full body)
  if (kt_ext > 0) {
    for (; t < kt_ext; ++t) ktile(t, std::true_type{}, std::true_type{}, std::false_type{});
    w4_mfma_drain();
    ext_scale();
  }
  for (; t + 2 < kt_total; ++t) ktile(t, std::true_type{}, std::true_type{}, std::true_type{});
  ktile(t, std::true_type{}, std::false_type{}, std::true_type{});
  ktile(t + 1, std::false_type{}, std::false_type{}, std::true_type{});
  w4_mfma_drain();
  __builtin_amdgcn_s_barrier();          // every wave is past its last LDS read: the stages become the output slabs

  const void* bias = seg ? p.bias1 : p.bias0;
  if (W4X & 8) {          // (every accumulator stays live: guide rule 17)
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 123.456f) ((float*)p.C)[0] = 1.f;
    return;
  }
  if (OUT_F32) {
#pragma unroll
    for (int i = 0; i < TA; ++i) {
      const int ml = wm * WR + i * 16 + frow;
      if (ml >= nrows) continue;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int nl = wn * 128 + j * 16 + fq * 4;
        if (nl >= ncols) continue;
        gemm_store4<true>(p, bias, row0 + ml, n0 + nl, ncols - nl, acc[i][j]);
      }
    }
  } else {
    constexpr int HR = 8 * TA;            // rows of one slab pass (half a wave row): 64 or 48
    typedef EpiSlab<HR, 128> Slab;
    char* slab = smem + wave * Slab::BYTES;
    const bool plain = bias == nullptr && p.act == VM_ACT_NONE;
    const bool fast_bias = bias != nullptr && p.act == VM_ACT_NONE && ncols - wn * 128 >= 128 && ((uintptr_t)bias & 7) == 0;
    f32x4_t bv[8];
    if (fast_bias) {
#pragma unroll
      for (int j = 0; j < 8; ++j) bv[j] = epi_bias4(bias, n0 + wn * 128 + j * 16 + fq * 4);
    }
    // (the two passes are two calls with a compile-time `half`: a `for (half)` loop the unroller gives up on turns every accumulator index into
    // a run-time one, and the 256 accumulators into a scratch array that each MFMA of the K loop then writes through)
    // (MODE is a compile-time tag too and the dispatch sits OUTSIDE the loops: with `if (fast_bias) .. else if (plain) ..` inside the loop body hipcc
    // merged the three forms into one per-element branch ladder — 460 instructions and 35 branches per 4 outputs, 14 us per tile)
    auto pass = [&](auto half_tag, auto mode_tag) {
      constexpr int half = decltype(half_tag)::value, MODE = decltype(mode_tag)::value;
#pragma unroll
      for (int i = 0; i < TA / 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const f32x4_t& a = acc[half * (TA / 2) + i][j];
          if constexpr (W4X & 32) { asm volatile("" :: "v"(a)); continue; }
          if constexpr (MODE == 2) epi_put4<2>(slab, Slab::PITCH, i * 16 + frow, j * 16 + fq * 4, p, bias, 0, 4, a, bv[j]);
          else if constexpr (MODE == 0) epi_put4<0>(slab, Slab::PITCH, i * 16 + frow, j * 16 + fq * 4, p, bias, 0, 4, a);
          else {
            const int nl = wn * 128 + j * 16 + fq * 4;
            epi_put4<1>(slab, Slab::PITCH, i * 16 + frow, j * 16 + fq * 4, p, bias, n0 + nl, ncols - nl, a);
          }
        }
      if constexpr (!(W4X & 16)) {
        const int rows_left = nrows - wm * WR - half * HR, cols_left = ncols - wn * 128;
        const bool vec_ok = (p.ldc % 8 == 0) && (!p.residual || p.ldr % 8 == 0);
        if (rows_left >= HR && cols_left >= 128 && vec_ok) w4_flush_full<HR>(slab, p, row0 + wm * WR + half * HR, n0 + wn * 128, lane);
        else epi_flush<HR, 128>(slab, p, row0 + wm * WR + half * HR, n0 + wn * 128, rows_left, cols_left, lane);
      }
    };
    const std::integral_constant<int, 0> h0{};
    const std::integral_constant<int, 1> h1{};
    if (fast_bias) { pass(h0, std::integral_constant<int, 2>{}); pass(h1, std::integral_constant<int, 2>{}); }
    else if (plain) { pass(h0, std::integral_constant<int, 0>{}); pass(h1, std::integral_constant<int, 0>{}); }
    else { pass(h0, std::integral_constant<int, 1>{}); pass(h1, std::integral_constant<int, 1>{}); }
  }
}

template <bool OUT_F32, int TA>
__global__ __launch_bounds__(256, 1) void gemm256w_k(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tm, tn;
  gemm_tile_id(p, tm, tn);
  gemm256w_tile<OUT_F32, TA>(p, smem, threadIdx.x, tm, tn);
}

