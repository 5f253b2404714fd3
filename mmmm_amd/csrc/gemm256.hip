// 256x256x64 bf16 NT GEMM for gfx950 — the large-shape variant of gemm_nt_k (same arguments / epilogue).
//
// Structure (after the guide's "256² 8-phase" principles, own schedule):
//   * 512 threads = 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 of C = 8 x 4 tiles of v_mfma_f32_16x16x32_bf16
//     (128 accumulator VGPRs); one workgroup per CU, 128 KiB of LDS = 2 stages x 4 half-tiles x 16 KiB.
//   * a K-tile is consumed in 4 phases, one C-quadrant (64 x 32 per wave, 16 MFMAs) each; the operand tiles are cut
//     into HALF-TILES along the *wave-local* halves (A-h0 = the first 64 rows of every wave row, B-h1 = the second 32
//     columns of every wave column, ...) so that phase 1 needs {A-h0, B-h0}, phase 2 {B-h1}, phase 3 {A-h1}, phase 4
//     nothing new. Each phase issues the LDS-DMA (buffer_load ... lds, 2 x 16 B per lane) of ONE half-tile of the
//     NEXT K-tile, in the order A-h0', B-h0', B-h1', A-h1': every half-tile is issued >= 3 phases before its first
//     read and overwrites LDS that was last read >= 4 phases earlier.
//   * loads stay in flight ACROSS the raw s_barriers: the only waits are counted s_waitcnt vmcnt(4)
//     (two half-tiles may remain outstanding), never vmcnt(0) inside the steady-state loop.
//   * the two waves of a SIMD (wave rows wm = 0 / 1) are staggered by one barrier interval so that one is in its
//     MFMA segment while the other is in its LDS-read / DMA-issue segment.
#include <mutex>
#include "vm_common.hpp"
#include "gemm_common.hpp"
#include "vm_tile.hpp"
#include <type_traits>

namespace {

#ifndef VM_GEMM_W4_DEFAULT
#define VM_GEMM_W4_DEFAULT 3
#endif
constexpr int W4_LDS_BYTES = 160 * 1024;   // four-wave form: activation two K-tiles deep, weight three (the output slabs overlay its third stage)
constexpr int HALF_BYTES = 128 * 128;      // 128 rows x 128 B (64 bf16)
constexpr int STAGE_BYTES2 = 4 * HALF_BYTES;
constexpr int LDS_BYTES2 = 2 * STAGE_BYTES2;

// LDS-DMA of one half-tile: 16 wave-instructions of 1 KiB (8 rows), 2 per wave. KIND 0: activation (A-h{half}); with MI
// 16-row fragments per wave and half (MI = 4: 256-row tile, MI = 3: 192-row tile) LDS row r -> tile row
// (r / (16 MI)) * 32 MI + half * 16 MI + r % (16 MI); the LDS rows past 32 MI (MI = 3: the last 4 instructions) are issued
// with an out-of-range offset — zero fill, no memory traffic — so every wave still issues two loads and the counted
// vmcnt waits stay uniform. KIND 1: weight (B-h{half}), r -> (r>>5)*64 + half*32 + (r&31).
template <int KIND, int MI = 4>
__device__ __forceinline__ void stage_half(__amdgpu_buffer_rsrc_t rsrc, int ld_bytes, int koff, int half, char* lds_half,
                                           int wave, int lane) {
  const int r8 = lane >> 3, slot = lane & 7;
  const int chunk = slot ^ r8;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int inst = wave * 2 + i;
    const int r = inst * 8 + r8;
    int voff;
    if (KIND == 0) {
      constexpr int HR = 16 * MI;                 // rows of one wave-row inside a half
      const int trow = (r / HR) * (2 * HR) + half * HR + r % HR;
      voff = (MI == 4 || r < 2 * HR) ? trow * ld_bytes + chunk * 16 : 0x40000000;
    } else {
      voff = ((r >> 5) * 64 + half * 32 + (r & 31)) * ld_bytes + chunk * 16;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds_half + inst * 1024), 16, voff, koff, 0, 0);
  }
}

// NN form of the weight operand (dgrad dx = dy W with W [N, K] as stored: the contraction index is the ROW of the stored matrix).
// LDS-DMA of one half-tile = 64 contraction rows x (4 wave columns x 32 output columns) bf16 = [64][256 B], image (b) of the guide's
// transposed-read recipe (vm_tile.hpp: tile_off / swz), 16 wave-instructions of 1 KiB = 4 rows each, 2 per wave. `row0` = first
// contraction row of the K-tile, `cols_valid` = output columns of this tile that exist (chunks past them read zero).
__device__ __forceinline__ void stage_half_nn(__amdgpu_buffer_rsrc_t rsrc, int ld_bytes, int row0, int half, int cols_valid, char* lds_half,
                                              int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int inst = wave * 2 + i;
    const int row = inst * 4 + (lane >> 4);
    const int chunk = (lane & 15) ^ swz(row);
    const int col = (chunk >> 2) * 64 + half * 32 + (chunk & 3) * 8;           // output column inside the 256-column tile
    const int voff = (col + 8 <= cols_valid) ? (row0 + row) * ld_bytes + col * 2 : 0x7FFFFFF0;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds_half + inst * 1024), 16, voff, 0, 0, 0);
  }
}

// weight fragment of the 16x16x32 MFMA out of the NN image: lane (frow = column, fq) gets rows kbase + 8 fq + 0..7 of column
// 16-block `c0` (in 16-byte chunks: c0, c0 + 1) through two ds_read_b64_tr_b16 (4 rows x 16 columns per 16-lane group each)
__device__ __forceinline__ bf16x8_t frag_tr16(const char* tile, int kbase, int c0, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int q = i >> 2, pp = i & 3;
  const int chunk = c0 + (pp >> 1);
  const int rowA = kbase + 8 * g + q, rowB = rowA + 4;
  const char* a = tile + tile_off(rowA, chunk) + ((pp & 1) << 3);
  const char* c = tile + tile_off(rowB, chunk) + ((pp & 1) << 3);
  u16x4_t lo = __builtin_bit_cast(u16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a)));
  u16x4_t hi = __builtin_bit_cast(u16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(c)));
  u16x8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, r);
}

#define VM_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// experiment switches (tools only; the shipped build defines none of them). Measured on the step's shapes: without the
// post-MFMA barrier -5 %, without the row stagger -8..12 %, s_setprio around the MFMA clusters -1.5 % (removed).
#ifdef VM_G256_NO_POST_BARRIER
#define POST_MMA_BARRIER()
#else
#define POST_MMA_BARRIER() __builtin_amdgcn_s_barrier()
#endif

// epilogue shared by the two 256x256 kernels: accumulators -> (bias, residual) -> C. Every wave must have passed its last
// LDS read (the staging memory is reused for the per-wave output slabs).
template <bool OUT_F32, int MI>
__device__ __forceinline__ void epilogue256(const GemmParams& p, f32x4_t (&acc)[2 * MI][4], char* smem, int wave, int lane, int wm, int wn,
                                            int row0, int nrows, int n0, int ncols, int seg) {
  constexpr int WR = 32 * MI;          // rows of one wave-row (128 or 96)
  constexpr int HR = 16 * MI;          // rows of one slab pass
  const int frow = lane & 15, fq = lane >> 4;
  if VM_DBG(p, 32) { if (acc[0][0][0] == 123.456f) ((float*)p.C)[0] = 1.f; return; }   // timing experiment: no C traffic
  const void* bias = seg ? p.bias1 : p.bias0;
  if (OUT_F32) {
#pragma unroll
    for (int i = 0; i < 2 * MI; ++i) {
      const int ml = wm * WR + i * 16 + frow;
      if (ml >= nrows) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nl = wn * 64 + j * 16 + fq * 4;
        if (nl >= ncols) continue;
        gemm_store4<true>(p, bias, row0 + ml, n0 + nl, ncols - nl, acc[i][j]);
      }
    }
  } else {
    // every wave has passed the final barrier: LDS is free. (16 MI) x 64 slab per wave, two passes (upper / lower rows).
    typedef EpiSlab<HR, 64> Slab;
    char* slab = smem + wave * Slab::BYTES;
    const bool plain = bias == nullptr && p.act == VM_ACT_NONE;
    const bool fast_bias = bias != nullptr && p.act == VM_ACT_NONE && ncols - wn * 64 >= 64 && ((uintptr_t)bias & 7) == 0;
    f32x4_t bv[4];
    if (fast_bias) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = epi_bias4(bias, n0 + wn * 64 + j * 16 + fq * 4);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (fast_bias) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            epi_put4<2>(slab, Slab::PITCH, i * 16 + frow, j * 16 + fq * 4, p, bias, 0, 4, acc[half * MI + i][j], bv[j]);
      } else if (plain) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            epi_put4<0>(slab, Slab::PITCH, i * 16 + frow, j * 16 + fq * 4, p, bias, 0, 4, acc[half * MI + i][j]);
      } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int nl = wn * 64 + j * 16 + fq * 4;
            epi_put4<1>(slab, Slab::PITCH, i * 16 + frow, j * 16 + fq * 4, p, bias, n0 + nl, ncols - nl, acc[half * MI + i][j]);
          }
      }
      epi_flush<HR, 64>(slab, p, row0 + wm * WR + half * HR, n0 + wn * 64, nrows - wm * WR - half * HR, ncols - wn * 64, lane);
    }
  }
}

// MI = 16-row activation fragments per wave and half: 4 -> 256 x 256 tile, 3 -> 192 x 256 tile (same LDS image, same phase
// structure, 12 instead of 16 MFMAs per phase). The 192-row tile exists for tile-count quantisation: [6280 x 1792] is 175
// tiles of 256 rows (68 % of one round over 256 CUs) but 231 tiles of 192 rows, [6280 x 5376] is 525 (3 rounds) vs 693
// (3 rounds of 0.75 the work).
//
// gemm256_segment: output tile (tm, tn) over the MAIN K-tiles [m_begin, m_end) (the LoRA extension tiles belong to the segment
// that starts at main tile 0). ROLE 0: the whole K range -> epilogue (what the one-tile-per-workgroup kernel runs). The stream-K
// kernel (gemm256sk_k below) also uses ROLE 1: a K range that does not start at 0 — the fp32 accumulators go to this worker's slab
// and its flag is published; ROLE 2: a K range [0, m_end) with m_end short of the tile's K — the workers [sk_w0, sk_w1) hold the
// rest: their slabs are added in worker order (a fixed order: the sum does not depend on who finished when), then the epilogue.
template <bool OUT_F32, int MI, bool F8, bool BNN, bool SK>
__device__ __forceinline__ void gemm256_segment(const GemmParams& p, char* smem, const int tid, int tm, int tn, int m_begin, int m_end, int role,
                                                int sk_self, int sk_w0, int sk_w1) {
  static_assert(!(F8 && BNN), "the NN weight form exists for bf16 only");
  constexpr int BMT = 64 * MI;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  int row0, nrows, seg;
  gemm_tile_rows<BMT>(p, tm, row0, nrows, seg);
  if (nrows <= 0) return;
  const int n0 = tn * 256;
  const int ncols = min(256, p.N - n0);
  const char* Bw = seg ? p.B1 : p.B0;
  const char* B2w = seg ? p.B2_1 : p.B2_0;

  const int lda_b = (int)p.lda * 2, ldb_b = (int)p.ldb * 2;
  const __amdgpu_buffer_rsrc_t rA = make_rsrc(p.A, (int64_t)row0 * lda_b, VM_DBG(p, 1) ? 0 : nrows * lda_b);
  // NT: rows n0 .. n0 + ncols of B [N, K]; NN: all K contraction rows of B [K, N], shifted to output column n0
  const __amdgpu_buffer_rsrc_t rB = BNN ? make_rsrc(Bw, (int64_t)n0 * 2, VM_DBG(p, 1) ? 0 : (int)(((int64_t)p.K * ldb_b - (int64_t)n0 * 2)))
                                        : make_rsrc(Bw, (int64_t)n0 * ldb_b, VM_DBG(p, 1) ? 0 : ncols * ldb_b);
  // F8: the main operands are e4m3 bytes — a 128-byte LDS row holds 128 k instead of 64, everything else (DMA, swizzle, phases) is
  // unchanged; the LoRA extension tiles stay bf16
  // (SK: only the segment that starts at main K-tile 0 carries the extension tiles; a segment always holds >= 1 main tile)
  const int kt_ext = (!SK || m_begin == 0) ? p.K2 / 64 : 0, kt_main = SK ? m_end - m_begin : (F8 ? p.K / 128 : p.K / 64), kt_total = kt_ext + kt_main;
  const int m_first = (SK && !VM_DBG(p, 1024)) ? m_begin : 0;      // (timing experiment 1024: every segment reads from k = 0 — aligned K phases, wrong results)
  __amdgpu_buffer_rsrc_t rA2 = rA, rB2 = rB;
  int lda2_b = 0, ldb2_b = 0;
  if (kt_ext > 0) {
    lda2_b = (int)p.lda2 * 2; ldb2_b = (int)p.ldb2 * 2;
    rA2 = make_rsrc(p.A2, (int64_t)row0 * lda2_b, nrows * lda2_b);
    rB2 = make_rsrc(B2w, (int64_t)n0 * ldb2_b, ncols * ldb2_b);
  }

  // half-tile h of K-tile t into stage (t & 1): h = 0 A-h0, 1 A-h1, 2 B-h0, 3 B-h1
  auto stage = [&](int t, int h) {
    char* dst = smem + (t & 1) * STAGE_BYTES2 + h * HALF_BYTES;
    const bool ext = t < kt_ext;
    const int koff = (ext ? t : t - kt_ext + m_first) * 128;
    if (h < 2) stage_half<0, MI>(ext ? rA2 : rA, ext ? lda2_b : lda_b, koff, h, dst, wave, lane);
    else if (BNN && !ext) stage_half_nn(rB, ldb_b, (t - kt_ext + m_first) * 64, h - 2, ncols, dst, wave, lane);
    else stage_half<1>(ext ? rB2 : rB, ext ? ldb2_b : ldb_b, koff, h - 2, dst, wave, lane);
  };

  f32x4_t acc[2 * MI][4];   // [m-tile of the wave's 32 MI rows][n-tile of its 64 columns]
#pragma unroll
  for (int i = 0; i < 2 * MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t aF[MI][2];      // activation fragments of the current m-half: [m-tile][k-substep]
  bf16x8_t bF[2][2][2];    // weight fragments: [n-half][n-tile][k-substep]

  const int frow = lane & 15, fq = lane >> 4;
  const int slot0 = fq ^ (frow & 7);
  const int off_k0 = frow * 128 + slot0 * 16;
  const int off_k1 = frow * 128 + (slot0 ^ 4) * 16;

  auto read_a = [&](const char* st, int mh) {   // wave's rows of A-h{mh}: wm*16*MI + i*16 + frow
    const char* base = st + mh * HALF_BYTES + (wm * 16 * MI) * 128;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      aF[i][0] = *reinterpret_cast<const bf16x8_t*>(base + i * 2048 + off_k0);
      aF[i][1] = *reinterpret_cast<const bf16x8_t*>(base + i * 2048 + off_k1);
    }
  };
  auto read_b = [&](const char* st, int nh) {   // wave's rows of B-h{nh}: wn*32 + j*16 + frow
    const char* base = st + (2 + nh) * HALF_BYTES + (wn * 32) * 128;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bF[nh][j][0] = *reinterpret_cast<const bf16x8_t*>(base + j * 2048 + off_k0);
      bF[nh][j][1] = *reinterpret_cast<const bf16x8_t*>(base + j * 2048 + off_k1);
    }
  };
  // NN image of B-h{nh}: column 16-blocks wn * 4 + 2 j (in chunks), rows 32 ks + 8 fq + 0..7 as two 4-row transposed reads. The eight
  // per-lane byte offsets (j, ks, upper / lower 4 rows) are loop invariants: computed once here, the halves and stages differ by constants.
  int nn_off[2][2][2];
  if constexpr (BNN) {
    const int gi = lane & 15, gg = lane >> 4, gq = gi >> 2, gp = gi & 3;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
          nn_off[j][ks][hl] = tile_off(32 * ks + 8 * gg + gq + 4 * hl, wn * 4 + 2 * j + (gp >> 1)) + ((gp & 1) << 3);
  }
  auto read_b_nn = [&](const char* st, int nh) {
    const char* base = st + (2 + nh) * HALF_BYTES;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const u16x4_t lo = __builtin_bit_cast(u16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + nn_off[j][ks][0])));
        const u16x4_t hi = __builtin_bit_cast(u16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + nn_off[j][ks][1])));
        const u16x8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        bF[nh][j][ks] = __builtin_bit_cast(bf16x8_t, r);
      }
  };
  auto mma = [&](int mh, int nh, auto f8_tag) {
    if constexpr (decltype(f8_tag)::value) {
      // one v_mfma_f32_16x16x128_f8f6f4 (e4m3 x e4m3, unit block scales) per accumulator tile and K-tile: twice the bf16 matrix rate.
      // A lane's two 16-byte fragment halves are the k-chunks fq and fq + 4 of the row — not 32 consecutive k, but the weight and the
      // activation operand use the SAME k permutation, so the sum over k is complete.
      typedef __attribute__((ext_vector_type(8))) int i32x8_t;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const i32x4_t a0 = __builtin_bit_cast(i32x4_t, aF[i][0]), a1 = __builtin_bit_cast(i32x4_t, aF[i][1]);
        const i32x8_t av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const i32x4_t b0 = __builtin_bit_cast(i32x4_t, bF[nh][j][0]), b1 = __builtin_bit_cast(i32x4_t, bF[nh][j][1]);
          const i32x8_t bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
          acc[mh * MI + i][nh * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bv, av, acc[mh * MI + i][nh * 2 + j], 0, 0, 0, 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[mh * MI + i][nh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bF[nh][j][ks], aF[i][ks], acc[mh * MI + i][nh * 2 + j], 0, 0, 0);
    }
  };

  // Wave rows are STAGGERED by one barrier interval: the two waves that share a SIMD (wave w and w+4, i.e. wm = 0
  // and wm = 1) alternate between the LDS/DMA segment R_p and the MFMA segment M_p of a phase, so the matrix pipe of
  // every SIMD always has one wave feeding it (guide "Two waves per SIMD", item 9 / 8-phase template `if (wr==1)`).
  // Timeline in barrier intervals: row 0 runs R_p in interval 2p and M_p in 2p+1; row 1 runs R_p in 2p+1, M_p in 2p+2.
  // A half-tile issued in phase p is first read by row 0 in interval 2(p+3); every wave must have retired its part one
  // barrier earlier, i.e. by the end of interval 2p+5 = end of R_{p+2} for row 1. The same counted wait at the end of
  // EVERY wave's R segment is sufficient for both rows (row 0 merely retires one interval early) and keeps the steady
  // state free of wave-dependent branches. The loop is peeled (HN = "a next K-tile exists") so it has no data-dependent
  // branches between MFMA clusters either.
  auto ktile = [&](int t, auto hn_tag, auto f8_tag, auto nn_tag) {
    constexpr bool HN = decltype(hn_tag)::value;
    constexpr bool NNB = decltype(nn_tag)::value;      // this K-tile's weight image is the NN one (main tiles of a BNN launch)
    const char* st = smem + (t & 1) * STAGE_BYTES2;
    // ---- phase 1: quadrant (0,0); needs A-h0, B-h0; issues A-h0'; retires B-h1 of this tile
    read_a(st, 0);
    if constexpr (NNB) read_b_nn(st, 0); else read_b(st, 0);
    if (HN) stage(t + 1, 0);
    if (HN) VM_WAIT_VMCNT(4); else VM_WAIT_VMCNT(2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    mma(0, 0, f8_tag);
    __builtin_amdgcn_sched_barrier(0);
    POST_MMA_BARRIER();
    // ---- phase 2: quadrant (0,1); needs B-h1; issues B-h0'; retires A-h1 of this tile
    if constexpr (NNB) read_b_nn(st, 1); else read_b(st, 1);
    if (HN) stage(t + 1, 2);
    if (HN) VM_WAIT_VMCNT(4); else VM_WAIT_VMCNT(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    mma(0, 1, f8_tag);
    __builtin_amdgcn_sched_barrier(0);
    POST_MMA_BARRIER();
    // ---- phase 3: quadrant (1,1); needs A-h1; issues B-h1'
    read_a(st, 1);
    if (HN) stage(t + 1, 3);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    mma(1, 1, f8_tag);
    __builtin_amdgcn_sched_barrier(0);
    POST_MMA_BARRIER();
    // ---- phase 4: quadrant (1,0); B-h0 fragments are still in registers; issues A-h1'; retires A-h0', B-h0'
    if (HN) { stage(t + 1, 1); VM_WAIT_VMCNT(4); }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    mma(1, 0, f8_tag);
    __builtin_amdgcn_sched_barrier(0);
    POST_MMA_BARRIER();
  };
  auto ext_scale = [&]() {
    if (p.drop_p > 0.f) {
#pragma unroll
      for (int i = 0; i < 2 * MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          gemm_ext_scale4<true>(p, row0 + wm * 32 * MI + i * 16 + frow, n0 + wn * 64 + j * 16 + fq * 4, acc[i][j]);
    } else if (p.alpha2 != 1.f) {      // (rsLoRA with r = 64, alpha = 8: the scale is exactly 1 — nothing to do)
#pragma unroll
      for (int i = 0; i < 2 * MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] *= p.alpha2;
    }
  };

  // prologue: K-tile 0 completely
  stage(0, 0); stage(0, 2); stage(0, 3); stage(0, 1);
  VM_WAIT_VMCNT(0);
  __builtin_amdgcn_s_barrier();
#ifndef VM_G256_NO_STAGGER
  if (wm == 1) __builtin_amdgcn_s_barrier();
#endif

  // extension tiles (LoRA rank slab; the main K always follows), the scale between the two loops — never inside one:
  // with the scale in the loop body the compiler hoists the 128 loop-invariant mask hashes and spills them
  int t = 0;
  const std::integral_constant<bool, F8> main_kind{};
  const std::integral_constant<bool, BNN> main_nn{};
  if (kt_ext > 0) {
    for (; t < kt_ext; ++t) ktile(t, std::true_type{}, std::false_type{}, std::false_type{});
    ext_scale();
  }
  for (; t + 1 < kt_total; ++t) ktile(t, std::true_type{}, main_kind, main_nn);
  ktile(kt_total - 1, std::false_type{}, main_kind, main_nn);
#ifndef VM_G256_NO_STAGGER
  if (wm == 0) __builtin_amdgcn_s_barrier();
#endif

  if constexpr (SK) {
    // every wave is past its last LDS read and holds no vector-memory operation in flight (the last K-tile waited vmcnt(0))
    constexpr int SLAB_BYTES = 8 * (8 * MI) * 1024;       // 8 waves x (2 MI x 4) accumulator registers x 64 lanes x 16 B, in register order
    if (role == 1) {
      // contributor: accumulators -> slab[sk_self] with write-through (sc1) stores, 1 KiB contiguous per wave-instruction; every wave
      // drains its stores, the workgroup meets, ONE lane publishes the epoch (guide "Workgroup dispatch ... inter-workgroup visibility":
      // sc1 payload -> vmcnt(0) of every storing wave -> barrier -> sc1 flag; the consumer polls with sc1 loads and reads with sc1 loads)
      const __amdgpu_buffer_rsrc_t rS = make_rsrc((const char*)p.sk_slabs, (int64_t)sk_self * SLAB_BYTES, VM_DBG(p, 256) ? 0 : SLAB_BYTES);
#pragma unroll
      for (int i = 0; i < 2 * MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, acc[i][j]), rS, ((wave * (8 * MI) + i * 4 + j) * 64 + lane) * 16, 0, 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (tid == 0) __hip_atomic_store(p.sk_flags + sk_self, p.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (role == 2 && !VM_DBG(p, 512)) {
      for (int w = sk_w0; w < sk_w1; ++w) {
        if (tid == 0) {
          // bounded spin (the contributor ran its slab FIRST, long before this owner reached the end of its range): give up rather than hang
          int spins = 0;
          while (__hip_atomic_load(p.sk_flags + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.sk_epoch) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1 << 22)) { __hip_atomic_store(p.sk_flags + p.sk_workers, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
          }
        }
        __builtin_amdgcn_s_barrier();
        const __amdgpu_buffer_rsrc_t rS = make_rsrc((const char*)p.sk_slabs, (int64_t)w * SLAB_BYTES, SLAB_BYTES);
#pragma unroll
        for (int i = 0; i < 2 * MI; ++i) {
          f32x4_t part[4];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            part[j] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rS, ((wave * (8 * MI) + i * 4 + j) * 64 + lane) * 16, 0, 16));
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] += part[j];
        }
      }
    }
  }

  if constexpr (F8) {
    // dequantise: acc[m][n] *= row_scale[m] * col_scale[n] (the extension operands were pre-divided by the same scales on the host)
    const float* cs = seg ? p.col_scale1 : p.col_scale0;
    f32x4_t cv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fq * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) cv[j][r] = (n + r < p.N) ? cs[n + r] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 2 * MI; ++i) {
      const int ml = wm * 32 * MI + i * 16 + frow;
      const float rs = ml < nrows ? p.row_scale[row0 + ml] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] *= cv[j] * rs;
    }
  }
  epilogue256<OUT_F32, MI>(p, acc, smem, wave, lane, wm, wn, row0, nrows, n0, ncols, seg);
}

#include "gemm256w.hpp"

template <bool OUT_F32, int MI, bool F8 = false, bool BNN = false>
__global__ __launch_bounds__(512, 2) void gemm256_k(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tm, tn;
  gemm_tile_id(p, tm, tn);
  gemm256_segment<OUT_F32, MI, F8, BNN, false>(p, smem, threadIdx.x, tm, tn, 0, 0, 0, 0, 0, 0);
}

// Stream-K form: ONE workgroup per CU (grid = p.sk_workers) walks a list of segments.
//   * real tile count T from the device-side row counts (the token-routed two-segment form launches an upper bound);
//   * the first Tsk = T mod W (+ W when T >= W) tiles are the stream-K region: their Tsk x km main K-tiles are cut into W equal
//     contiguous ranges, worker v owns [v U / W, (v + 1) U / W). A range that starts inside a tile begins with a ROLE-1 segment
//     (slab + flag, done FIRST), a range that ends inside a tile ends with a ROLE-2 segment (done LAST: by then the slabs it needs were
//     published a whole range ago — the spin is a formality), whole tiles in between are ROLE 0. With U / W >= km (T >= W) a tile has
//     at most two parts;
//   * the remaining T - Tsk tiles (a multiple of W) are data-parallel rounds: worker v runs tiles Tsk + v + W j, whole.
// Worker ids are the XCD-contiguous remap of blockIdx, tiles the GROUP_M-grouped order of gemm_tile_id: workers that share an L2 run
// neighbouring tiles, in the stream-K region and in the rounds.
template <bool OUT_F32, int MI, bool F8 = false>
__global__ __launch_bounds__(512, 2) void gemm256sk_k(const GemmParams p) {
  constexpr int BMT = 64 * MI;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int W = gridDim.x;
  int v = blockIdx.x;
  { const int q = W >> 3, r = W & 7, xcd = v & 7; v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3); }
  int M = p.M, split = p.split;
  if (p.counts_dev) {
    split = __builtin_amdgcn_readfirstlane(p.counts_dev[0]);
    M = min(p.M, __builtin_amdgcn_readfirstlane(p.counts_dev[1]));
  }
  int tiles_m;
  if (split < 0) tiles_m = (M + BMT - 1) / BMT;
  else { split = min(split, M); tiles_m = (split + BMT - 1) / BMT + (M - split + BMT - 1) / BMT; }
  const int tiles_n = p.tiles_n;
  const int T = tiles_m * tiles_n;
  const int km = F8 ? p.K / 128 : p.K / 64;
  int Tsk = (T % W) + ((T >= W && (T % W)) ? W : 0);
  if ((int64_t)Tsk * km < W) Tsk = 0;      // fewer K-tiles than workers (the device-side row count came out tiny): whole tiles only
  auto tile_of = [&](int id, int& tm, int& tn) {
    const int per_group = GROUP_M * tiles_n;
    const int g = id / per_group, gm0 = g * GROUP_M, gsz = min(GROUP_M, tiles_m - gm0), rem = id - g * per_group;
    tm = gm0 + rem % gsz;
    tn = rem / gsz;
  };
  auto start_of = [&](int w) { return (int)(((int64_t)w * Tsk * km) / W); };
  // ONE loop over the worker's segments (one inlined copy of the tile body): the stream-K range first, then the rounds
  int u = start_of(v);
  const int u_end = start_of(v + 1);
  int dp_tile = Tsk + v;
  while (true) {
    int tile, m0 = 0, m1 = km, role = 0, w0 = 0, w1 = 0;
    if (u < u_end) {
      tile = u / km; m0 = u - tile * km; m1 = min(km, m0 + (u_end - u));
      if (m0 > 0) role = 1;
      else if (m1 < km) {
        // the workers behind this one whose ranges start inside the tile (each non-empty: the launcher requires U >= W)
        role = 2; w0 = v + 1; w1 = w0;
        const int tile_end = (tile + 1) * km;
        while (w1 < W && start_of(w1) < tile_end) ++w1;
      }
      u += m1 - m0;
    } else if (dp_tile < T) {
      tile = dp_tile; dp_tile += W;
    } else break;
    int tm, tn;
    tile_of(tile, tm, tn);
    // Nothing the tile body derives from the lane id or from the parameters may be hoisted out of THIS loop: hoisted, the K loop's and the
    // epilogue's address registers are all live at once (measured: 103 VGPR + 108 SGPR spill slots on a body that sits at 256 VGPRs). The lane
    // id is laundered and the parameters are re-read from the kernarg segment per segment (scalar loads, ~50 dwords).
    int tid_l = threadIdx.x;
    asm volatile("" : "+v"(tid_l));
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) GemmParams* kparams_t;
    kparams_t kp = (kparams_t)__builtin_amdgcn_kernarg_segment_ptr();      // `p` is the kernel's only argument: offset 0 of the segment
    asm volatile("" : "+s"(kp));
    const GemmParams q = *kp;
#else
    const GemmParams q = p;
#endif
    gemm256_segment<OUT_F32, MI, F8, false, true>(q, smem, tid_l, tm, tn, m0, m1, role, v, w0, w1);
    __builtin_amdgcn_s_barrier();          // the epilogue's slab reads are done before the next segment stages into the same LDS
  }
}

}  // namespace

// internal (tools/ubench/gemm_w4_bench, tests, kernels.GEMM_W4): 0 = the eight-wave form everywhere, 1 = bf16 NT launches run the four-wave form (gemm256w_k),
// 2 = only the launches the scheduler runs on 192-row tiles do (the 192-row body has no register spills and wins 3-8 % with the weight coming from HBM;
// the 256-row body loses to the eight-wave form as soon as the launch carries a LoRA extension WITH a scale or a dropout mask: profiles/r6_gemm_w4.txt),
// 3 = mode 2 + the 256-row launches that need no scale / mask on the extension (forward launches)
static int& w4_mode() { static int mode = VM_GEMM_W4_DEFAULT; return mode; }
extern "C" int vm_gemm_w4_mode_(int mode) { if (mode < 0 || mode > 3) return VM_ERR_BAD_ARG; w4_mode() = mode; return VM_OK; }
extern "C" int vm_gemm_w4_mode_get_(void) { return w4_mode(); }

// called by gemm_launch (gemm.hip) when the shape fills the chip with 256x256 tiles; f8 != 0: e4m3 main operands (vm_gemm_fp8)
extern "C" int vm_gemm256_launch_(const void* params, int out_f32, int segmented, int tile_rows, int f8, void* stream) {
  // f8: 0 bf16 NT, 1 e4m3 NT, 2 bf16 with the weight in NN form (p.b_nn)
  GemmParams p = *(const GemmParams*)params;
  if (tile_rows != 256 && tile_rows != 192) return VM_ERR_BAD_ARG;
  p.tiles_m = p.tiles_m_override > 0 ? p.tiles_m_override : (p.M + tile_rows - 1) / tile_rows + (segmented ? 1 : 0);
  p.tiles_n = (p.N + 255) / 256;
  static std::once_flag attr_once;          // (called from the main thread and from autograd's backward thread)
  static bool attr_ok = false;
  std::call_once(attr_once, [] {
    const void* fns[14] = {(const void*)gemm256_k<false, 4>, (const void*)gemm256_k<true, 4>, (const void*)gemm256_k<false, 3>,
                          (const void*)gemm256_k<true, 3>, (const void*)gemm256_k<false, 4, true>, (const void*)gemm256_k<true, 4, true>,
                          (const void*)gemm256_k<false, 3, true>, (const void*)gemm256_k<true, 3, true>,
                          (const void*)gemm256_k<false, 4, false, true>, (const void*)gemm256_k<false, 3, false, true>,
                          (const void*)gemm256w_k<8, true>, (const void*)gemm256w_k<6, true>, (const void*)gemm256w_k<8, false>, (const void*)gemm256w_k<6, false>};
    bool ok = true;
    for (int i = 0; i < 14; ++i) ok = ok && hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, i < 10 ? LDS_BYTES2 : W4_LDS_BYTES) == hipSuccess;
    attr_ok = ok;
  });
  if (!attr_ok) return VM_ERR_LAUNCH;
  const dim3 grid(p.tiles_m * p.tiles_n), block(512);
  hipStream_t st = (hipStream_t)stream;
  // the launch needs the LoRA extension's scale / dropout mask on its accumulators (the four-wave form has an instantiation without that code)
  const bool w4_scale = p.K2 > 0 && (p.drop_p > 0.f || p.alpha2 != 1.f);
#define VM_G256_LAUNCH(O, M_, F_) hipLaunchKernelGGL((gemm256_k<O, M_, F_>), grid, block, LDS_BYTES2, st, p)
  if (f8 == 2) {
    if (out_f32) return VM_ERR_UNSUPPORTED;
    if (tile_rows == 256) hipLaunchKernelGGL((gemm256_k<false, 4, false, true>), grid, block, LDS_BYTES2, st, p);
    else hipLaunchKernelGGL((gemm256_k<false, 3, false, true>), grid, block, LDS_BYTES2, st, p);
  } else if (f8) {
    if (tile_rows == 256) { if (out_f32) VM_G256_LAUNCH(true, 4, true); else VM_G256_LAUNCH(false, 4, true); }
    else { if (out_f32) VM_G256_LAUNCH(true, 3, true); else VM_G256_LAUNCH(false, 3, true); }
  } else if ((w4_mode() == 1 || (w4_mode() >= 2 && tile_rows == 192) || (w4_mode() == 3 && !w4_scale)) && p.K >= 192 && !out_f32 && p.act == VM_ACT_NONE) {          // (>= 3 main K-tiles: the weight is staged three deep)
    // the four-wave form (bf16 output, no fused activation) is persistent: one workgroup per CU walks the tile list (gemm256w.hpp);
    // 128 KiB of stages + 32 KiB of output slabs
    static const int cus = [] { int dev = 0, n = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256; return n; }();
    const int tiles = p.tiles_m * p.tiles_n;
    const dim3 grid4(tiles < cus ? tiles : cus), block4(256);
    if (tile_rows == 256) { if (w4_scale) hipLaunchKernelGGL((gemm256w_k<8, true>), grid4, block4, W4_LDS_BYTES, st, p); else hipLaunchKernelGGL((gemm256w_k<8, false>), grid4, block4, W4_LDS_BYTES, st, p); }
    else { if (w4_scale) hipLaunchKernelGGL((gemm256w_k<6, true>), grid4, block4, W4_LDS_BYTES, st, p); else hipLaunchKernelGGL((gemm256w_k<6, false>), grid4, block4, W4_LDS_BYTES, st, p); }
  } else {
    if (tile_rows == 256) { if (out_f32) VM_G256_LAUNCH(true, 4, false); else VM_G256_LAUNCH(false, 4, false); }
    else { if (out_f32) VM_G256_LAUNCH(true, 3, false); else VM_G256_LAUNCH(false, 3, false); }
  }
#undef VM_G256_LAUNCH
  return VM_OK;
}

// Stream-K launch (gemm_launch chooses it by cost, gemm.hip: sk_plan): `workers` workgroups, one per CU. The workspace holds
// `workers` fp32 accumulator slabs, then `workers` + 1 flag words (the last one: the owners' give-up mark, checked by the tests).
extern "C" int64_t vm_gemm256sk_workspace_(int workers) { return (int64_t)workers * (8 * 32 * 1024) + (int64_t)(workers + 1) * 4; }
extern "C" int vm_gemm256sk_launch_(const void* params, int out_f32, int segmented, int tile_rows, int f8, int workers, void* workspace,
                                    unsigned epoch, void* stream) {
  GemmParams p = *(const GemmParams*)params;
  if ((tile_rows != 256 && tile_rows != 192) || workers < 8 || !workspace) return VM_ERR_BAD_ARG;
  p.tiles_m = (p.M + tile_rows - 1) / tile_rows + (segmented ? 1 : 0);      // upper bound; the kernel counts the real tile rows
  p.tiles_n = (p.N + 255) / 256;
  p.sk_slabs = (float*)workspace;
  p.sk_flags = (unsigned*)((char*)workspace + (int64_t)workers * (8 * 32 * 1024));
  p.sk_epoch = epoch;
  p.sk_workers = workers;
  static std::once_flag attr_once;
  static bool attr_ok = false;
  std::call_once(attr_once, [] {
    const void* fns[8] = {(const void*)gemm256sk_k<false, 4>, (const void*)gemm256sk_k<true, 4>, (const void*)gemm256sk_k<false, 3>,
                         (const void*)gemm256sk_k<true, 3>, (const void*)gemm256sk_k<false, 4, true>, (const void*)gemm256sk_k<true, 4, true>,
                         (const void*)gemm256sk_k<false, 3, true>, (const void*)gemm256sk_k<true, 3, true>};
    bool ok = true;
    for (const void* f : fns) ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES2) == hipSuccess;
    attr_ok = ok;
  });
  if (!attr_ok) return VM_ERR_LAUNCH;
  const dim3 grid(workers), block(512);
  hipStream_t st = (hipStream_t)stream;
#define VM_SK_LAUNCH(O, M_, F_) hipLaunchKernelGGL((gemm256sk_k<O, M_, F_>), grid, block, LDS_BYTES2, st, p)
  if (f8) {
    if (tile_rows == 256) { if (out_f32) VM_SK_LAUNCH(true, 4, true); else VM_SK_LAUNCH(false, 4, true); }
    else { if (out_f32) VM_SK_LAUNCH(true, 3, true); else VM_SK_LAUNCH(false, 3, true); }
  } else {
    if (tile_rows == 256) { if (out_f32) VM_SK_LAUNCH(true, 4, false); else VM_SK_LAUNCH(false, 4, false); }
    else { if (out_f32) VM_SK_LAUNCH(true, 3, false); else VM_SK_LAUNCH(false, 3, false); }
  }
#undef VM_SK_LAUNCH
  return VM_OK;
}
