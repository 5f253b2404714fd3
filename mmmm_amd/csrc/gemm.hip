// bf16 MFMA GEMM for gfx950:  C[M,N] = act(A[M,K]·B[N,K]^T + alpha2·A2[M,K2]·B2[N,K2]^T + bias) + residual
//
// Replaces every nn.Linear / peft lora.Linear of the VividMed step (op sites listed in
// include/vividmed_hip.h) including the token-type gated 2-expert form of CogVLM's visual expert
// (reference modeling_cogvlm.py:87-98, 243-245, 277-279) as a 2-segment grouped GEMM.
//
// v1 structure: 128x128x64 tile, 4 waves (2x2), each wave 64x64 = 4x4
// v_mfma_f32_16x16x32_bf16 tiles; operands staged HBM -> LDS with buffer_load ... lds (16 B / lane,
// bounds-checked so ragged M/N need no branches); LDS image is lane-linear with the XOR swizzle on
// the *source* chunk (guide rule 21); double-buffered, one barrier per K-step.
// The weight is the MFMA "A" operand and the activation the "B" operand, so a lane's 4 accumulator
// registers are 4 consecutive output columns n (8-byte / 16-byte stores).
#include "vm_common.hpp"
#include "gemm_common.hpp"
#include <vector>
#include <mutex>
#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 128;
constexpr int TILE_BYTES = BM * 128;          // 16 KiB per operand tile (128 rows x 128 B)
constexpr int STAGE_BYTES = 2 * TILE_BYTES;   // A + B
constexpr int LDS_BYTES = 2 * STAGE_BYTES;    // double buffer = 64 KiB

// Issue the LDS-DMA loads of one 128x64 bf16 operand tile. `rsrc` covers the tile's valid rows
// (rows past the end read as zero), `ld_bytes` is the row pitch, `koff` the byte offset of the K-tile.
// `rows_valid`: 8-row pieces that lie entirely past it are NOT issued (their LDS rows keep whatever they held: the products of those rows
// are never stored). A launch of a few rows — the tails launch of the two-launch plan, the mask decoder's token-count GEMMs — walks K one
// tile at a time, and what a K-tile costs such a workgroup is mostly the issue of its DMA pieces (~60-130 cycles each, zero-fill or not).
template <int ROWS = 128>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, int ld_bytes, int koff,
                                           char* lds_tile, int wave, int lane, int rows_valid = 1 << 30) {
  constexpr int PW = ROWS / 32;   // wave-instructions (8 rows each) per wave
  const int r8 = lane >> 3, slot = lane & 7;
  const int chunk = slot ^ r8;  // source chunk that lands in LDS slot `slot` of row (.. + r8)
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    if ((wave * PW + i) * 8 >= rows_valid) continue;          // (wave-uniform)
    const int row = (wave * PW + i) * 8 + r8;
    const int voff = row * ld_bytes + chunk * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds_tile + (wave * PW + i) * 1024), 16, voff, koff, 0, 0);
  }
}

// Split an fp32 fragment (8 consecutive k of one row, as two f32x4) into NS bf16 terms x = t0 + t1 (+ t2) + O(2^-8NS |x|):
// t0 = bf16(x) (round to nearest even), t1 = bf16(x - t0), ... Each subtraction is exact in fp32.
template <int NS>
__device__ __forceinline__ void split_f32x8(const f32x4_t& lo4, const f32x4_t& hi4, bf16x8_t (&t)[NS]) {
  float x[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    bf16x8_t h;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (__bf16)x[e];
    t[s] = h;
    if (s + 1 < NS) {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] -= (float)h[e];
    }
  }
}

// ESZ = 2: bf16 operands, v_mfma_f32_16x16x32_bf16, K-tile 64.
// ESZ = 4: f32 operands, K-tile 32, same 128-byte LDS rows. NS = 0: v_mfma_f32_16x16x4_f32 (exact f32 fma chain, 1/16 of the bf16
//   MFMA rate). NS = 2 / 3: "split-bf16" — every fp32 fragment is split IN REGISTERS into NS bf16 terms and the product is
//   summed from the NS(NS+1)/2 leading cross terms on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: NS = 2 -> a0b0 + a0b1 + a1b0
//   (3 MFMAs per K = 32 where the exact form needs 8 at 1/16 rate: 5.3x the matrix rate; products carry 16 mantissa bits, relative
//   error ~2^-17 per term, sign-random, so a K-long dot product is good to ~1e-5 .. 1e-6 of its magnitude); NS = 3 -> 6 MFMAs,
//   24 mantissa bits = fp32 products (2.7x). The fp32 fragment layout (lane (frow, fq) holds k = 8 fq .. 8 fq + 7) IS the bf16
//   16x16x32 operand layout, so the split needs no data movement; its ~2.5 VALU instructions per element overlap with the MFMAs
//   of the co-resident waves.
// BMT = 128: 4 waves 2x2, wave tile 64 x 64.  BMT = 64 (fp32 output only): 4 waves 1x4, wave tile 64 rows x 32 columns —
// twice the workgroups for the M ~ 3k fp32 linears of SAM / iSAM, whose 128-row grids (150 tiles) leave 40 % of the CUs idle.
// BNT = output columns per tile: 128, or 32 for the TAILS launch of the two-launch plan (sched_plan kind 3) — a few dozen rows against all N
// columns: with 128-column tiles 64 workgroups each streamed 1 MB of weights through ONE CU's DMA path (~25 GB/s: 44 us at K = 4160, 110 us at
// 11 072); 32-column tiles put the same bytes on every CU.
template <int ESZ, bool OUT_F32, int BMT = 128, int NS = 0, int BNT = BN>
__global__ __launch_bounds__(256, 2) void gemm_nt_k(const GemmParams p) {
  static_assert(BMT == 128 || (BMT == 64 && OUT_F32), "the 64-row tile has the direct fp32 epilogue only");
  static_assert(NS == 0 || ESZ == 4, "the split applies to fp32 operands");
  static_assert(BNT == BN || (BNT == 32 && BMT == 128 && ESZ == 2 && !OUT_F32), "the 32-column tile exists for the bf16 tails launch");
  constexpr int B_TILE = BNT * 128;                      // weight tile bytes
  constexpr int BKE = 128 / ESZ;  // K elements per tile
  constexpr int NI = BMT == 128 ? BNT / 32 : 2;          // 16-wide n sub-tiles per wave
  constexpr int A_TILE = BMT * 128;                      // activation tile bytes
  constexpr int STAGE = A_TILE + B_TILE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = BMT == 128 ? wave >> 1 : 0, wn = BMT == 128 ? wave & 1 : wave;

  int tm, tn;
  gemm_tile_id(p, tm, tn);
  int row0, nrows, seg;
  gemm_tile_rows<BMT>(p, tm, row0, nrows, seg);
  if (nrows <= 0) return;
  const int n0 = tn * BNT;
  const int ncols = min(BNT, p.N - n0);

  const char* Bw = seg ? p.B1 : p.B0;
  const char* B2w = seg ? p.B2_1 : p.B2_0;

  // ---- buffer resources (wave-uniform): OOB rows read as zero
  const int lda_b = (int)p.lda * ESZ, ldb_b = (int)p.ldb * ESZ;
  const int kbeg = p.ksplit > 1 ? (int)blockIdx.y * p.kchunk : 0;              // split-K: this workgroup's K range
  const int kloc = p.ksplit > 1 ? min(p.kchunk, p.K - kbeg) : p.K;
  const int kskip = kbeg * ESZ;
  __amdgpu_buffer_rsrc_t rA = make_rsrc(p.A, (int64_t)row0 * lda_b + kskip, VM_DBG(p, 1) ? 0 : nrows * lda_b - kskip);
  __amdgpu_buffer_rsrc_t rB = make_rsrc(Bw, (int64_t)n0 * ldb_b + kskip, VM_DBG(p, 1) ? 0 : ncols * ldb_b - kskip);

  const int kt_ext = p.K2 / BKE;
  const int kt_main = kloc / BKE;
  const int kt_total = kt_ext + kt_main;

  __amdgpu_buffer_rsrc_t rA2 = rA, rB2 = rB;
  int lda2_b = 0, ldb2_b = 0;
  if (kt_ext > 0) {
    lda2_b = (int)p.lda2 * ESZ; ldb2_b = (int)p.ldb2 * ESZ;
    rA2 = make_rsrc(p.A2, (int64_t)row0 * lda2_b, nrows * lda2_b);
    rB2 = make_rsrc(B2w, (int64_t)n0 * ldb2_b, ncols * ldb2_b);
  }

  f32x4_t acc[NI][4];  // [n-subtile i][m-subtile j]
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // per-lane fragment read offsets (bytes) inside a tile.
  // bf16: lane (frow, fq) holds k = 32*ks + 8*fq + 0..7  -> source chunk 4*ks + fq.
  // f32 : lane (frow, fq) holds k = 8*fq + 0..7 (two 16-B chunks 2*fq, 2*fq+1) and feeds MFMA k-step s
  //       with element s; A and B use the same k permutation, so the sum over k is complete.
  const int frow = lane & 15, fq = lane >> 4;
  int off_k0, off_k1;
  if (ESZ == 2) {
    const int slot_k0 = fq ^ (frow & 7);
    off_k0 = frow * 128 + slot_k0 * 16;
    off_k1 = frow * 128 + (slot_k0 ^ 4) * 16;
  } else {
    off_k0 = frow * 128 + ((2 * fq) ^ (frow & 7)) * 16;
    off_k1 = frow * 128 + ((2 * fq + 1) ^ (frow & 7)) * 16;
  }

  // one K-step: issue the LDS-DMA of tile t+1 (HN), MFMA over tile t, wait for the DMA, barrier.
  // The K loop is peeled on HN / on the extension boundary so the steady state has no data-dependent branches
  // between the MFMA clusters (scalar branches cost tens of cycles each on this chip).
  auto kstep = [&](int t, auto hn_tag, auto ext_next_tag) {
    constexpr bool HN = decltype(hn_tag)::value;
    constexpr bool EXT_NEXT = decltype(ext_next_tag)::value;      // tile t+1 is an extension tile
    const int buf = t & 1;
    if (HN) {
      char* sa = smem + (buf ^ 1) * STAGE;
      char* sb = sa + A_TILE;
      if (EXT_NEXT) {
        stage_tile<BMT>(rA2, lda2_b, (t + 1) * 128, sa, wave, lane, nrows);
        stage_tile<BNT>(rB2, ldb2_b, (t + 1) * 128, sb, wave, lane);
      } else {
        const int koff = (t + 1 - kt_ext) * 128;
        stage_tile<BMT>(rA, lda_b, koff, sa, wave, lane, nrows);
        stage_tile<BNT>(rB, ldb_b, koff, sb, wave, lane);
      }
    }
    const char* sa = smem + buf * STAGE + wm * (64 * 128);
    const char* sb = smem + buf * STAGE + A_TILE + wn * (NI * 16 * 128);
    if (ESZ == 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int off = ks ? off_k1 : off_k0;
        bf16x8_t xa[4], wb[NI];
#pragma unroll
        for (int j = 0; j < 4; ++j) xa[j] = *reinterpret_cast<const bf16x8_t*>(sa + j * 2048 + off);
#pragma unroll
        for (int i = 0; i < NI; ++i) wb[i] = *reinterpret_cast<const bf16x8_t*>(sb + i * 2048 + off);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[i], xa[j], acc[i][j], 0, 0, 0);
      }
    } else {
      f32x4_t xa[4][2], wb[NI][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xa[j][0] = *reinterpret_cast<const f32x4_t*>(sa + j * 2048 + off_k0);
        xa[j][1] = *reinterpret_cast<const f32x4_t*>(sa + j * 2048 + off_k1);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        wb[i][0] = *reinterpret_cast<const f32x4_t*>(sb + i * 2048 + off_k0);
        wb[i][1] = *reinterpret_cast<const f32x4_t*>(sb + i * 2048 + off_k1);
      }
      if constexpr (NS == 0) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
          for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[i][s >> 2][s & 3], xa[j][s >> 2][s & 3], acc[i][j], 0, 0, 0);
      } else {
        constexpr int NSS = NS > 0 ? NS : 1;
        bf16x8_t xs[4][NSS], ws[NI][NSS];
#pragma unroll
        for (int j = 0; j < 4; ++j) split_f32x8<NSS>(xa[j][0], xa[j][1], xs[j]);
#pragma unroll
        for (int i = 0; i < NI; ++i) split_f32x8<NSS>(wb[i][0], wb[i][1], ws[i]);
        // cross terms a_s b_t with s + t < NS, smallest first
#pragma unroll
        for (int d = NSS - 1; d >= 0; --d)
#pragma unroll
          for (int sw = 0; sw <= d; ++sw)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws[i][sw], xs[j][d - sw], acc[i][j], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // next tile landed and everyone is done reading `buf`
  };
  auto ext_scale = [&]() {
    // end of the LoRA extension: scale, and (dgrad) apply the inverted-dropout mask of the forward's LoRA input
    if (p.drop_p > 0.f) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          gemm_ext_scale4<true>(p, row0 + wm * 64 + j * 16 + frow, n0 + wn * (NI * 16) + i * 16 + fq * 4, acc[i][j]);
    } else if (p.alpha2 != 1.f) {      // (rsLoRA with r = 64, alpha = 8: the scale is exactly 1 — nothing to do)
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] *= p.alpha2;
    }
  };
  const std::true_type T_{};
  const std::false_type F_{};

  if (kt_ext > 0) {
    stage_tile<BMT>(rA2, lda2_b, 0, smem, wave, lane, nrows);
    stage_tile<BNT>(rB2, ldb2_b, 0, smem + A_TILE, wave, lane);
  } else {
    stage_tile<BMT>(rA, lda_b, 0, smem, wave, lane, nrows);
    stage_tile<BNT>(rB, ldb_b, 0, smem + A_TILE, wave, lane);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // tile 0 landed

  int t = 0;
  for (; t + 1 < kt_ext; ++t) kstep(t, T_, T_);               // extension tiles followed by an extension tile
  if (kt_ext > 0) {                                           // last extension tile
    if (kt_main > 0) kstep(t, T_, F_); else kstep(t, F_, F_);
    ext_scale();
    ++t;
  }
  for (; t + 1 < kt_total; ++t) kstep(t, T_, F_);             // steady state
  if (t < kt_total) kstep(t, F_, F_);                         // last main tile

  // ---- epilogue
  if VM_DBG(p, 32) { if (acc[0][0][0] == 123.456f) ((float*)p.C)[0] = 1.f; return; }   // timing experiment: no C traffic
  const void* bias = seg ? p.bias1 : p.bias0;
  if (OUT_F32) {
    const bool splitk = p.ksplit > 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ml = wm * 64 + j * 16 + frow;
      if (ml >= nrows) continue;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int nl = wn * (NI * 16) + i * 16 + fq * 4;
        if (nl >= ncols) continue;
        if (!splitk) { gemm_store4<true>(p, bias, row0 + ml, n0 + nl, ncols - nl, acc[i][j]); continue; }
        // split-K partial: atomically accumulated; bias / residual enter once, with the first K range
        float* cp = (float*)p.C + (int64_t)(row0 + ml) * p.ldc + n0 + nl;
        for (int r = 0; r < 4 && r < ncols - nl; ++r) {
          float x = acc[i][j][r];
          if (blockIdx.y == 0) {
            if (bias) x += ((const float*)bias)[n0 + nl + r];
            if (p.residual) x += ((const float*)p.residual)[(int64_t)(row0 + ml) * p.ldr + n0 + nl + r];
          }
          atomicAdd(cp + r, x);
        }
      }
    }
  } else {
    // all waves are past the last K-step barrier: the staging LDS is free. 64 x 64 slab per wave.
    typedef EpiSlab<64, NI * 16> Slab;
    char* slab = smem + wave * Slab::BYTES;
    const bool plain = bias == nullptr && p.act == VM_ACT_NONE;
    const bool fast_bias = bias != nullptr && p.act == VM_ACT_NONE && ncols - wn * (NI * 16) >= NI * 16 && ((uintptr_t)bias & 7) == 0;
    if (fast_bias) {
      f32x4_t bv[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) bv[i] = epi_bias4(bias, n0 + wn * (NI * 16) + i * 16 + fq * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < NI; ++i)
          epi_put4<2>(slab, Slab::PITCH, j * 16 + frow, i * 16 + fq * 4, p, bias, 0, 4, acc[i][j], bv[i]);
    } else
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int nl = wn * (NI * 16) + i * 16 + fq * 4;
        if (plain) epi_put4<0>(slab, Slab::PITCH, j * 16 + frow, i * 16 + fq * 4, p, bias, n0 + nl, ncols - nl, acc[i][j]);
        else epi_put4<1>(slab, Slab::PITCH, j * 16 + frow, i * 16 + fq * 4, p, bias, n0 + nl, ncols - nl, acc[i][j]);
      }
    // same-wave LDS round trip: the compiler orders the ds_reads behind the ds_writes (lgkmcnt)
    epi_flush<64, NI * 16>(slab, p, row0 + wm * 64, n0 + wn * (NI * 16), nrows - wm * 64, ncols - wn * (NI * 16), lane);
  }
}


// ------------------------------------------------------------------ fp32 operands, split ONCE per element while staging
// gemm_nt_k<4, ., ., NS> stages raw fp32 tiles by LDS-DMA and every wave splits the fragments it reads: the activation rows are
// shared by the 4 (64-row tile) or 2 (128-row tile) waves of a row, so the split of one element runs up to 4 times and the loop is
// VALU-bound (64-row tile, NS = 2: ~140 vector instructions against 24 MFMAs per K-tile). Here a K-tile goes global -> registers ->
// split_f32x8 -> NS bf16 planes in LDS (one split per element: 3 fragments per thread), and the waves read bf16 operand fragments.
// Same terms, same products, same order as the in-register form: bit-identical results. No LoRA extension (K2 = 0: the fp32 islands
// have none). Plane image: 16-row blocks of 1 KiB, row r of a block at 64 r, its 16-byte chunk k (8 consecutive K) at slot
// (k + 2 (r >> 2)) & 3 — the four 16-lane groups of a ds_read_b128 each touch 16 distinct slots of the 256-byte bank row.
template <int BMT, int NS>
__global__ __launch_bounds__(256, 2) void gemm_nt_f32p_k(const GemmParams p) {
  constexpr int NI = BMT == 128 ? 4 : 2;
  constexpr int UA = BMT / 64;                           // staging units (row, 8 K) per thread: activation; the weight tile has 2
  constexpr int A_PLANE = BMT * 64, B_PLANE = BN * 64;
  constexpr int STAGE = NS * (A_PLANE + B_PLANE);
  extern __shared__ __attribute__((aligned(16))) char smem[];          // 2 * STAGE
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = BMT == 128 ? wave >> 1 : 0, wn = BMT == 128 ? wave & 1 : wave;
  int tm, tn;
  gemm_tile_id(p, tm, tn);
  int row0, nrows, seg;
  gemm_tile_rows<BMT>(p, tm, row0, nrows, seg);
  if (nrows <= 0) return;
  const int n0 = tn * BN;
  const int ncols = min(BN, p.N - n0);
  const char* Bw = seg ? p.B1 : p.B0;
  const int lda_b = (int)p.lda * 4, ldb_b = (int)p.ldb * 4;
  const int kbeg = p.ksplit > 1 ? (int)blockIdx.y * p.kchunk : 0;
  const int kloc = p.ksplit > 1 ? min(p.kchunk, p.K - kbeg) : p.K;
  const int kskip = kbeg * 4;
  const __amdgpu_buffer_rsrc_t rA = make_rsrc(p.A, (int64_t)row0 * lda_b + kskip, nrows * lda_b - kskip);
  const __amdgpu_buffer_rsrc_t rB = make_rsrc(Bw, (int64_t)n0 * ldb_b + kskip, ncols * ldb_b - kskip);
  const int kt_total = kloc / 32;

  f32x4_t acc[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int urow = tid >> 2, uk = tid & 3;
  auto plane_off = [](int row, int k) { return (row >> 4) * 1024 + (row & 15) * 64 + (((k + 2 * ((row & 15) >> 2)) & 3) << 4); };
  // (Two K-tiles of loads in flight were tried — the memory latency is ~8 K-tiles of MFMA work — and changed nothing: 61 vs 62 us on
  // [3136 x 3072 x 768]. What bounds these launches is operand traffic: 64 x 128 tiles of fp32 operands are 21 flop per byte, the
  // 1 176 workgroups of that shape pull 693 MB through the L2s, 11 TB/s at the measured time.)
  f32x4_t ga[UA][2], gb[2][2];
  auto load = [&](int t) {
    const int koff = t * 128;
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const int voff = (urow + 64 * u) * lda_b + uk * 32;
      ga[u][0] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rA, voff, koff, 0));
      ga[u][1] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rA, voff + 16, koff, 0));
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int voff = (urow + 64 * u) * ldb_b + uk * 32;
      gb[u][0] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rB, voff, koff, 0));
      gb[u][1] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rB, voff + 16, koff, 0));
    }
  };
  auto split_write = [&](int buf) {
    char* sa = smem + buf * STAGE;
    char* sb = sa + NS * A_PLANE;
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      bf16x8_t t[NS];
      split_f32x8<NS>(ga[u][0], ga[u][1], t);
#pragma unroll
      for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x8_t*>(sa + s * A_PLANE + plane_off(urow + 64 * u, uk)) = t[s];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      bf16x8_t t[NS];
      split_f32x8<NS>(gb[u][0], gb[u][1], t);
#pragma unroll
      for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x8_t*>(sb + s * B_PLANE + plane_off(urow + 64 * u, uk)) = t[s];
    }
  };
  const int frow = lane & 15, fq = lane >> 4;
  const int lo = frow * 64 + (((fq + 2 * (frow >> 2)) & 3) << 4);
  auto compute = [&](int buf) {
    const char* sa = smem + buf * STAGE + wm * (4 * 1024) + lo;
    const char* sb = smem + buf * STAGE + NS * A_PLANE + wn * (NI * 1024) + lo;
    bf16x8_t xs[4][NS], ws[NI][NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int j = 0; j < 4; ++j) xs[j][s] = *reinterpret_cast<const bf16x8_t*>(sa + s * A_PLANE + j * 1024);
#pragma unroll
      for (int i = 0; i < NI; ++i) ws[i][s] = *reinterpret_cast<const bf16x8_t*>(sb + s * B_PLANE + i * 1024);
    }
    // cross terms a_s b_t with s + t < NS, smallest first (the order of gemm_nt_k)
#pragma unroll
    for (int d = NS - 1; d >= 0; --d)
#pragma unroll
      for (int sw = 0; sw <= d; ++sw)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws[i][sw], xs[j][d - sw], acc[i][j], 0, 0, 0);
  };
  load(0);
  split_write(0);
  __syncthreads();
  for (int t = 0; t < kt_total; ++t) {
    const int buf = t & 1;
    const bool more = t + 1 < kt_total;
    if (more) load(t + 1);                                    // global loads in flight under this K-tile's MFMAs
    compute(buf);
    if (more) split_write(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue (the fp32 branch of gemm_nt_k)
  const void* bias = seg ? p.bias1 : p.bias0;
  const bool splitk = p.ksplit > 1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ml = wm * 64 + j * 16 + frow;
    if (ml >= nrows) continue;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int nl = wn * (NI * 16) + i * 16 + fq * 4;
      if (nl >= ncols) continue;
      if (!splitk) { gemm_store4<true>(p, bias, row0 + ml, n0 + nl, ncols - nl, acc[i][j]); continue; }
      float* cp = (float*)p.C + (int64_t)(row0 + ml) * p.ldc + n0 + nl;
      for (int r = 0; r < 4 && r < ncols - nl; ++r) {
        float x = acc[i][j][r];
        if (blockIdx.y == 0) {
          if (bias) x += ((const float*)bias)[n0 + nl + r];
          if (p.residual) x += ((const float*)p.residual)[(int64_t)(row0 + ml) * p.ldr + n0 + nl + r];
        }
        atomicAdd(cp + r, x);
      }
    }
  }
}

// ------------------------------------------------------------------ event profiling
struct ProfRec { hipEvent_t a, b; double flops; double bytes; };
struct ProfState {
  std::mutex mu;
  unsigned mask = 0;       // bit k: bracket launches of kind k
  int stride = 1;          // bracket every stride-th launch of a kind (uniform sample: 2 hipEventRecord per bracketed launch)
  unsigned long long seen[4] = {0, 0, 0, 0};
  std::vector<ProfRec> recs[4];
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
};
ProfState& prof() { static ProfState s; return s; }
double& prof_last_bytes() { static double b = 0; return b; }

}  // namespace

// shared with the other translation units
extern "C" int vm_prof_begin_(int kind, void* stream, void** tok) {
  ProfState& s = prof();
  if (!((s.mask >> kind) & 1u)) { *tok = nullptr; return 0; }
  std::lock_guard<std::mutex> lk(s.mu);
  // pseudo-random 1-in-stride sample (a fixed period would alias with the per-layer launch pattern)
  if (s.stride > 1 && ((unsigned)(s.seen[kind]++ * 2654435761ull >> 13) % (unsigned)s.stride)) { *tok = nullptr; return 0; }
  ProfRec r;
  if (!s.pool.empty()) { r.a = s.pool.back().first; r.b = s.pool.back().second; s.pool.pop_back(); }
  else { if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { *tok = nullptr; return 0; } }
  r.flops = 0; r.bytes = 0;
  (void)hipEventRecord(r.a, (hipStream_t)stream);
  s.recs[kind].push_back(r);
  *tok = (void*)(uintptr_t)(s.recs[kind].size());  // 1-based index
  return 0;
}
extern "C" int vm_prof_end2_(int kind, void* stream, void* tok, double flops, double bytes);
extern "C" int vm_prof_end_(int kind, void* stream, void* tok, double flops) { return vm_prof_end2_(kind, stream, tok, flops, 0.0); }
extern "C" int vm_prof_end2_(int kind, void* stream, void* tok, double flops, double bytes) {
  if (!tok) return 0;
  ProfState& s = prof();
  std::lock_guard<std::mutex> lk(s.mu);
  ProfRec& r = s.recs[kind][(size_t)(uintptr_t)tok - 1];
  r.flops = flops; r.bytes = bytes;
  (void)hipEventRecord(r.b, (hipStream_t)stream);
  return 0;
}

extern "C" {

int vm_version(void) { return 600; }      /* 600: round 6 (vm_attn_f32_args.causal / row_of_pos, head width 112); 510: VM_TN_GROUP_MAX 24 -> 32 (vm_tn_skinny_group_bf16 takes up to 32 items); 500: round 5 (vm_gemm_args.workspace / workspace_bytes, vm_gemm_workspace_bytes); 400: round 4 (vm_attn_args.workspace / workspace_bytes); 300: round 3 (vm_gemm_args.b_nn / f32_split, vm_attn_f32_args.f32_split) */

int vm_device_arch(char* name_host, int len) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return VM_ERR_LAUNCH;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return VM_ERR_LAUNCH;
  int i = 0;
  for (; i < len - 1 && prop.gcnArchName[i]; ++i) name_host[i] = prop.gcnArchName[i];
  if (len > 0) name_host[i] = 0;
  return VM_OK;
}

int vm_prof_enable(int kind_mask) { prof().mask = (unsigned)kind_mask & 0xFu; return VM_OK; }

/* algorithmic operand + result bytes (A, B, extension operands read once, C written once) summed by the LAST
 * vm_prof_collect call (bf16 / fp32 GEMM kinds; 0 for the others) */
int vm_prof_last_bytes(double* bytes_host) { if (!bytes_host) return VM_ERR_BAD_ARG; *bytes_host = prof_last_bytes(); return VM_OK; }

int vm_prof_stride(int every) { if (every < 1) return VM_ERR_BAD_ARG; prof().stride = every; return VM_OK; }

int vm_prof_reset(void) {
  ProfState& s = prof();
  std::lock_guard<std::mutex> lk(s.mu);
  for (auto& v : s.recs) { for (auto& r : v) s.pool.push_back({r.a, r.b}); v.clear(); }
  return VM_OK;
}

int vm_prof_collect(int kind, double* total_ms_host, double* total_flops_host, int64_t* launches_host) {
  if (kind < 0 || kind > 3) return VM_ERR_BAD_ARG;
  ProfState& s = prof();
  std::lock_guard<std::mutex> lk(s.mu);
  double ms = 0, fl = 0, by = 0;
  for (auto& r : s.recs[kind]) {
    if (hipEventSynchronize(r.b) != hipSuccess) return VM_ERR_LAUNCH;
    float t = 0;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return VM_ERR_LAUNCH;
    ms += t; fl += r.flops; by += r.bytes;
  }
  prof_last_bytes() = by;
  if (total_ms_host) *total_ms_host = ms;
  if (total_flops_host) *total_flops_host = fl;
  if (launches_host) *launches_host = (int64_t)s.recs[kind].size();
  return VM_OK;
}

extern "C" int vm_gemm256_launch_(const void* params, int out_f32, int segmented, int tile_rows, int f8, void* stream);
extern "C" int vm_gemm256sk_launch_(const void* params, int out_f32, int segmented, int tile_rows, int f8, int workers, void* workspace,
                                    unsigned epoch, void* stream);
extern "C" int64_t vm_gemm256sk_workspace_(int workers);

// ------------------------------------------------------------------ work scheduler of the 256-column kernel
// One workgroup per CU: a launch of T tiles costs ceil(T / W) ROUNDS, however little the last one holds. Three ways to run a shape:
//   DP-256 / DP-192  one tile per workgroup (gemm256_k), 256- or 192-row tiles;
//   SK-256           stream-K (gemm256sk_k): W persistent workgroups, the T mod W (+ W) leftover tiles cut into W equal K ranges with
//                    an fp32 slab hand-off, the rest as whole-tile rounds — no partial round, 256-row tiles throughout.
// Model (us, measured on MI355X with tools/ubench/gemm_bench: DESIGN.md section 3): a K-tile of the 256-row body takes T_K per CU, of the
// 192-row body 0.78 T_K; every segment a workgroup runs pays a fixed C_SEG (descriptors + first K-tile's latency + output stores); a
// split tile adds C_FIX (slab out, slab in, flag). The 128-tile kernel (two workgroups per CU) is for shapes with a handful of tiles.
struct SkPlan { int kind; int rows; };     // kind 0: 128-tile kernel, 1: DP, 2: stream-K, 3: FULL 256-row tiles + a tails launch of 128 x 128 tiles
static int cu_count() {
  static int n = [] { int dev = 0; hipDeviceProp_t prop; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
                      return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256; }();
  return n;
}
// 0 never (default: measured, DESIGN.md section 3 "stream-K": with the operands switched off the stream-K launch of [4128 x 4096] x 4160 takes
// 112 us against 143 for two rounds of 192-row tiles, with them 203 against 168 — workgroups that share an operand panel no longer walk K in
// step, every XCD's L2 then fetches each panel once per WORKGROUP instead of once), 1 by the cost model below, 2 whenever legal (tests,
// tools/ubench/gemm_bench): vm_gemm_sched_mode_
static int& sk_mode() { static int mode = 0; return mode; }
// 1 (default): the scheduler may split a launch into full 256-row tiles + a tails launch (plan kind 3); vm_gemm_tails_mode_(0) turns it off (A/B, tests)
static int& tails_mode() { static int mode = 1; return mode; }
static bool tails_ok() { return tails_mode() != 0; }
extern "C" int vm_gemm_tails_mode_(int mode) { if (mode < 0 || mode > 4) return VM_ERR_BAD_ARG; tails_mode() = mode; return VM_OK; }      // 2: whenever legal (tests)
extern "C" int vm_gemm_sched_mode_(int mode) { if (mode < 0 || mode > 2) return VM_ERR_BAD_ARG; sk_mode() = mode; return VM_OK; }
static unsigned sk_next_epoch() { static std::atomic<unsigned> e{0}; unsigned v; do { v = ++e; } while (v == 0); return v; }

// vm_gemm_force_tile_(128|192|256) forces the one-tile-per-workgroup kernel with that tile, -192 removes the 192-row form from the choice,
// 0 gives the choice back to the cost model (tests / A-B measurements). `kt` = K-tiles per output tile (extension included), `tk` = us per
// K-tile of the 256-row body.
constexpr double SK_C_SEG = 7.5, SK_C_FIX = 9.0, SK_T_K_BF16 = 1.42, SK_T_K_F8 = 2.0;
static int& forced_tile() { static int t = 0; return t; }
extern "C" int vm_gemm_force_tile_(int tile) {
  if (tile != 0 && tile != 128 && tile != 192 && tile != 256 && tile != -192) return VM_ERR_BAD_ARG;
  forced_tile() = tile;
  return VM_OK;
}
static SkPlan sched_plan(int M, int N, int kt, double tk, bool segmented, bool sk_ok, bool tails = true) {
  const int forced = forced_tile();
  if (forced == 128) return {0, 0};
  if (forced == 256 || forced == 192) return {1, forced};
  if (kt < 2) return {0, 0};
  const int W = cu_count();
  const int64_t tn = (N + 255) / 256, sg = segmented ? 1 : 0;
  const int64_t t256 = ((M + 255) / 256 + sg) * tn, t192 = ((M + 191) / 192 + sg) * tn;
  const int64_t t128 = ((M + 127) / 128 + sg) * ((N + 127) / 128);
  const double work = kt * tk;
  const double c256 = (double)((t256 + W - 1) / W) * (work + SK_C_SEG);
  const double c192 = forced == -192 ? 1e30 : (double)((t192 + W - 1) / W) * (0.78 * work + SK_C_SEG);
  // 128 x 128 tiles, two workgroups per CU: a quarter of a 256-row tile's work at ~0.6 of its rate per workgroup pair
  const double c128 = (double)((t128 + 2 * W - 1) / (2 * W)) * (0.85 * work + 4.0);
  double csk = 1e30;
  if (sk_ok && sk_mode() > 0 && t256 * (int64_t)kt >= 4 * (int64_t)W && (t256 % W)) {
    const double share = (double)t256 / W;                                      // tiles' worth of K-tiles per worker
    const int parts = t256 >= W ? 2 : (int)((W + t256 - 1) / t256) + 1;         // workers a split tile is spread over (worst case)
    csk = share * work + (double)((t256 + W - 1) / W) * SK_C_SEG + SK_C_FIX + 3.0 * (parts - 2);
    if (sk_mode() == 2) csk = 0.0;
  }
  const double cdp = c192 < c256 ? c192 : c256;
  // FULL + TAILS (two launches): the 256-row kernel runs only the tiles that are full — a partial tile's workgroup leaves at once — and the
  // rows behind each segment's last full tile (< 256 per segment) go to the 128 x 128 kernel in a second, short launch. It pays where the
  // partial tiles are what opens a new round: the decoder's N = 4096 linears at 4128 / 4176 rows are 16 full tiles x 16 + 32 nearly empty ones
  // (two segments of 2064 rows = 8 x 256 + 16) — 288 tiles = two rounds of 192-row tiles at 1.56 tile-times, against ONE round of 256 full tiles
  // + ~0.2 for the tails. The host only knows M (the segment boundary is a device count): floor(M / 256) x tn bounds the full tiles from above.
  double cft = 1e30;
  if (tails && tails_ok() && forced == 0) {
    const int64_t tfull = (int64_t)(M / 256) * tn;
    if (tfull > 0) cft = (double)((tfull + W - 1) / W) * (work + SK_C_SEG) + (0.5 * kt + 8.0);        // tails: a 128 x 32 workgroup walks K at ~0.5 us per K-tile (measured: 37 us at K = 4160, 87 at 11 072, 27 at 1 856)
  }
  if (tails_mode() >= 2 && cft < 1e29) return {3, 256};
  if (cft < cdp && cft < c128 && cft < csk) return {3, 256};
  if (csk < cdp && csk < c128) return {2, 256};
  if (c128 < cdp) return {0, 0};
  return {1, c192 < c256 ? 192 : 256};
}

// fp32 GEMM arithmetic: 0 = exact f32 MFMA, 2 = split-bf16 with 3 products, 3 = split-bf16 with 6 products (default: fp32 products).
static int& f32_mode() {
  static int mode = 3;
  return mode;
}


// timing-experiment builds (-DVM_GEMM_DEBUG_BUILD): VM_GEMM_DEBUG=<bits> in the environment of the first call, or vm_gemm_debug_set_
#ifdef VM_GEMM_DEBUG_BUILD
static int& gemm_dbg() { static int dbg = [] { const char* e = getenv("VM_GEMM_DEBUG"); return e ? atoi(e) : 0; }(); return dbg; }
extern "C" int vm_gemm_debug_set_(int bits) { gemm_dbg() = bits; return VM_OK; }
#else
static int gemm_dbg() { return 0; }
#endif

static int gemm_launch(const vm_gemm_args* a, void* stream, int esz) {
  const int bke = 128 / esz, al = 16 / esz;
  if (!a || !a->A || !a->B || !a->C) return VM_ERR_BAD_ARG;
  if (a->M <= 0 || a->N <= 0) return VM_OK;
  if (a->K <= 0 || a->K % bke || (a->K2 % bke) || a->K2 < 0) return VM_ERR_BAD_ARG;
  if (a->lda % al || a->ldb % al) return VM_ERR_BAD_ARG;
  if (a->K2 > 0 && (!a->A2 || !a->B2 || a->lda2 % al || a->ldb2 % al)) return VM_ERR_BAD_ARG;
  if (a->out_dtype != VM_BF16 && a->out_dtype != VM_F32) return VM_ERR_BAD_ARG;
  if (esz == 4 && a->out_dtype != VM_F32) return VM_ERR_UNSUPPORTED;
  if (a->ldc % 4) return VM_ERR_BAD_ARG;
  if (a->drop_p > 0.f && (a->N % 4)) return VM_ERR_BAD_ARG;
  const bool segmented = a->counts_dev != nullptr || a->split >= 0;
  if (segmented && !a->B_1) return VM_ERR_BAD_ARG;
  // 32-bit buffer offsets inside one tile: 128 rows * pitch must fit
  if ((int64_t)BM * a->lda * esz + (int64_t)a->K * esz >= (1ll << 31)) return VM_ERR_UNSUPPORTED;
  if (!a->b_nn && (int64_t)BN * a->ldb * esz + (int64_t)a->K * esz >= (1ll << 31)) return VM_ERR_UNSUPPORTED;

  GemmParams p;
  p.A = (const char*)a->A; p.lda = a->lda;
  p.B0 = (const char*)a->B; p.B1 = (const char*)(a->B_1 ? a->B_1 : a->B); p.ldb = a->ldb;
  p.A2 = (const char*)a->A2; p.lda2 = a->lda2;
  p.B2_0 = (const char*)a->B2; p.B2_1 = (const char*)(a->B2_1 ? a->B2_1 : a->B2); p.ldb2 = a->ldb2;
  p.K2 = a->K2; p.alpha2 = a->alpha2;
  p.bias0 = a->bias; p.bias1 = a->bias_1 ? a->bias_1 : a->bias;
  p.residual = a->residual; p.ldr = a->ldr;
  p.C = a->C; p.ldc = a->ldc;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.counts_dev = a->counts_dev;
  p.split = segmented ? (a->counts_dev ? 0 : a->split) : -1;
  p.act = a->act;
  p.drop_p = a->drop_p; p.drop_seed = a->drop_seed;
  p.ksplit = 1; p.kchunk = a->K;
  p.row_scale = nullptr; p.col_scale0 = p.col_scale1 = nullptr;
  p.b_nn = 0;
  p.row_filter = 0; p.tail_base = 256; p.tiles_m_override = 0;
  if (a->ksplit > 1) {
    if (a->out_dtype != VM_F32 || a->act != VM_ACT_NONE || a->K2 != 0) return VM_ERR_BAD_ARG;
    const int kt = a->K / bke;
    const int per = (kt + a->ksplit - 1) / a->ksplit;
    p.kchunk = per * bke;
    p.ksplit = (kt + per - 1) / per;
  }
  // fp32 operands: 128-row or 64-row tiles by rounds x cost. A CU holds 2 workgroups of the 128-row tile (198-241 VGPRs) and 3 of
  // the 64-row tile (48 KiB of LDS each); a 64-row tile costs ~0.55 of a 128-row one (same weight tile, half the rows). The
  // heads' shapes sit right at the quantisation edges: [3136 x 3072] is 600 tiles of 128 rows = 2 rounds over 512 slots (the second
  // 17 % full) but 1176 tiles of 64 rows = 2 cheap rounds over 768 slots; [3136 x 768] fills 29 % of one round either way.
  bool bm64 = false;
  if (esz == 4 && p.ksplit <= 1 && a->M > 64) {
    const int64_t tn = (a->N + BN - 1) / BN, sg = segmented ? 1 : 0;
    const int64_t t128 = ((a->M + 127) / 128 + sg) * tn, t64 = ((a->M + 63) / 64 + sg) * tn;
    const int64_t r128 = (t128 + 511) / 512, r64 = (t64 + 767) / 768;
    bm64 = r64 * 55 < r128 * 100;
  }
  const int bm = bm64 ? 64 : BM;
  p.tiles_m = (a->M + bm - 1) / bm + (segmented ? 1 : 0);
  p.tiles_n = (a->N + BN - 1) / BN;
  p.dbg = gemm_dbg();
  const int grid = p.tiles_m * p.tiles_n;
  const int kind = esz == 2 ? VM_PROF_GEMM_BF16 : VM_PROF_GEMM_F32;
  if (a->f32_split < 0 || a->f32_split > 3) return VM_ERR_BAD_ARG;
  const int fmode = a->f32_split == 0 ? f32_mode() : (a->f32_split == 1 ? 0 : a->f32_split);

  // (every argument check comes BEFORE the profiling bracket opens: an early return must not leave an unmatched begin)
  if (a->b_nn && (esz != 2 || a->out_dtype != VM_BF16 || p.ksplit > 1 || a->N % 8 || (int64_t)a->K * a->ldb * 2 >= (1ll << 31))) return VM_ERR_UNSUPPORTED;
  void* tok = nullptr;
  vm_prof_begin_(kind, stream, &tok);
  const bool sk_ok = esz == 2 && !a->b_nn && a->workspace && a->workspace_bytes >= vm_gemm256sk_workspace_(cu_count());
  // (the tails launch is the bf16-output 128 x 32 kernel on an NT weight: other calls are planned WITHOUT plan 3, so that they get the best of
  // the remaining plans — 256-row, 192-row or 128 x 128 tiles — instead of a fixed fallback)
  const bool tails_legal = !a->b_nn && a->out_dtype == VM_BF16;
  SkPlan plan = (esz == 2 && p.ksplit <= 1) ? sched_plan(a->M, a->N, (a->K + a->K2) / 64, SK_T_K_BF16, segmented, sk_ok, tails_legal) : SkPlan{0, 0};
  int big = plan.kind ? plan.rows : 0;
  if (a->b_nn) {
    // weight given as [K, N] (contraction-major, e.g. W itself for dx = dy W): only the 256-column kernel has that operand path
    big = 192;                                                            // (the 256-row NN form spills 15 VGPRs)
    p.b_nn = 1;
  }
  if (big && plan.kind == 2 && !a->b_nn) {
    const int rc = vm_gemm256sk_launch_(&p, a->out_dtype == VM_F32, segmented ? 1 : 0, big, 0, cu_count(), a->workspace, sk_next_epoch(), stream);
    if (rc != VM_OK) { vm_prof_end2_(kind, stream, tok, 0.0, 0.0); return rc; }
  } else if (big && plan.kind == 3) {
    GemmParams pf = p;
    pf.row_filter = 1;                                          // launch 1: the full 256-row tiles, indexed compactly (tiles_m = an upper bound of their rows)
    pf.tiles_m_override = a->M / 256;
    const int rc = tails_mode() == 4 ? VM_OK : vm_gemm256_launch_(&pf, 0, segmented ? 1 : 0, 256, 0, stream);      // (modes 3 / 4: timing experiments — only the first / only the second launch)
    if (rc != VM_OK) { vm_prof_end2_(kind, stream, tok, 0.0, 0.0); return rc; }
    GemmParams pt = p;
    pt.row_filter = 2; pt.tail_base = 256;                      // launch 2: < 256 rows per segment in 128-row tiles (two per segment)
    pt.tiles_m = segmented ? 4 : 2;
    pt.tiles_n = (a->N + 31) / 32;                              // 32-column tiles: every CU streams a slice of the weights
    if (tails_mode() != 3) hipLaunchKernelGGL((gemm_nt_k<2, false, 128, 0, 32>), dim3(pt.tiles_m * pt.tiles_n), dim3(256), 2 * (128 * 128 + 32 * 128), (hipStream_t)stream, pt);
  } else if (big) {
    const int rc = vm_gemm256_launch_(&p, a->out_dtype == VM_F32, segmented ? 1 : 0, big, a->b_nn ? 2 : 0, stream);
    if (rc != VM_OK) { vm_prof_end2_(kind, stream, tok, 0.0, 0.0); return rc; }
  } else if (esz == 4 && a->K2 == 0 && fmode == 2 && bm64) {
    // operands split once per element while staging (gemm_nt_f32p_k); two products' planes fit the same LDS as the raw fp32 tiles.
    // (64-row tiles only: the 128-row form with two K-tiles of loads in flight needs 254 VGPRs + 20 spill slots)
    hipLaunchKernelGGL((gemm_nt_f32p_k<64, 2>), dim3(grid, 1), dim3(256), 2 * 2 * (64 + 128) * 64, (hipStream_t)stream, p);
  } else if (esz == 4 && a->K2 == 0 && fmode == 3 && bm64) {
    static std::once_flag once;
    static bool ok = false;
    std::call_once(once, [] { ok = hipFuncSetAttribute((const void*)gemm_nt_f32p_k<64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 3 * (64 + 128) * 64) == hipSuccess; });
    if (!ok) { vm_prof_end2_(kind, stream, tok, 0.0, 0.0); return VM_ERR_LAUNCH; }
    hipLaunchKernelGGL((gemm_nt_f32p_k<64, 3>), dim3(grid, 1), dim3(256), 2 * 3 * (64 + 128) * 64, (hipStream_t)stream, p);
  } else if (esz == 4 && bm64) {
    const int lds = 2 * (64 * 128 + TILE_BYTES);
    switch (fmode) {
      case 0: hipLaunchKernelGGL((gemm_nt_k<4, true, 64, 0>), dim3(grid, 1), dim3(256), lds, (hipStream_t)stream, p); break;
      case 2: hipLaunchKernelGGL((gemm_nt_k<4, true, 64, 2>), dim3(grid, 1), dim3(256), lds, (hipStream_t)stream, p); break;
      default: hipLaunchKernelGGL((gemm_nt_k<4, true, 64, 3>), dim3(grid, 1), dim3(256), lds, (hipStream_t)stream, p); break;
    }
  } else if (esz == 4) {
    switch (fmode) {
      case 0: hipLaunchKernelGGL((gemm_nt_k<4, true, 128, 0>), dim3(grid, p.ksplit), dim3(256), LDS_BYTES, (hipStream_t)stream, p); break;
      case 2: hipLaunchKernelGGL((gemm_nt_k<4, true, 128, 2>), dim3(grid, p.ksplit), dim3(256), LDS_BYTES, (hipStream_t)stream, p); break;
      default: hipLaunchKernelGGL((gemm_nt_k<4, true, 128, 3>), dim3(grid, p.ksplit), dim3(256), LDS_BYTES, (hipStream_t)stream, p); break;
    }
  }
  else if (a->out_dtype == VM_F32)
    hipLaunchKernelGGL((gemm_nt_k<2, true>), dim3(grid, p.ksplit), dim3(256), LDS_BYTES, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL((gemm_nt_k<2, false>), dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, p);
  vm_prof_end2_(kind, stream, tok, 2.0 * (double)a->M * (double)a->N * (double)(a->K + a->K2),
                ((double)a->M * (a->K + a->K2) + (double)a->N * (a->K + a->K2)) * esz + (double)a->M * a->N * (a->out_dtype == VM_F32 ? 4 : 2));
  VM_LAUNCH_CHECK();
  return VM_OK;
}

/* internal (tests, tools): what the scheduler would run for a bf16-output NT shape — kind 0 = 128 x 128 tiles, 1 = one 256-column tile per workgroup
 * (`rows` 256 or 192), 2 = stream-K (only with a workspace AND vm_gemm_sched_mode_ > 0), 3 = full 256-row tiles + a tails launch (vm_gemm_tails_mode_) */
int vm_gemm_plan_(int M, int N, int K, int K2, int segmented, int with_workspace, int* kind_host, int* rows_host) {
  const SkPlan pl = sched_plan(M, N, (K + K2) / 64, SK_T_K_BF16, segmented != 0, with_workspace != 0);
  if (kind_host) *kind_host = pl.kind;
  if (rows_host) *rows_host = pl.rows;
  return VM_OK;
}
int vm_gemm_workspace_bytes(int64_t* bytes_host) { if (!bytes_host) return VM_ERR_BAD_ARG; *bytes_host = vm_gemm256sk_workspace_(cu_count()); return VM_OK; }
int vm_gemm_bf16(const vm_gemm_args* a, void* stream) { return gemm_launch(a, stream, 2); }
int vm_gemm_f32(const vm_gemm_args* a, void* stream) { return gemm_launch(a, stream, 4); }
int vm_gemm_f32_mode_get_(void) { return f32_mode(); }      /* internal: the process default as the other translation units see it */
int vm_gemm_f32_mode(int mode) {
  if (mode != 0 && mode != 2 && mode != 3) return VM_ERR_BAD_ARG;
  f32_mode() = mode;
  return VM_OK;
}

/* fp8 (OCP e4m3) main product on v_mfma_f32_16x16x128_f8f6f4 with per-row / per-output-channel scales, LoRA extension in bf16:
 *   C[m][n] = act( sa[m] sb[n] sum_k A8[m][k] B8[n][k]  +  alpha2 mask sum_r A2[m][r] B2[n][r]  + bias ) + residual
 * The extension operands must arrive PRE-DIVIDED by the same scales (A2[m][:] / sa[m], B2[n][:] / sb[n]): the kernel accumulates the
 * extension first, then the raw fp8 products, and multiplies the sum by sa[m] sb[n] once. */
int vm_gemm_fp8(const vm_gemm_args* a, const float* row_scale, const float* col_scale, const float* col_scale_1, void* stream) {
  if (!a || !a->A || !a->B || !a->C || !row_scale || !col_scale) return VM_ERR_BAD_ARG;
  if (a->M <= 0 || a->N <= 0) return VM_OK;
  if (a->K <= 0 || a->K % 128 || a->K2 % 64 || a->K2 < 0) return VM_ERR_BAD_ARG;
  if (a->lda % 16 || a->ldb % 16) return VM_ERR_BAD_ARG;                      /* bytes = elements for fp8 */
  if (a->K2 > 0 && (!a->A2 || !a->B2 || a->lda2 % 8 || a->ldb2 % 8)) return VM_ERR_BAD_ARG;
  if (a->out_dtype != VM_BF16 && a->out_dtype != VM_F32) return VM_ERR_BAD_ARG;
  if (a->ldc % 4 || a->ksplit > 1) return VM_ERR_BAD_ARG;
  if (a->drop_p > 0.f && (a->N % 4)) return VM_ERR_BAD_ARG;
  const bool segmented = a->counts_dev != nullptr || a->split >= 0;
  if (segmented && (!a->B_1 || !col_scale_1)) return VM_ERR_BAD_ARG;
  if ((int64_t)256 * a->lda + a->K >= (1ll << 31) || (int64_t)256 * a->ldb + a->K >= (1ll << 31)) return VM_ERR_UNSUPPORTED;
  GemmParams p;
  p.A = (const char*)a->A; p.lda = a->lda / 2;                                /* the kernel turns element counts into bytes with x 2 */
  p.B0 = (const char*)a->B; p.B1 = (const char*)(a->B_1 ? a->B_1 : a->B); p.ldb = a->ldb / 2;
  p.A2 = (const char*)a->A2; p.lda2 = a->lda2;
  p.B2_0 = (const char*)a->B2; p.B2_1 = (const char*)(a->B2_1 ? a->B2_1 : a->B2); p.ldb2 = a->ldb2;
  p.K2 = a->K2; p.alpha2 = a->alpha2;
  p.bias0 = a->bias; p.bias1 = a->bias_1 ? a->bias_1 : a->bias;
  p.residual = a->residual; p.ldr = a->ldr;
  p.C = a->C; p.ldc = a->ldc;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.counts_dev = a->counts_dev;
  p.split = segmented ? (a->counts_dev ? 0 : a->split) : -1;
  p.act = a->act;
  p.drop_p = a->drop_p; p.drop_seed = a->drop_seed;
  p.ksplit = 1; p.kchunk = a->K;
  p.b_nn = 0;
  p.row_filter = 0; p.tail_base = 256; p.tiles_m_override = 0;
  p.row_scale = row_scale; p.col_scale0 = col_scale; p.col_scale1 = col_scale_1 ? col_scale_1 : col_scale;
  p.dbg = gemm_dbg();
  const bool sk_ok = a->workspace && a->workspace_bytes >= vm_gemm256sk_workspace_(cu_count());
  SkPlan plan = sched_plan(a->M, a->N, a->K / 128 + a->K2 / 64, SK_T_K_F8, segmented, sk_ok, false);      // (no fp8 128 x 128 kernel for a tails launch)
  if (!plan.kind) plan = SkPlan{1, 256};                                      /* the fp8 main loop exists in the 256-column kernel only */
  void* tok = nullptr;
  vm_prof_begin_(VM_PROF_GEMM_BF16, stream, &tok);
  const int rc = plan.kind == 2 ? vm_gemm256sk_launch_(&p, a->out_dtype == VM_F32, segmented ? 1 : 0, plan.rows, 1, cu_count(), a->workspace, sk_next_epoch(), stream)
                                : vm_gemm256_launch_(&p, a->out_dtype == VM_F32, segmented ? 1 : 0, plan.rows, 1, stream);
  if (rc != VM_OK) { vm_prof_end2_(VM_PROF_GEMM_BF16, stream, tok, 0.0, 0.0); return rc; }
  vm_prof_end2_(VM_PROF_GEMM_BF16, stream, tok, 2.0 * (double)a->M * (double)a->N * (double)(a->K + a->K2),
                ((double)a->M + (double)a->N) * (a->K + 2.0 * a->K2) + (double)a->M * a->N * (a->out_dtype == VM_F32 ? 4 : 2));
  VM_LAUNCH_CHECK();
  return VM_OK;
}

}  // extern "C"
