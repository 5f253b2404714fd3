// Shared pieces of the NT GEMM kernels (128x128 tile in gemm.hip, 256x256 tile in gemm256.hip).
#pragma once
#include "vm_common.hpp"

typedef __attribute__((address_space(3))) void* lds_ptr_t;
#ifndef VM_GROUP_M
#define VM_GROUP_M 8
#endif
constexpr int GROUP_M = VM_GROUP_M;

// Timing experiments on the GEMM kernels (operands without traffic, epilogue without stores, no XCD remap: the knock-out measurements
// of DESIGN.md section 5) are compiled in only with -DVM_GEMM_DEBUG_BUILD (then selected at run time by VM_GEMM_DEBUG=<bits>); the
// shipped library carries none of their branches.
#ifdef VM_GEMM_DEBUG_BUILD
#define VM_DBG(p, bit) (((p).dbg & (bit)) != 0)
#else
#define VM_DBG(p, bit) (false)
#endif

struct GemmParams {
  const char* A; int64_t lda;          // leading dimensions in ELEMENTS
  const char* B0; const char* B1; int64_t ldb;
  const char* A2; int64_t lda2;
  const char* B2_0; const char* B2_1; int64_t ldb2;
  int K2; float alpha2;
  const void* bias0; const void* bias1;
  const void* residual; int64_t ldr;
  void* C; int64_t ldc;
  int M, N, K;
  const int32_t* counts_dev;
  int split;
  int act;
  float drop_p; uint64_t drop_seed;
  int tiles_m, tiles_n;
  // fp8 main product (gemm256_k<.., F8 = true>): A / B hold e4m3 bytes, K counts fp8 elements; the epilogue multiplies the
  // accumulators by row_scale[m] * col_scale[n] (per-row activation scales, per-output-channel weight scales of each segment)
  const float* row_scale; const float* col_scale0; const float* col_scale1;
  int b_nn;             // gemm256_k<.., BNN>: the main B operand is stored [contraction][output column] (a weight as it sits in HBM, for dx = dy W)
  int ksplit, kchunk;   // split-K (fp32 atomics into a zeroed C): blockIdx.y owns K range [y*kchunk, (y+1)*kchunk)
  // stream-K (gemm256sk_k): fp32 accumulator slabs [sk_workers][8 waves][2 MI x 4 registers][64 lanes][4], one flag word per worker
  // (= sk_epoch once the worker's slab is complete; epochs grow monotonically per process, so nothing is ever re-zeroed)
  float* sk_slabs; unsigned* sk_flags; unsigned sk_epoch; int sk_workers;
  // row filter of the two-launch plan (gemm.hip: sched_plan kind 3): 0 every tile; 1 only the FULL tiles of each row segment (a partial
  // tile's workgroup leaves at once); 2 only the rows behind the last full `tail_base`-row tile of each segment — tile index tm = 2 * segment +
  // (0 | 1) of the 128-row kernel
  int row_filter, tail_base;
  int tiles_m_override;      // row_filter 1: tile rows of the grid (vm_gemm256_launch_ would size it for every tile)
  int dbg;   // timing-experiment builds only (-DVM_GEMM_DEBUG_BUILD): bit0 = zero-record descriptors (no operand traffic), bit1 = no XCD remap
};

// Buffer resource from provably wave-uniform words (avoids hipcc's waterfall loops, guide T20).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const char* base, int64_t byte_off, int bytes) {
  const uint64_t a = (uint64_t)(base + byte_off);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const int n = __builtin_amdgcn_readfirstlane(bytes);
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}


// XCD-aware bijective tile remap (blocks b and b+8 share an XCD/L2: give each XCD a contiguous chunk of tiles), then
// GROUP_M-grouped ordering so that concurrently running tiles share A/B panels.
__device__ __forceinline__ void gemm_tile_id(const GemmParams& p, int& tm, int& tn) {
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  if (!VM_DBG(p, 2)) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int per_group = GROUP_M * p.tiles_n;
  const int g = bid / per_group;
  const int gm0 = g * GROUP_M;
  const int gsz = min(GROUP_M, p.tiles_m - gm0);
  tm = gm0 + (bid % per_group) % gsz;
  tn = (bid % per_group) / gsz;
}

// rows of m-tile `tm` (device-side counts for the token-routed 2-segment form)
template <int BM_>
__device__ __forceinline__ void gemm_tile_rows(const GemmParams& p, int tm, int& row0, int& nrows, int& seg) {
  int M = p.M, split = p.split;
  if (p.counts_dev) {
    split = __builtin_amdgcn_readfirstlane(p.counts_dev[0]);
    M = min(p.M, __builtin_amdgcn_readfirstlane(p.counts_dev[1]));
  }
  seg = 0;
  if (p.row_filter == 2) {
    // the tails launch: tm = 2 * segment + k, k-th BM_-row tile behind the segment's last full tail_base-row tile
    if (split >= 0) split = min(split, M);
    seg = tm >> 1;
    const int s_begin = (seg && split >= 0) ? split : 0, s_end = (split >= 0 && !seg) ? split : M;
    if (seg && split < 0) { row0 = 0; nrows = 0; return; }
    row0 = s_begin + ((s_end - s_begin) / p.tail_base) * p.tail_base + (tm & 1) * BM_;
    nrows = min(BM_, s_end - row0);
    return;
  }
  if (p.row_filter == 1) {
    // the full-tiles launch: tm counts FULL tiles only (segment 0's, then segment 1's) — the grid holds no workgroup that would leave at once,
    // so the XCD-contiguous chunks of the tile order stay equally loaded (with the partial tiles in the grid, an XCD whose chunk held none of
    // them ran 36 tiles on 32 CUs: two rounds, 171 us where one round takes ~105)
    if (split < 0) { row0 = tm * BM_; nrows = (row0 + BM_ <= M) ? BM_ : 0; return; }
    split = min(split, M);
    const int f0 = split / BM_;
    if (tm < f0) { row0 = tm * BM_; nrows = BM_; }
    else { seg = 1; row0 = split + (tm - f0) * BM_; nrows = (row0 + BM_ <= M) ? BM_ : 0; }
    return;
  }
  if (split < 0) {
    row0 = tm * BM_; nrows = min(BM_, M - row0);
  } else {
    split = min(split, M);
    const int t0 = (split + BM_ - 1) / BM_;
    if (tm < t0) { row0 = tm * BM_; nrows = min(BM_, split - row0); }
    else { seg = 1; row0 = split + (tm - t0) * BM_; nrows = min(BM_, M - row0); }
  }
}

// LoRA-extension post-scale of a lane's 4 consecutive-n accumulator registers (+ dgrad dropout mask).
// DROP is a compile-time flag: the caller branches ONCE on p.drop_p around the whole accumulator sweep.
template <bool DROP>
__device__ __forceinline__ void gemm_ext_scale4(const GemmParams& p, int64_t m, int n, f32x4_t& a) {
  const float a2 = p.alpha2;
  if (!DROP) { a *= a2; return; }
  const float inv_keep = 1.0f / (1.0f - p.drop_p);
  const uint64_t hsh = vm_hash4(p.drop_seed, ((uint64_t)m * (uint64_t)p.N + (uint64_t)n) >> 2);
  const unsigned thr = vm_drop_threshold(p.drop_p);
#pragma unroll
  for (int r = 0; r < 4; ++r) a[r] *= vm_keep_bits(hsh, r, thr) ? a2 * inv_keep : 0.f;
}

// epilogue of one lane-owned group C[m][n .. n+3]: bias, activation, residual with torch's bf16 rounding points
template <bool OUT_F32>
__device__ __forceinline__ void gemm_store4(const GemmParams& p, const void* bias, int64_t m, int n, int ncols_left,
                                            const f32x4_t& acc) {
  float v[4] = {acc[0], acc[1], acc[2], acc[3]};
  const bool full = ncols_left >= 4;
  if (OUT_F32) {
    const float* bp = (const float*)bias;
    const float* rp = (const float*)p.residual;
    float* cp = (float*)p.C + m * p.ldc + n;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (!full && r >= ncols_left) break;
      float x = v[r];
      if (bp) x += bp[n + r];
      if (p.act == VM_ACT_GELU) x = gelu_erf(x);
      else if (p.act == VM_ACT_RELU) x = fmaxf(x, 0.f);
      if (rp) x += rp[m * p.ldr + n + r];
      v[r] = x;
    }
    if (full) *reinterpret_cast<f32x4_t*>(cp) = (f32x4_t){v[0], v[1], v[2], v[3]};
    else for (int r = 0; r < 4 && r < ncols_left; ++r) cp[r] = v[r];
  } else {
    const unsigned short* bp = (const unsigned short*)bias;
    const unsigned short* rp = (const unsigned short*)p.residual;
    unsigned short* cp = (unsigned short*)p.C + m * p.ldc + n;
    unsigned short o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (!full && r >= ncols_left) { o[r] = 0; continue; }
      float x = v[r];
      if (bp) x += bf2f(bp[n + r]);
      // torch rounds the linear's output to bf16 before the activation and before the residual add
      if (p.act == VM_ACT_GELU) x = gelu_erf(bf2f(f2bf(x)));
      else if (p.act == VM_ACT_RELU) x = fmaxf(x, 0.f);
      if (rp) x = bf2f(f2bf(x)) + bf2f(rp[m * p.ldr + n + r]);
      o[r] = f2bf(x);
    }
    if (full) *reinterpret_cast<u16x4_t*>(cp) = (u16x4_t){o[0], o[1], o[2], o[3]};
    else for (int r = 0; r < 4 && r < ncols_left; ++r) cp[r] = o[r];
  }
}

// Coalesced bf16 epilogue through LDS: every wave parks its RxC accumulator sub-tile (bias / activation applied, bf16)
// in a private LDS slab and writes it back as whole 16-byte chunks of contiguous rows (8 lanes x 16 B = one 128-B line
// per row), adding the residual with the same wide accesses. The direct form issues 8-byte stores scattered over 16 rows
// per instruction and was measured store-issue bound (65-110 us fixed per GEMM at M=3648, N=12288).
// slab: [ROWS][PITCH bytes], PITCH = COLS*2 + 16 keeps 16-B alignment and staggers banks.
template <int ROWS, int COLS>
struct EpiSlab {
  static constexpr int PITCH = COLS * 2 + 16;
  static constexpr int BYTES = ROWS * PITCH;
};

// bf16 bias of a lane's 4 consecutive columns as floats (interior column group, 8-byte aligned bias pointer)
__device__ __forceinline__ f32x4_t epi_bias4(const void* bias, int n) {
  const u16x4_t b = *reinterpret_cast<const u16x4_t*>((const unsigned short*)bias + n);
  return (f32x4_t){bf2f(b[0]), bf2f(b[1]), bf2f(b[2]), bf2f(b[3])};
}

// write one lane-owned group (row r, cols c..c+3 of the slab) after bias + activation.
// MODE 0: no bias, no activation (no per-element branches at all); MODE 2: bias pre-loaded by the caller into `bv`
// (one 8-byte load per column group, reused for every row tile), no activation; MODE 1: generic.
template <int MODE>
__device__ __forceinline__ void epi_put4(char* slab, int pitch, int r, int c, const GemmParams& p, const void* bias, int n,
                                         int ncols_left, const f32x4_t& acc, const f32x4_t bv = (f32x4_t){0.f, 0.f, 0.f, 0.f}) {
  u16x4_t o;
  if (MODE == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[e]);
  } else if (MODE == 2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[e] + bv[e]);
  } else {
    const unsigned short* bp = (const unsigned short*)bias;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = acc[e];
      if (bp && e < ncols_left) x += bf2f(bp[n + e]);
      if (p.act == VM_ACT_GELU) x = gelu_erf(bf2f(f2bf(x)));
      else if (p.act == VM_ACT_RELU) x = fmaxf(x, 0.f);
      o[e] = f2bf(x);
    }
  }
  *reinterpret_cast<u16x4_t*>(slab + r * pitch + c * 2) = o;
}

// flush ROWS x COLS bf16 from the slab to C[m0 + r][n0 + c], one wave, 16 B per lane; rows >= rows_valid and columns
// >= cols_valid are skipped; the residual (bf16, same layout as C) is added after rounding, as torch does.
// Interior tiles with aligned leading dimensions take a path without per-element control flow.
// the output tile's 16-byte stores: default cache policy, or (-DVM_EPI_NT_STORE, an A/B build: tools/build_variant_lib.sh) marked non-temporal.
// [r6] measured inside the step, A B A B in one call: 305.5 / 305.3 ms default, 306.7 / 306.8 ms non-temporal — the output is the next kernel's input; stays default.
#ifdef VM_EPI_NT_STORE
#define EPI_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define EPI_STORE(ptr, val) (*(ptr) = (val))
#endif
template <int ROWS, int COLS>
__device__ __forceinline__ void epi_flush(const char* slab, const GemmParams& p, int64_t m0, int n0, int rows_valid,
                                          int cols_valid, int lane) {
  constexpr int PITCH = EpiSlab<ROWS, COLS>::PITCH;
  constexpr int CPR = COLS / 8;                 // 16-byte chunks per row
  constexpr int RPI = 64 / CPR;                 // rows per wave-instruction
  const int ch = lane % CPR, rr = lane / CPR;
  const unsigned short* rp = (const unsigned short*)p.residual;
  const bool vec_ok = (p.ldc % 8 == 0) && (n0 % 8 == 0) && (!rp || p.ldr % 8 == 0);
  if VM_DBG(p, 64) return;                       // timing experiment: epilogue without stores
  if (vec_ok && cols_valid >= COLS) {
    // full-width column block: whole 16-byte chunks; a ragged last row tile only predicates rows
    unsigned short* cbase = (unsigned short*)p.C + (m0 + rr) * p.ldc + n0 + ch * 8;
    const char* sbase = slab + rr * PITCH + ch * 16;
    const int rlim = rows_valid - rr;            // iteration `it` is live iff it * RPI < rlim
    if (!rp) {
#pragma unroll
      for (int it = 0; it < ROWS / RPI; ++it)
        if (it * RPI < rlim)
          EPI_STORE(reinterpret_cast<u16x8_t*>(cbase + (int64_t)it * RPI * p.ldc), *reinterpret_cast<const u16x8_t*>(sbase + it * RPI * PITCH));
    } else {
      const unsigned short* rbase = rp + (m0 + rr) * p.ldr + n0 + ch * 8;
#pragma unroll
      for (int it = 0; it < ROWS / RPI; ++it) {
        if (it * RPI >= rlim) continue;
        u16x8_t v = *reinterpret_cast<const u16x8_t*>(sbase + it * RPI * PITCH);
        const u16x8_t rv = *reinterpret_cast<const u16x8_t*>(rbase + (int64_t)it * RPI * p.ldr);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(v[e]) + bf2f(rv[e]));
        EPI_STORE(reinterpret_cast<u16x8_t*>(cbase + (int64_t)it * RPI * p.ldc), v);
      }
    }
    return;
  }
  // edge tiles / unaligned leading dimensions: element-wise, not performance relevant
  for (int it = 0; it < ROWS / RPI; ++it) {
    const int r = it * RPI + rr;
    if (r >= rows_valid) continue;
    const int c = ch * 8;
    const int64_t m = m0 + r;
    const unsigned short* sp = reinterpret_cast<const unsigned short*>(slab + r * PITCH + c * 2);
    for (int e = 0; e < 8; ++e) {
      if (c + e >= cols_valid) break;
      float x = bf2f(sp[e]);
      if (rp) x = bf2f(f2bf(x + bf2f(rp[m * p.ldr + n0 + c + e])));
      ((unsigned short*)p.C)[m * p.ldc + n0 + c + e] = f2bf(x);
    }
  }
}
