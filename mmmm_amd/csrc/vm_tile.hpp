// 256-byte-row LDS tiles of bf16 read as 32x32x16 MFMA operands, row-wise (ds_read_b128) or transposed
// (ds_read_b64_tr_b16). Shared by the attention kernels and the row-contraction (TN) GEMM.
#pragma once
#include "vm_common.hpp"

constexpr int ROWB = 256;            // LDS row pitch in bytes (128 bf16)

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

// XOR swizzle of the 16-byte chunk index, conflict-free for both ds_read_b128 row reads and
// ds_read_b64_tr_b16 transposed reads of a 256-byte-row tile (guide T10 image (b)).
__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int tile_off(int row, int chunk) { return row * ROWB + ((chunk ^ swz(row)) << 4); }

// A/B operand fragment of a 32x32x16 MFMA read along rows: lane -> row r0+(lane&31), k = 16*s + 8*(lane>>5) + 0..7
__device__ __forceinline__ bf16x8_t frag_row(const char* tile, int r0, int s, int lane) {
  return *reinterpret_cast<const bf16x8_t*>(tile + tile_off(r0 + (lane & 31), 2 * s + (lane >> 5)));
}

// Transposed fragment: operand element j of lane half h is tile[row = rbase + 8*(j>>2) + 4*h + (j&3)][col = 32*b + (lane&31)]
// (the k order in which an accumulator tile's registers 8s..8s+7 appear as the other operand).
__device__ __forceinline__ bf16x8_t frag_tr(const char* tile, int rbase, int b, int lane) {
  const int i = lane & 15, g = (lane >> 4) & 1, h = lane >> 5;
  const int q4 = i >> 2, pp = i & 3;
  const int chunk = 4 * b + 2 * g + (pp >> 1);
  const int rowA = rbase + 4 * h + q4, rowB = rowA + 8;
  const char* a = tile + tile_off(rowA, chunk) + ((pp & 1) << 3);
  const char* c = tile + tile_off(rowB, chunk) + ((pp & 1) << 3);
  u16x4_t lo = __builtin_bit_cast(u16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a)));
  u16x4_t hi = __builtin_bit_cast(u16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(c)));
  u16x8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, r);
}


// frag_tr through inline assembly. hipcc cannot see what a ds_read_b64_tr_b16 BUILTIN reads: it drains the LDS-DMA queue (s_waitcnt
// vmcnt(0)) in front of the first one that follows a DMA issue — in a ring that stages step i + 1 and then computes step i, the next
// step's loads are waited for before the current step's first MFMA, i.e. nothing overlaps inside a workgroup — and it waits for every
// pair of reads on its own. The assembly form is invisible to it: a batch is issued, then tr_asm_wait() (a wait-only statement +
// sched_barrier: the consumers cannot move above it), then the MFMAs.
// Lane addresses: frag_tr(tile, 16 ks, b, lane) reads tile rows 16 ks + 4 h + q4 (lo) and + 8 (hi). The image's XOR puts the column
// block b at bits 6..7 of the address as b ^ q4, so ONE base per row group serves every block ("^ (b << 6)"; tile bases are multiples
// of 256 bytes), ks rides as the immediate 16 ks ROWB, and the operand tile / ring stage are lane-uniform adds.
struct TrLane { unsigned a, b; };          // addresses of block 0, rows 4 h + q4 and 8 + 4 h + q4, in tile 0
__device__ __forceinline__ TrLane tr_lane(const char* tile, int lane) {
  const int i = lane & 15, g = (lane >> 4) & 1, h = lane >> 5;
  const int q4 = i >> 2, pp = i & 3;
  const int chunk = 2 * g + (pp >> 1);
  const int rowA = 4 * h + q4, rowB = rowA + 8;
  return {(unsigned)(size_t)tile + tile_off(rowA, chunk) + ((pp & 1) << 3), (unsigned)(size_t)tile + tile_off(rowB, chunk) + ((pp & 1) << 3)};
}
template <int IMM>
__device__ __forceinline__ bf16x8_t tr_asm2(unsigned addrA, unsigned addrB) {
  s16x4_t lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addrA), "i"(IMM));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addrB), "i"(IMM));
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ void tr_asm_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
