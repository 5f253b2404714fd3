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

// One MFMA of the four-wave body: the builtin. (An inline-assembly form with the accumulator constrained to the accumulator file ("+a") was
// built first, when an epilogue loop that the unroller gave up on had turned the accumulators into a scratch array and the allocator shuffled
// tiles between the register halves inside the K loop; it ran, and it was WRONG for the 192-row tile with a LoRA extension: hipcc does not
// know that such a statement is an MFMA, and wherever it places a v_accvgpr_read / _write of its own next to one, the wait states an MFMA's
// result needs are missing — guide section 5.7 item 4. With every accumulator index a compile-time constant the builtin's loops carry no
// accumulator moves at all. -DVM_W4_EXPERIMENT=512 keeps the assembly form for the record.)
__device__ __forceinline__ void w4_mfma(f32x4_t& acc, const bf16x8_t& b, const bf16x8_t& a) {
  if constexpr (W4X & 512) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(b), "v"(a));
  else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc, 0, 0, 0);
}
__device__ __forceinline__ void w4_mfma_drain() {
  if constexpr (W4X & 512) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
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
//   * the accumulators (64 tiles x 4 = 256 registers) live in the accumulator half of the unified register file (hipcc puts them there by
//     itself as long as every accumulator index is a compile-time constant);
//   * LDS (160 KiB): the ACTIVATION tile two K-tiles deep, the WEIGHT tile three deep — [A0 | A1 | B0 | B1 | B2], 32 KiB each, row-major with the
//     16-byte chunk XOR (row & 7) applied to the SOURCE address of the LDS-DMA (guide rule 21), the image of gemm_nt_k. A frozen weight comes from
//     HBM every time it is used and a launch pays 4-5 % for that with one K-tile of lead (7-12 % in the first version of this kernel, which had two
//     stages of (A | B) and lost 3 % inside the step for it: profiles/r6_clock_power.txt); the activation was just written and is cache-warm. vmcnt
//     retires in order, so the deeper operand only keeps its lead if its pieces are the YOUNGEST in the queue when the other operand is waited for:
//     per K-tile the activation pieces of K-tile t + 2 go out first, then the weight pieces of K-tile t + 3, and the barrier's wait leaves the eight
//     weight pieces of K-tile t + 2 in flight;
//   * ONE barrier per K-tile, between its substeps: in front of it every wave has received its second-substep fragments (the last
//     reads of the current stages) and has waited for its own pieces of K-tile t + 1; behind it the reads of K-tile t + 1 and the DMA into the two
//     stages just vacated are legal. 14-16 DMA pieces per wave and K-tile, one behind every fourth MFMA of a second substep.
// Same K order per output element as the eight-wave form (extension tiles, then the main tiles, k ascending): bit-identical results.
//
// PERSISTENT: one workgroup per CU (grid = min(tiles, CUs)) walks tiles v, v + W, v + 2 W, ... of the GROUP_M-grouped, XCD-contiguous tile
// order. What that buys is the seam between two tiles: behind the last K-tile's barrier both LDS stages are free, so the NEXT tile's
// first two K-tiles are requested BEFORE the current tile's epilogue (their HBM / L2 latency runs under the conversion and the slab
// round trip), and the epilogue's global stores are never waited for — they drain under the next tile's K loop (one workgroup per
// tile paid the prologue's latency and the drain of 128 KiB of stores per tile with nothing to hide them: 8-13 us per tile, measured
// with the epilogue knocked out). The output slabs ARE the weight's third stage (8 KiB per wave: 32 rows x 256 B, 16-byte chunks XOR (row & 15) —
// conflict-free for the row-of-4 flush reads, two-way for the 8-byte fragment writes): the seam's prologue fills stages 0 and 1 only, and K-tile 2's
// weight pieces go out behind the next tile's first barrier, when every wave has left its epilogue.
// vmcnt is ONE in-order queue for LDS-DMA pieces and stores: [tile's K-tile 0 pieces][K-tile 1 pieces][previous tile's stores] is the order
// at a seam, so the first two waits of a tile leave exactly the younger operations outstanding (their count is known on the fast path;
// any other epilogue path declares "unknown" and the tile starts with full waits).

// bf16 output, no fused activation (the launcher sends everything else to the eight-wave form)
// One tile of the persistent loop (`first`: the workgroup's first tile — its prologue is issued here, not by a predecessor). EVERYTHING derived
// from the lane id is derived inside, from the laundered `tid` the caller hands over per tile: as invariants of the tile loop hipcc computed the
// ~60 lane-dependent addresses (DMA offsets, fragment offsets, slab and row addresses) once, kept them live across the K loop, spilled them, and
// every reload from scratch is a vector-memory operation behind an s_waitcnt vmcnt(0) — in front of each LDS-DMA inside the K loop, and pass by
// pass through the epilogue, where it drained the next tile's prologue and this tile's stores. Across tiles only scalars survive.
// Returns the id of the next tile of this workgroup (< 0: none); `younger` in / out: see ktile.
template <int TA, bool SCALE>
__device__ __forceinline__ int w4_tile(const GemmParams& p, const int tid, const int id, const int id_end, const int id_step, const bool first, int& younger, int (&rc)[5]) {
  // (the dynamic LDS is named HERE, not handed in as a `char*`: through a generic pointer hipcc no longer saw that the LDS-DMA destination is
  // wave-uniform and wrapped every piece in a waterfall loop — guide T20)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BMT = 32 * TA;           // tile rows: 256 (TA = 8) or 192 (TA = 6)
  constexpr int WR = 16 * TA;            // rows of one wave row
  constexpr int OPB = 256 * 128;         // LDS bytes of one operand tile (A occupies its first BMT rows)
  constexpr int BB = 2 * OPB;            // LDS: [A stage 0 | A stage 1 | B stage 0 | B stage 1 | B stage 2]: the activation two K-tiles deep, the WEIGHT three
  constexpr int SLAB0 = BB + 2 * OPB;    // the output slabs (4 x 8 KiB) ARE B stage 2: free from a tile's last barrier until the next tile's first one
  constexpr int NP = TA + 8;             // LDS-DMA pieces per wave and K-tile
  constexpr int NPASS = TA / 2;          // epilogue passes of 32 rows
  constexpr int NSTORE = NPASS * 8;      // global stores per wave and tile on the interior path
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  struct Tile { int row0, nrows, seg, n0, ncols; __amdgpu_buffer_rsrc_t rA, rB; };          // (the extension's descriptors are built where they are used: K-tile 0's pieces)
  const int lda_b = (int)p.lda * 2, ldb_b = (int)p.ldb * 2;
  // (scalars, said so: as plain ints hipcc kept `kt_ext > 0` as a 0 / 1 VECTOR value across the tile loop, spilled it, and reloaded it per tile behind a vmcnt(0))
  int kt_ext = p.K2 / 64, kt_total = p.K2 / 64 + p.K / 64;
  asm volatile("" : "+s"(kt_ext), "+s"(kt_total));          // pinned to the scalar file (readfirstlane took the detour through a vector register — which was then kept, and spilled)
  const int lda2_b = kt_ext ? (int)p.lda2 * 2 : 0, ldb2_b = kt_ext ? (int)p.ldb2 * 2 : 0;
  // rows / columns of tile `id`; false: the tile has no rows (the token-routed form launches an upper bound of tile rows)
  auto locate = [&](int id, Tile& tl) -> bool {
    const int per_group = GROUP_M * p.tiles_n;
    const int g = id / per_group, gm0 = g * GROUP_M, gsz = min(GROUP_M, p.tiles_m - gm0), rem = id - g * per_group;
    const int tm = gm0 + rem % gsz, tn = rem / gsz;
    gemm_tile_rows<BMT>(p, tm, tl.row0, tl.nrows, tl.seg);
    tl.n0 = tn * 256;
    tl.ncols = min(256, p.N - tl.n0);
    return tl.nrows > 0;
  };
  auto describe = [&](Tile& tl) {
    tl.rA = make_rsrc(p.A, (int64_t)tl.row0 * lda_b, tl.nrows * lda_b);
    tl.rB = make_rsrc(tl.seg ? p.B1 : p.B0, (int64_t)tl.n0 * ldb_b, tl.ncols * ldb_b);
  };
  auto find = [&](int id, Tile& tl) -> int {      // first tile with rows at or after `id` in this workgroup's list, -1: none
    for (; id < id_end; id += id_step)
      if (locate(id, tl)) return id;
    return -1;
  };

  // piece q of this wave: rows 8 (wave + 4 q) .. + 7 of the operand tile, 128 B each; lane -> (row r8, LDS slot): source chunk slot ^ r8
  const int r8 = lane >> 3, c16 = ((lane & 7) ^ r8) * 16;
  const int prow = wave * 8 + r8;
  // per-lane byte offsets of this wave's pieces inside the MAIN operands (invariant over K-tiles: the K-tile travels in the scalar offset)
  int voA[TA], voB[8];
#pragma unroll
  for (int q = 0; q < TA; ++q) voA[q] = (prow + 32 * q) * lda_b + c16;
#pragma unroll
  for (int q = 0; q < 8; ++q) voB[q] = (prow + 32 * q) * ldb_b + c16;
  // idx < TA: A piece idx, else B piece idx - TA (compile-time after unrolling). MAIN: K-tile t is known to be a main tile (no selects)
  // `stg`: byte offset of the destination stage (A: (t & 1) OPB, B: BB + (t % 3) OPB — handed in, the callers keep t % 3 as a rolling scalar)
  auto piece = [&](const Tile& tl, int t, int stg, int idx, auto main_tag) {
    constexpr bool MAIN = decltype(main_tag)::value;
    const bool ext = !MAIN && t < kt_ext;
    // (provably wave-uniform for hipcc: with the K-tile counter in a vector register it wrapped every piece in a waterfall loop over the scalar offset)
    const int koff = __builtin_amdgcn_readfirstlane((ext ? t : t - kt_ext) * 128);
    const bool isb = idx >= TA;
    const int q = isb ? idx - TA : idx;
    char* dst = smem + __builtin_amdgcn_readfirstlane(stg) + (wave + 4 * q) * 1024;
    const int vo = MAIN ? (isb ? voB[q] : voA[q]) : (prow + 32 * q) * (isb ? (ext ? ldb2_b : ldb_b) : (ext ? lda2_b : lda_b)) + c16;
    __amdgpu_buffer_rsrc_t rs = isb ? tl.rB : tl.rA;
    if constexpr (!MAIN) {
      if (ext) rs = isb ? make_rsrc(tl.seg ? p.B2_1 : p.B2_0, (int64_t)tl.n0 * ldb2_b, tl.ncols * ldb2_b) : make_rsrc(p.A2, (int64_t)tl.row0 * lda2_b, tl.nrows * lda2_b);
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, vo, koff, 0, 0);
  };
  auto prologue = [&](const Tile& tl) {          // K-tiles 0 and 1 of a tile (the launcher guarantees >= 3 main K-tiles)
#pragma unroll
    for (int q = 0; q < NP; ++q) { if (!(W4X & 1)) piece(tl, 0, q < TA ? 0 : BB, q, std::false_type{}); }
#pragma unroll
    for (int q = 0; q < NP; ++q) { if (!(W4X & 1)) piece(tl, 1, q < TA ? OPB : BB + OPB, q, std::false_type{}); }
  };

  f32x4_t acc[TA][8];
  bf16x8_t fa[2][TA], fb[2][8];      // fragment sets: [set][tile]; set 0 = first substep (k 0..31), set 1 = second
  const int frow_ = lane & 15, fq_ = lane >> 4;
  const int slot0 = fq_ ^ (frow_ & 7);
  const int offk[2] = {frow_ * 128 + slot0 * 16, frow_ * 128 + (slot0 ^ 4) * 16};
  const int a_base = wm * WR * 128, b_base = wn * 128 * 128;
  // fragment idx of substep ks out of the stages `sa` (activation) / `sb` (weight) into set `set`: idx < 8 a B tile, else an A tile — in the
  // order the next substep consumes them (its first row of MFMAs takes all eight B fragments and A tile 0)
  auto fread = [&](const char* sa, const char* sb, int ks, int set, int idx) {
    if (idx < 8) fb[set][idx] = *reinterpret_cast<const bf16x8_t*>(sb + b_base + idx * 2048 + offk[ks]);
    else fa[set][idx - 8] = *reinterpret_cast<const bf16x8_t*>(sa + a_base + (idx - 8) * 2048 + offk[ks]);
  };
  constexpr int NF = TA + 8;               // fragments per substep
  constexpr int NS = 8 * TA;               // MFMA slots per substep (one MFMA each)
  // A single wave on its SIMD issues in order: an MFMA keeps the matrix pipe busy for 16 cycles and the issue port for 8 of them, every
  // other instruction of the wave costs >= 4 issue cycles — so the companions (fragment reads, DMA pieces with their M0 write) are dealt
  // out ONE PER MFMA SLOT and the order is pinned slot by slot. Clustered behind a row of eight MFMAs (2 reads + 2 pieces + their scalar
  // set-up, ~40 issue cycles) they left the matrix pipe idle for ~30 cycles per row, a quarter of the loop (measured: 8192^3 854 -> 766 us).
  // Slot k of a substep = MFMA (i = k / 8, j = k % 8). Reads go to slots 4 r + 1 (r-th fragment), pieces to slots 4 q + 3.
  constexpr int RSTEP = NS / NF >= 4 ? 4 : NS / NF, PSTEP = NS / NP >= 4 ? 4 : NS / NP;

  // one K-tile; `bs` = t % 3, the weight's stage. H1: K-tile t + 1 exists (its first-substep fragments are read behind the second substep).
  // H2: K-tile t + 2 exists — its ACTIVATION pieces are issued behind the second substep, into the A stage this K-tile leaves; its WEIGHT pieces went
  // out one K-tile earlier and are the 8 youngest operations of the queue at this K-tile's barrier: they stay in flight. H3: K-tile t + 3 exists — its
  // weight pieces are issued behind the second substep into the B stage this K-tile leaves, AFTER the activation pieces: vmcnt retires in order, so
  // an operand only keeps a longer lead if it is the youngest in the queue when the older one is waited for. The weight is the operand that comes
  // from HBM (a frozen weight is read once per pass; the activation was just written): two K-tiles of lead for it, one for the activation.
  // MAIN: K-tiles t + 2 and t + 3 are main tiles for certain.
  // `younger`: vector-memory operations issued between the pieces of K-tile 1 and the weight pieces of K-tile 2 that may still be in flight at the
  // first K-tile's barrier (the previous tile's stores at a seam: NSTORE, or -1 = unknown -> they are waited for)
  auto ktile = [&](const Tile& tl, int t, int bs, int younger, auto first_tag, auto h1_tag, auto h2_tag, auto h3_tag, auto main_tag) {
    constexpr bool H1 = decltype(h1_tag)::value, H2 = decltype(h2_tag)::value, H3 = decltype(h3_tag)::value, FIRST = decltype(first_tag)::value;
    const int sa_off = (t & 1) * OPB, sb_off = BB + bs * OPB;
    const char* curA = smem + sa_off;
    const char* curB = smem + sb_off;
    const char* nxtA = smem + (OPB - sa_off);
    const char* nxtB = smem + BB + (bs == 2 ? 0 : bs + 1) * OPB;
    // ---- substep 0: MFMAs from set 0, second-substep fragments into set 1 (the reads end well in front of the barrier's lgkmcnt(0))
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      w4_mfma(acc[k / 8][k % 8], fb[0][k % 8], fa[0][k / 8]);
      if (k % RSTEP == 1 && k / RSTEP < NF && !(W4X & 2)) fread(curA, curB, 1, 1, k / RSTEP);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (H1 && !(W4X & 4)) {
      // every read of the current stages has returned; this wave's pieces of K-tile t + 1 have landed (the weight pieces of K-tile t + 2 behind
      // them may not have); then everybody's
      constexpr int KEEP = H2 ? 8 : 0;
      if (FIRST && younger == NSTORE) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NSTORE + KEEP) : "memory");
      else if (FIRST && younger == 16) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(16 + KEEP) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(KEEP) : "memory");
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- substep 1: MFMAs from set 1, K-tile t + 1's first fragments into set 0, the DMA of K-tile t + 2 (activation) and t + 3 (weight)
    // into the stages just vacated
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      w4_mfma(acc[k / 8][k % 8], fb[1][k % 8], fa[1][k / 8]);
      if constexpr (H1) { if (k % RSTEP == 1 && k / RSTEP < NF && !(W4X & 2)) fread(nxtA, nxtB, 0, 0, k / RSTEP); }
      if (k % PSTEP == PSTEP - 1 && k / PSTEP < NP && !(W4X & 1)) {
        if constexpr (H2) { if (k / PSTEP < TA) piece(tl, t + 2, sa_off, k / PSTEP, main_tag); }
        if constexpr (H3) { if (k / PSTEP >= TA) piece(tl, t + 3, sb_off, k / PSTEP, main_tag); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // `rc` in: rows / columns of THIS tile (row0, nrows, seg, n0, ncols) — found by the caller for the first tile, by the previous tile's seam for the others (the
  // three integer divisions of `locate` at the loop head needed a scalar that the 256-row body kept in scratch: one more reload behind a vmcnt(0) that waited
  // for the previous tile's stores); out: the next tile's
  Tile cur;
  cur.row0 = rc[0]; cur.nrows = rc[1]; cur.seg = rc[2]; cur.n0 = rc[3]; cur.ncols = rc[4];
  describe(cur);
  if (first) prologue(cur);
  {
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // K-tile 0 has landed (K-tile 1's pieces and, at a seam, the previous tile's stores are younger and may stay in flight)
    if (W4X & 1) {}
    else if (younger == NSTORE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP + NSTORE) : "memory");
    else if (younger == 16) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP + 16) : "memory");
    else if (younger == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // the weight's third stage: K-tile 2's weight pieces go out HERE, behind the tile's first barrier — the stage is the previous tile's slab area, and
    // every wave of the workgroup has left its epilogue now
    // (K-tile 2 is a main tile unless the extension is more than two K-tiles deep: the MAIN form uses the per-piece offsets the K loop keeps in registers.
    // The generic form's base offset was spilled by the 256-row body and reloaded HERE behind an s_waitcnt vmcnt(0) — which waited for the previous
    // tile's stores, the very thing the seam is built to avoid)
    if (!(W4X & 1)) {
      if (kt_ext <= 2) {
#pragma unroll
        for (int q = TA; q < NP; ++q) piece(cur, 2, BB + 2 * OPB, q, std::true_type{});
      } else {
#pragma unroll
        for (int q = TA; q < NP; ++q) piece(cur, 2, BB + 2 * OPB, q, std::false_type{});
      }
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) fread(smem, smem + BB, 0, 0, f);

    // (the scale of the LoRA extension sits BETWEEN the loops, never inside one: see the eight-wave form. The launcher sends a call here only
    // with >= 3 main K-tiles, so every extension tile has three successors and runs the full body; without an extension kt_total >= 3)
    // (the first K-tile of a tile is peeled: it alone looks at `younger`. ONE copy of the main loop and of the three tail bodies serves both
    // cases: with a copy per case hipcc's allocation of the extension case's copy spilled)
    const std::true_type yes{};
    const std::false_type no{};
    int t = 1, bs = 1;
    auto roll = [&]() { ++t; bs = bs == 2 ? 0 : bs + 1; };
    if (kt_ext > 0) {
      ktile(cur, 0, 0, younger, yes, yes, yes, yes, no);
      for (; t < kt_ext; roll()) ktile(cur, t, bs, 0, no, yes, yes, yes, no);
      w4_mfma_drain();
      // SCALE (compile time): the launch needs the extension's scale / dropout mask at all. Its mere PRESENCE costs the 256-row body ~5 us per tile on the
      // path that skips it (hipcc places 25 accumulator-quad spill stores and 156 accumulator reads in front of the branch), so launches with
      // alpha = 1 and no mask — every forward launch of an rsLoRA r = 64, alpha = 8 model — run an instantiation without it
      if constexpr (!SCALE) {
      } else if (__builtin_expect(p.drop_p > 0.f, 0)) {
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            gemm_ext_scale4<true>(p, cur.row0 + wm * WR + i * 16 + frow_, cur.n0 + wn * 128 + j * 16 + fq_ * 4, acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);       // one tile's hash at a time: interleaved for ILP, the 64 hashes spilled the accumulators
          }
      } else if (__builtin_expect(p.alpha2 != 1.f, 0)) {
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[i][j] *= p.alpha2;
      }
    } else if (kt_total > 3) {
      ktile(cur, 0, 0, younger, yes, yes, yes, yes, yes);
    } else {
      ktile(cur, 0, 0, younger, yes, yes, yes, no, yes);
    }
    for (; t + 3 < kt_total; roll()) ktile(cur, t, bs, 0, no, yes, yes, yes, yes);
    if (t + 2 < kt_total) { ktile(cur, t, bs, 0, no, yes, yes, no, yes); roll(); }
    if (t + 1 < kt_total) { ktile(cur, t, bs, 0, no, yes, no, no, yes); roll(); }
    if (t < kt_total) ktile(cur, t, bs, 0, no, no, no, no, yes);
    w4_mfma_drain();
    __builtin_amdgcn_s_barrier();          // every wave is past its last LDS read: all five stages are free

    // ---- the seam: the next tile's first two K-tiles go out before this tile's epilogue. Only what the epilogue needs of THIS tile stays
    // live (five scalars); the next tile's descriptors are rebuilt at the loop head (scalar work) instead of living across the epilogue.
    const int row0 = cur.row0, nrows = cur.nrows, n0 = cur.n0, ncols = cur.ncols, seg = cur.seg;
    const int nid = find(id + id_step, cur);
    if (nid >= 0) { describe(cur); prologue(cur); rc[0] = cur.row0; rc[1] = cur.nrows; rc[2] = cur.seg; rc[3] = cur.n0; rc[4] = cur.ncols; }

    // ---- epilogue: accumulators (+ bias) -> bf16 -> this wave's slab -> whole 256-byte rows of C (+ residual), 32 rows per pass
    int stores = -1;                       // global stores this wave issued for the tile, if the path knows (-1: unknown)
    if (W4X & 8) {          // (every accumulator stays live: guide rule 17)
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (sum == 123.456f) ((float*)p.C)[0] = 1.f;
    } else {
      // (laundered once more: the slab / row addresses are computed HERE, behind the K loop, not in front of it)
      int frow = frow_, fq = fq_, lane_e = lane;
      asm volatile("" : "+v"(frow), "+v"(fq), "+v"(lane_e));
      char* slab = smem + SLAB0 + wave * 8192;
      const unsigned short* bias = (const unsigned short*)(seg ? p.bias1 : p.bias0);
      const unsigned short* rp = (const unsigned short*)p.residual;
      const int rows_left = nrows - wm * WR, cols_left = ncols - wn * 128;      // of this wave's WR x 128 block
      const bool vec_ok = (p.ldc % 8 == 0) && (!rp || p.ldr % 8 == 0);
      const bool interior = rows_left >= WR && cols_left >= 128 && vec_ok;
      // bias of the lane's column groups (guarded element loads: once per tile, any alignment, ragged N)
      f32x4_t bv[8];
      if (bias && cols_left >= 128 && (((uintptr_t)bias + 2 * (uintptr_t)n0) & 7) == 0) {
        // full-width column block, 8-byte aligned: eight vector loads in flight together (the guarded form below is 32 dependent round trips:
        // ~4 us per tile on the ViT-E fc1 shape, where the eight-wave form's epi_bias4 path costs 0.5)
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[j] = epi_bias4(bias, n0 + wn * 128 + j * 16 + fq * 4);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          bv[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
          if (bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int c = j * 16 + fq * 4 + e; if (c < cols_left) bv[j][e] = bf2f(bias[n0 + wn * 128 + c]); }
          }
        }
      }
      const int ch = lane_e & 15, rr = lane_e >> 4;             // flush: 16 chunks of 16 B per row, 4 rows per wave-instruction
      // 32 rows per pass: A tiles 2 ps, 2 ps + 1 (`ps` is a compile-time tag and the passes are separate calls: a loop the unroller gives
      // up on turns the accumulator indices into run-time ones and the 256 accumulators into a scratch array)
      auto pass = [&](auto ps_tag) {
        constexpr int ps = decltype(ps_tag)::value;
        if (ps > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the previous pass's slab reads are in registers
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            f32x4_t a = acc[2 * ps + ii][j];
            if constexpr (W4X & 32) { asm volatile("" :: "v"(a)); continue; }
            if (bias) a += bv[j];
            const u16x4_t o = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3])};
            *reinterpret_cast<u16x4_t*>(slab + (ii * 16 + frow) * 256 + (((2 * j + (fq >> 1)) ^ frow) << 4) + (fq & 1) * 8) = o;
          }
        // the flush reads what the fill wrote through another vector type: nothing but this fence tells hipcc that they alias (without it the
        // reads were scheduled above the later writes). The LDS itself executes a wave's accesses in order: no wait is needed.
        asm volatile("" ::: "memory");
        if constexpr (W4X & 16) return;
        const int64_t m0 = row0 + wm * WR + ps * 32;
        unsigned short* cp = (unsigned short*)p.C + n0 + wn * 128 + ch * 8;
        if (interior) {
          u16x8_t vv[8], rv[8];
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int row = it * 4 + rr;
            vv[it] = *reinterpret_cast<const u16x8_t*>(slab + row * 256 + ((ch ^ (row & 15)) << 4));
          }
          if (rp) {
#pragma unroll
            for (int it = 0; it < 8; ++it) rv[it] = *reinterpret_cast<const u16x8_t*>(rp + (m0 + it * 4 + rr) * p.ldr + n0 + wn * 128 + ch * 8);
#pragma unroll
            for (int it = 0; it < 8; ++it)
#pragma unroll
              for (int e = 0; e < 8; ++e) vv[it][e] = f2bf(bf2f(vv[it][e]) + bf2f(rv[it][e]));
          }
#pragma unroll
          for (int it = 0; it < 8; ++it) *reinterpret_cast<u16x8_t*>(cp + (m0 + it * 4 + rr) * p.ldc) = vv[it];
        } else {
          // edge tiles / odd leading dimensions: the same slab, row and column predicates, compact code (not performance relevant)
#pragma unroll 1
          for (int it = 0; it < 8; ++it) {
            const int row = it * 4 + rr;
            if (ps * 32 + row >= rows_left || ch * 8 >= cols_left) continue;
            u16x8_t x = *reinterpret_cast<const u16x8_t*>(slab + row * 256 + ((ch ^ (row & 15)) << 4));
            const int64_t m = m0 + row;
            const int nv = min(8, cols_left - ch * 8);
            if (rp) {
#pragma unroll
              for (int e = 0; e < 8; ++e) if (e < nv) x[e] = f2bf(bf2f(x[e]) + bf2f(rp[m * p.ldr + n0 + wn * 128 + ch * 8 + e]));
            }
            if (nv == 8 && vec_ok) *reinterpret_cast<u16x8_t*>(cp + m * p.ldc) = x;
            else {
#pragma unroll
              for (int e = 0; e < 8; ++e) if (e < nv) cp[m * p.ldc + e] = x[e];
            }
          }
        }
      };
      pass(std::integral_constant<int, 0>{});
      pass(std::integral_constant<int, 1>{});
      pass(std::integral_constant<int, 2>{});
      if constexpr (TA == 8) pass(std::integral_constant<int, 3>{});
      stores = (interior && !rp && !(W4X & 16)) ? NSTORE : -1;
    }
    younger = stores;
    return nid;
  }
}

template <int TA, bool SCALE = true>
__global__ __launch_bounds__(256, 1) void gemm256w_k(const GemmParams p_in) {
  // the token-routed form keeps its segment boundary and row count on the device: read ONCE per workgroup here (gemm_tile_rows would load them for every
  // tile of the persistent loop — a global load and an s_waitcnt vmcnt(0) at every seam, which waited for the previous tile's stores and the prologue)
  GemmParams p = p_in;
  if (p.counts_dev) {
    const int c0 = __builtin_amdgcn_readfirstlane(p.counts_dev[0]), c1 = __builtin_amdgcn_readfirstlane(p.counts_dev[1]);
    p.split = c0;
    p.M = min(p.M, c1);
    p.counts_dev = nullptr;
  }
  // ---- the tile list of this workgroup. The one-tile-per-workgroup kernels give XCD x (blocks b with b % 8 == x share an L2) one CONTIGUOUS
  // chunk of the GROUP_M-grouped tile order and the dispatcher walks every chunk front to back, 32 tiles at a time: consecutive rounds of an
  // XCD are neighbours in tile space (shared operand panels still in its L2 / the Infinity Cache). Same walk here: workgroup (x = b % 8,
  // slot = b / 8) takes chunk_x[slot], chunk_x[slot + S], ... with S = W / 8 slots per XCD. (W % 8 != 0 only when there are fewer tiles than
  // CUs — one tile per workgroup, through the same bijective map.)
  constexpr int BMT = 32 * TA;
  const int W = gridDim.x;
  const int T = p.tiles_m * p.tiles_n;
  int id0, id_end, id_step;
  {
    const int b = blockIdx.x, x = b & 7;
    if (VM_DBG(p, 2)) { id0 = b; id_end = T; id_step = W; }
    else if (W & 7) { const int q = W >> 3, r = W & 7; id0 = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3); id_end = id0 + 1; id_step = 1; }
    else {
      const int q = T >> 3, r = T & 7;
      const int c0 = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
      id0 = c0 + (b >> 3); id_end = c0 + q + (x < r ? 1 : 0); id_step = W >> 3;
    }
  }
  // first tile with rows (the token-routed form launches an upper bound of tile rows)
  int id = -1;
  int rc[5] = {0, 0, 0, 0, 0};
  for (int c = id0; c < id_end; c += id_step) {
    const int per_group = GROUP_M * p.tiles_n;
    const int g = c / per_group, gm0 = g * GROUP_M, gsz = min(GROUP_M, p.tiles_m - gm0), rem = c - g * per_group;
    int row0, nrows, seg;
    gemm_tile_rows<BMT>(p, gm0 + rem % gsz, row0, nrows, seg);
    if (nrows > 0) { id = c; rc[0] = row0; rc[1] = nrows; rc[2] = seg; rc[3] = (rem / gsz) * 256; rc[4] = min(256, p.N - rc[3]); break; }
  }
  int younger = 0;            // operations behind the two prologue K-tiles in the queue: none for the first tile
  bool first = true;
  if constexpr (W4X & 64) {       // experiment: XCD x starts x * ~3 us late (are the epilogues' store bursts a chip-wide collision?)
    for (int k = 0; k < (int)(blockIdx.x & 7) * 6; ++k) __builtin_amdgcn_s_sleep(16);
  }
  // (the thread id is REBUILT per tile — wave index from a scalar, lane from mbcnt — instead of being kept: the 256-row body kept it in scratch, and its reload
  // at the loop header sat behind an s_waitcnt vmcnt(0) that waited for the previous tile's stores and the prologue every time round)
  const int wave_s = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  while (id >= 0) {
    int lane_l;           // (volatile: as a pure builtin the lane id is loop-invariant, gets hoisted, kept — and spilled again)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_l));
    int tid_l = wave_s * 64 + lane_l;
    id = w4_tile<TA, SCALE>(p, tid_l, id, id_end, id_step, first, younger, rc);
    first = false;
  }
}

