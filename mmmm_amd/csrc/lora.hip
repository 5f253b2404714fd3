// LoRA-side kernels of the fused LoRA linear (peft lora.Linear, conf/lora.yaml r = 64) that do not fit the big
// NT GEMM tile:
//
//  vm_lora_down : t[M,64] = drop(x)[M,K] · A[64,K]^T           (forward A-projection, and u = dy · B in the backward)
//                 skinny output => one 16-row slab per workgroup, K split over its 4 waves, dropout fused on the
//                 activation fragment (no dropped copy of x in HBM), per-row-segment weights for the gated experts.
//  vm_gemm_tn   : C[P,Q] (+)= X[M,P]^T · Y[M,Q]  contracted over token rows (dA, dB, and full weight gradients).
//                 Both operands are staged row-major exactly as they sit in HBM and fed to the MFMA through
//                 ds_read_b64_tr_b16 transposed reads, so no transposed copies of activations are ever written.
#include "vm_common.hpp"
#include "vm_tile.hpp"

extern "C" int vm_prof_begin_(int kind, void* stream, void** tok);
extern "C" int vm_prof_end_(int kind, void* stream, void* tok, double flops);

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int64_t byte_off, int bytes) {
  const uint64_t a = (uint64_t)((const char*)base + byte_off);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const int n = __builtin_amdgcn_readfirstlane(bytes);
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}

// ============================================================================ vm_lora_down
struct DownP {
  const unsigned short* x; int64_t ldx;
  const unsigned short* A0; const unsigned short* A1; int64_t lda;
  unsigned short* t; int64_t ldt;
  int M, K;
  const int32_t* counts_dev; int split;
  float drop_p; uint64_t seed;
};

// 8 waves per 16-row slab: wave w takes the 32-wide k-steps w, w+8, w+16, ... (any K % 32 == 0), 4 k-steps of loads in
// flight per wave; partial sums are reduced through LDS. (4 waves / contiguous K quarters left the chip at ~3.5 waves
// per CU and latency bound: 57 us per call at M=3648, K=4096.)
constexpr int DOWN_WAVES = 8;
__global__ __launch_bounds__(DOWN_WAVES * 64) void lora_down_k(const DownP p) {
  __shared__ float red[DOWN_WAVES][16][64 + 1];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int M = p.M, split = p.split;
  if (p.counts_dev) {
    split = __builtin_amdgcn_readfirstlane(p.counts_dev[0]);
    M = min(p.M, __builtin_amdgcn_readfirstlane(p.counts_dev[1]));
  }
  int row0, nrows, seg = 0;
  const int tb = blockIdx.x;
  if (split < 0) { row0 = tb * 16; nrows = min(16, M - row0); }
  else {
    split = min(split, M);
    const int t0 = (split + 15) / 16;
    if (tb < t0) { row0 = tb * 16; nrows = min(16, split - row0); }
    else { seg = 1; row0 = split + (tb - t0) * 16; nrows = min(16, M - row0); }
  }
  if (nrows <= 0) return;
  const unsigned short* A = seg ? p.A1 : p.A0;
  const int frow = lane & 15, fq = lane >> 4;
  const bool rvalid = frow < nrows;
  const int64_t m = row0 + frow;
  const unsigned short* xr = p.x + (rvalid ? m : row0) * p.ldx;
  const bool drop = p.drop_p > 0.f;
  const unsigned thr = vm_drop_threshold(p.drop_p);
  const float inv_keep = drop ? 1.0f / (1.0f - p.drop_p) : 1.0f;
  const int nsteps = p.K / 32;

  f32x4_t acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  auto load_x = [&](int step) -> u16x8_t {
    u16x8_t xv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (rvalid && step < nsteps) xv = *reinterpret_cast<const u16x8_t*>(xr + step * 32 + 8 * fq);
    return xv;
  };
  for (int s0 = wave; s0 < nsteps; s0 += DOWN_WAVES * 4) {
    u16x8_t xv[4];
    bf16x8_t wa[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int step = s0 + u * DOWN_WAVES;
      xv[u] = load_x(step);
      const int kk = min(step, nsteps - 1) * 32 + 8 * fq;
#pragma unroll
      for (int i = 0; i < 4; ++i) wa[u][i] = *reinterpret_cast<const bf16x8_t*>(A + (int64_t)(16 * i + frow) * p.lda + kk);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int step = s0 + u * DOWN_WAVES;
      if (step >= nsteps) break;
      const int kk = step * 32 + 8 * fq;
      if (drop) {
        const uint64_t idx = (uint64_t)m * (uint64_t)p.K + (uint64_t)kk;       // multiple of 8
        const uint64_t h0 = vm_hash4(p.seed, idx >> 2), h1 = vm_hash4(p.seed, (idx >> 2) + 1);
#pragma unroll
        for (int e = 0; e < 8; ++e)   // same rounding as the standalone dropout kernel: bf16(x * 1/(1-p))
          xv[u][e] = vm_keep_bits(e < 4 ? h0 : h1, e & 3, thr) ? f2bf(bf2f(xv[u][e]) * inv_keep) : (unsigned short)0;
      }
      const bf16x8_t xb = __builtin_bit_cast(bf16x8_t, xv[u]);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[u][i], xb, acc[i], 0, 0, 0);
    }
  }
  // D[row = r_local][col = m_local]: lane holds r = 16 i + 4 fq + e for m = frow
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][frow][16 * i + 4 * fq + e] = acc[i][e];
  __syncthreads();
  // 16 x 64 outputs over 512 threads: 2 consecutive r per thread
  const int om = tid >> 5, orr = (tid & 31) * 2;
  if (om < nrows) {
    float v0 = 0.f, v1 = 0.f;
#pragma unroll
    for (int w = 0; w < DOWN_WAVES; ++w) { v0 += red[w][om][orr]; v1 += red[w][om][orr + 1]; }
    const unsigned packed = (unsigned)f2bf(v0) | ((unsigned)f2bf(v1) << 16);
    *reinterpret_cast<unsigned*>(p.t + (int64_t)(row0 + om) * p.ldt + orr) = packed;
  }
}

// ============================================================================ vm_gemm_tn
struct TnP {
  const unsigned short* X; int64_t ldx; int P;
  const unsigned short* Y; int64_t ldy; int Q;
  void* C; int64_t ldc;
  int M;
  const int32_t* counts_dev; int segment;   // row range from device counts: -1 = [0, M), 0 = [0,c0), 1 = [c0,c1)
  const int32_t* nrows_dev;                 // optional device row count when segment < 0
  int splits;                               // > 1: fp32 atomic accumulation into C (pre-zeroed)
  int out_f32;
  float drop_p; uint64_t seed; int drop_cols;   // inverted dropout on Y: element (m, q) of a [*, drop_cols] tensor
  float alpha;
  int tiles_p, tiles_q;
};

// stage 32 rows x 128 bf16 columns (256 B per row) of a row-major operand: LDS image is lane-linear, the swizzle is
// applied to the SOURCE chunk (rule 21); rows / columns outside the operand read as zero (buffer bounds + ld check)
__device__ __forceinline__ void stage_rows(const unsigned short* base, int64_t ld, int row_begin, int rows_valid,
                                           int col0, int cols_total, char* tile, int wave, int lane) {
  // the buffer covers rows [row_begin, row_begin + rows_valid); per-lane offset = r * ld*2 + col bytes
  const int ld_b = (int)ld * 2;
  __amdgpu_buffer_rsrc_t rs = make_rsrc(base, (int64_t)row_begin * ld_b, rows_valid > 0 ? rows_valid * ld_b : 0);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int inst = wave * 2 + i;                 // 8 wave-instructions of 1 KiB = 4 rows each
    const int row = inst * 4 + (lane >> 4);
    const int slot = lane & 15;
    const int chunk = slot ^ swz(row);
    const int col = col0 + chunk * 8;
    // columns past the operand width must read zero: push the offset out of the buffer
    const int voff = (col + 8 <= cols_total) ? row * ld_b + col * 2 : 0x7FFFFFF0;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(tile + inst * 1024), 16, voff, 0, 0, 0);
  }
}

__global__ __launch_bounds__(256, 2) void gemm_tn_k(const TnP p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * 32 * ROWB];   // [buf][X|Y][32 rows][256 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave >> 1, wq = wave & 1;
  const int tile = blockIdx.x;
  const int tp = tile / p.tiles_q, tq = tile % p.tiles_q;
  const int p0 = tp * 128, q0 = tq * 128;

  int rb = 0, re = p.M;
  if (p.counts_dev) {
    const int c0 = __builtin_amdgcn_readfirstlane(p.counts_dev[0]);
    const int c1 = min(p.M, __builtin_amdgcn_readfirstlane(p.counts_dev[1]));
    if (p.segment == 0) { rb = 0; re = min(c0, c1); }
    else if (p.segment == 1) { rb = min(c0, c1); re = c1; }
    else { rb = 0; re = c1; }
  } else if (p.nrows_dev) {
    re = min(p.M, __builtin_amdgcn_readfirstlane(p.nrows_dev[0]));
  }
  // split the row range over blockIdx.y in multiples of 32 rows
  const int total_steps = (re - rb + 31) / 32;
  const int per = (total_steps + p.splits - 1) / p.splits;
  const int s_begin = blockIdx.y * per, s_end = min(total_steps, s_begin + per);
  if (s_begin >= s_end && !(p.splits == 1)) return;

  f32x16_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  auto stage = [&](int step, int buf) {
    const int r0 = rb + step * 32;
    const int valid = min(32, re - r0);
    char* sx = smem + buf * (2 * 32 * ROWB);
    char* sy = sx + 32 * ROWB;
    stage_rows(p.X, p.ldx, r0, valid, p0, p.P, sx, wave, lane);
    stage_rows(p.Y, p.ldy, r0, valid, q0, p.Q, sy, wave, lane);
  };
  const bool drop = p.drop_p > 0.f;
  const unsigned thr = vm_drop_threshold(p.drop_p);
  const float inv_keep = drop ? 1.0f / (1.0f - p.drop_p) : 1.0f;

  if (s_begin < s_end) stage(s_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int st = s_begin; st < s_end; ++st) {
    const int buf = (st - s_begin) & 1;
    if (st + 1 < s_end) stage(st + 1, buf ^ 1);
    char* sx = smem + buf * (2 * 32 * ROWB);
    char* sy = sx + 32 * ROWB;
    if (drop) {
      // apply the inverted-dropout mask to the Y tile in place: thread -> 2 chunks of 8 consecutive columns
      const int r0 = rb + st * 32;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = tid + i * 256;
        const int row = c >> 4, chunk = c & 15;
        const int col = q0 + chunk * 8;
        char* addr = sy + tile_off(row, chunk);
        u16x8_t v = *reinterpret_cast<u16x8_t*>(addr);
        const uint64_t idx = (uint64_t)(r0 + row) * (uint64_t)p.drop_cols + (uint64_t)col;
        const uint64_t h0 = vm_hash4(p.seed, idx >> 2), h1 = vm_hash4(p.seed, (idx >> 2) + 1);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          v[e] = vm_keep_bits(e < 4 ? h0 : h1, e & 3, thr) ? f2bf(bf2f(v[e]) * inv_keep) : (unsigned short)0;
        *reinterpret_cast<u16x8_t*>(addr) = v;
      }
      __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8_t xa[2], yb[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) xa[a] = frag_tr(sx, 16 * s, 2 * wp + a, lane);
#pragma unroll
      for (int b = 0; b < 2; ++b) yb[b] = frag_tr(sy, 16 * s, 2 * wq + b, lane);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[a], yb[b], acc[a][b], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // D[i = p_local][j = q_local]: col = lane & 31 -> q, rows (r&3) + 8 (r>>2) + 4 h -> p
  const int h = lane >> 5;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int qq = q0 + wq * 64 + b * 32 + (lane & 31);
      if (qq >= p.Q) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pp = p0 + wp * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (pp >= p.P) continue;
        const float v = acc[a][b][r] * p.alpha;
        if (p.splits > 1) atomicAdd((float*)p.C + (int64_t)pp * p.ldc + qq, v);
        else if (p.out_f32) ((float*)p.C)[(int64_t)pp * p.ldc + qq] = v;
        else ((unsigned short*)p.C)[(int64_t)pp * p.ldc + qq] = f2bf(v);
      }
    }
}

}  // namespace

extern "C" {

int vm_lora_down(const void* x, int64_t ldx, const void* A0, const void* A1, int64_t lda, void* t, int64_t ldt,
                 int M, int K, int R, const int32_t* counts_dev, int split, float drop_p, uint64_t drop_seed, void* stream) {
  if (!x || !A0 || !t) return VM_ERR_BAD_ARG;
  if (M <= 0) return VM_OK;
  if (R != 64 || K % 32 || ldx % 8 || lda % 8 || ldt % 2) return VM_ERR_UNSUPPORTED;
  const bool segmented = counts_dev != nullptr || split >= 0;
  if (segmented && !A1) return VM_ERR_BAD_ARG;
  DownP p;
  p.x = (const unsigned short*)x; p.ldx = ldx;
  p.A0 = (const unsigned short*)A0; p.A1 = (const unsigned short*)(A1 ? A1 : A0); p.lda = lda;
  p.t = (unsigned short*)t; p.ldt = ldt;
  p.M = M; p.K = K;
  p.counts_dev = counts_dev;
  p.split = segmented ? (counts_dev ? 0 : split) : -1;
  p.drop_p = drop_p; p.seed = drop_seed;
  const int grid = (M + 15) / 16 + (segmented ? 1 : 0);
  void* tok = nullptr;
  vm_prof_begin_(VM_PROF_LORA, stream, &tok);
  hipLaunchKernelGGL(lora_down_k, dim3(grid), dim3(DOWN_WAVES * 64), 0, (hipStream_t)stream, p);
  vm_prof_end_(VM_PROF_LORA, stream, tok, 2.0 * M * 64.0 * K);
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_gemm_tn_bf16(const void* X, int64_t ldx, int P, const void* Y, int64_t ldy, int Q, void* C, int64_t ldc,
                    int out_dtype, int M, const int32_t* counts_dev, int segment, const int32_t* nrows_dev, int splits,
                    float alpha, float drop_p, uint64_t drop_seed, int drop_cols, void* stream) {
  if (!X || !Y || !C) return VM_ERR_BAD_ARG;
  if (P <= 0 || Q <= 0) return VM_OK;
  if (ldx % 8 || ldy % 8 || P % 8 || Q % 8) return VM_ERR_BAD_ARG;
  if (splits < 1) splits = 1;
  if (splits > 1 && out_dtype != VM_F32) return VM_ERR_BAD_ARG;
  if ((int64_t)32 * ldx * 2 + 256 >= (1ll << 31) || (int64_t)32 * ldy * 2 + 256 >= (1ll << 31)) return VM_ERR_UNSUPPORTED;
  TnP p;
  p.X = (const unsigned short*)X; p.ldx = ldx; p.P = P;
  p.Y = (const unsigned short*)Y; p.ldy = ldy; p.Q = Q;
  p.C = C; p.ldc = ldc; p.M = M;
  p.counts_dev = counts_dev; p.segment = counts_dev ? segment : -1;
  p.nrows_dev = nrows_dev;
  p.splits = splits; p.out_f32 = out_dtype == VM_F32;
  p.drop_p = drop_p; p.seed = drop_seed; p.drop_cols = drop_cols;
  p.alpha = alpha;
  p.tiles_p = (P + 127) / 128; p.tiles_q = (Q + 127) / 128;
  void* tok = nullptr;
  vm_prof_begin_(VM_PROF_LORA, stream, &tok);
  hipLaunchKernelGGL(gemm_tn_k, dim3(p.tiles_p * p.tiles_q, splits), dim3(256), 0, (hipStream_t)stream, p);
  vm_prof_end_(VM_PROF_LORA, stream, tok, 2.0 * (double)M * P * Q);
  VM_LAUNCH_CHECK();
  return VM_OK;
}

}  // extern "C"
