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
#include <vector>
#include <mutex>

namespace {

constexpr int BM = 128, BN = 128;
constexpr int TILE_BYTES = BM * 128;          // 16 KiB per operand tile (128 rows x 128 B)
constexpr int STAGE_BYTES = 2 * TILE_BYTES;   // A + B
constexpr int LDS_BYTES = 2 * STAGE_BYTES;    // double buffer = 64 KiB
constexpr int GROUP_M = 8;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

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
};

// Buffer resource from provably wave-uniform words (avoids hipcc's waterfall loops, guide T20).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const char* base, int64_t byte_off, int bytes) {
  const uint64_t a = (uint64_t)(base + byte_off);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const int n = __builtin_amdgcn_readfirstlane(bytes);
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}

// Issue the LDS-DMA loads of one 128x64 bf16 operand tile. `rsrc` covers the tile's valid rows
// (rows past the end read as zero), `ld_bytes` is the row pitch, `koff` the byte offset of the K-tile.
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, int ld_bytes, int koff,
                                           char* lds_tile, int wave, int lane) {
  const int r8 = lane >> 3, slot = lane & 7;
  const int chunk = slot ^ r8;  // source chunk that lands in LDS slot `slot` of row (.. + r8)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + r8;
    const int voff = row * ld_bytes + chunk * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds_tile + (wave * 4 + i) * 1024), 16, voff, koff, 0, 0);
  }
}

// ESZ = 2: bf16 operands, v_mfma_f32_16x16x32_bf16, K-tile 64.
// ESZ = 4: f32 operands, v_mfma_f32_16x16x4_f32 (exact f32 fma chain), K-tile 32. Same 128-byte LDS rows.
template <int ESZ, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void gemm_nt_k(const GemmParams p) {
  constexpr int BKE = 128 / ESZ;  // K elements per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- tile id: XCD-aware bijective remap, then grouped (GROUP_M) ordering
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int per_group = GROUP_M * p.tiles_n;
  const int g = bid / per_group;
  const int gm0 = g * GROUP_M;
  const int gsz = min(GROUP_M, p.tiles_m - gm0);
  const int tm = gm0 + (bid % per_group) % gsz;
  const int tn = (bid % per_group) / gsz;

  // ---- rows of this m-tile (device-side counts for the token-routed form)
  int M = p.M, split = p.split;
  if (p.counts_dev) {
    split = __builtin_amdgcn_readfirstlane(p.counts_dev[0]);
    M = min(p.M, __builtin_amdgcn_readfirstlane(p.counts_dev[1]));
  }
  int row0, nrows, seg = 0;
  if (split < 0) {
    row0 = tm * BM; nrows = min(BM, M - row0);
  } else {
    split = min(split, M);
    const int t0 = (split + BM - 1) / BM;
    if (tm < t0) { row0 = tm * BM; nrows = min(BM, split - row0); }
    else { seg = 1; row0 = split + (tm - t0) * BM; nrows = min(BM, M - row0); }
  }
  if (nrows <= 0) return;
  const int n0 = tn * BN;
  const int ncols = min(BN, p.N - n0);

  const char* Bw = seg ? p.B1 : p.B0;
  const char* B2w = seg ? p.B2_1 : p.B2_0;

  // ---- buffer resources (wave-uniform): OOB rows read as zero
  const int lda_b = (int)p.lda * ESZ, ldb_b = (int)p.ldb * ESZ;
  __amdgpu_buffer_rsrc_t rA = make_rsrc(p.A, (int64_t)row0 * lda_b, nrows * lda_b);
  __amdgpu_buffer_rsrc_t rB = make_rsrc(Bw, (int64_t)n0 * ldb_b, ncols * ldb_b);

  const int kt_ext = p.K2 / BKE;
  const int kt_main = p.K / BKE;
  const int kt_total = kt_ext + kt_main;

  __amdgpu_buffer_rsrc_t rA2 = rA, rB2 = rB;
  int lda2_b = 0, ldb2_b = 0;
  if (kt_ext > 0) {
    lda2_b = (int)p.lda2 * ESZ; ldb2_b = (int)p.ldb2 * ESZ;
    rA2 = make_rsrc(p.A2, (int64_t)row0 * lda2_b, nrows * lda2_b);
    rB2 = make_rsrc(B2w, (int64_t)n0 * ldb2_b, ncols * ldb2_b);
  }

  auto stage = [&](int t, int buf) {
    char* sa = smem + buf * STAGE_BYTES;       // activation tile (MFMA B operand)
    char* sb = sa + TILE_BYTES;                // weight tile (MFMA A operand)
    if (t < kt_ext) {
      stage_tile(rA2, lda2_b, t * 128, sa, wave, lane);
      stage_tile(rB2, ldb2_b, t * 128, sb, wave, lane);
    } else {
      const int koff = (t - kt_ext) * 128;
      stage_tile(rA, lda_b, koff, sa, wave, lane);
      stage_tile(rB, ldb_b, koff, sb, wave, lane);
    }
  };

  f32x4_t acc[4][4];  // [n-subtile i][m-subtile j]
#pragma unroll
  for (int i = 0; i < 4; ++i)
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

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // tile 0 landed

  for (int t = 0; t < kt_total; ++t) {
    const int buf = t & 1;
    if (t + 1 < kt_total) stage(t + 1, buf ^ 1);
    const char* sa = smem + buf * STAGE_BYTES + wm * (64 * 128);
    const char* sb = smem + buf * STAGE_BYTES + TILE_BYTES + wn * (64 * 128);
    if (ESZ == 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int off = ks ? off_k1 : off_k0;
        bf16x8_t xa[4], wb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xa[j] = *reinterpret_cast<const bf16x8_t*>(sa + j * 2048 + off);
#pragma unroll
        for (int i = 0; i < 4; ++i) wb[i] = *reinterpret_cast<const bf16x8_t*>(sb + i * 2048 + off);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[i], xa[j], acc[i][j], 0, 0, 0);
      }
    } else {
      f32x4_t xa[4][2], wb[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xa[j][0] = *reinterpret_cast<const f32x4_t*>(sa + j * 2048 + off_k0);
        xa[j][1] = *reinterpret_cast<const f32x4_t*>(sa + j * 2048 + off_k1);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wb[i][0] = *reinterpret_cast<const f32x4_t*>(sb + i * 2048 + off_k0);
        wb[i][1] = *reinterpret_cast<const f32x4_t*>(sb + i * 2048 + off_k1);
      }
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[i][s >> 2][s & 3], xa[j][s >> 2][s & 3], acc[i][j], 0, 0, 0);
    }
    if (t + 1 == kt_ext) {
      // end of the LoRA extension: scale, and (dgrad) apply the inverted-dropout mask of the
      // forward's LoRA input element (row m, feature n)
      const float a2 = p.alpha2;
      const bool drop = p.drop_p > 0.f;
      const float inv_keep = drop ? 1.0f / (1.0f - p.drop_p) : 1.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = row0 + wm * 64 + j * 16 + frow;
          const int n = n0 + wn * 64 + i * 16 + fq * 4;
          // n is a multiple of 4 and N % 4 == 0 is required with dropout: the 4 registers share one hash group
          const uint64_t hsh = drop ? vm_hash4(p.drop_seed, ((uint64_t)m * (uint64_t)p.N + (uint64_t)n) >> 2) : 0ull;
          const unsigned thr = vm_drop_threshold(p.drop_p);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float s = a2;
            if (drop) s = vm_keep_bits(hsh, r, thr) ? a2 * inv_keep : 0.f;
            acc[i][j][r] *= s;
          }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // next tile landed and everyone is done reading `buf`
  }

  // ---- epilogue
  const void* bias = seg ? p.bias1 : p.bias0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ml = wm * 64 + j * 16 + frow;
    if (ml >= nrows) continue;
    const int64_t m = row0 + ml;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nl = wn * 64 + i * 16 + fq * 4;
      if (nl >= ncols) continue;
      const int n = n0 + nl;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
      const bool full = nl + 3 < ncols;
      if (OUT_F32) {
        const float* bp = (const float*)bias;
        const float* rp = (const float*)p.residual;
        float* cp = (float*)p.C + m * p.ldc + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (!full && nl + r >= ncols) break;
          float x = v[r];
          if (bp) x += bp[n + r];
          if (p.act == VM_ACT_GELU) x = gelu_erf(x);
          else if (p.act == VM_ACT_RELU) x = fmaxf(x, 0.f);
          if (rp) x += rp[m * p.ldr + n + r];
          v[r] = x;
        }
        if (full) *reinterpret_cast<f32x4_t*>(cp) = (f32x4_t){v[0], v[1], v[2], v[3]};
        else for (int r = 0; r < 4 && nl + r < ncols; ++r) cp[r] = v[r];
      } else {
        const unsigned short* bp = (const unsigned short*)bias;
        const unsigned short* rp = (const unsigned short*)p.residual;
        unsigned short* cp = (unsigned short*)p.C + m * p.ldc + n;
        unsigned short o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (!full && nl + r >= ncols) { o[r] = 0; continue; }
          float x = v[r];
          if (bp) x += bf2f(bp[n + r]);
          // torch rounds the linear's output to bf16 before the activation and before the residual add
          if (p.act == VM_ACT_GELU) x = gelu_erf(bf2f(f2bf(x)));
          else if (p.act == VM_ACT_RELU) x = fmaxf(x, 0.f);
          if (rp) x = bf2f(f2bf(x)) + bf2f(rp[m * p.ldr + n + r]);
          o[r] = f2bf(x);
        }
        if (full) *reinterpret_cast<u16x4_t*>(cp) = (u16x4_t){o[0], o[1], o[2], o[3]};
        else for (int r = 0; r < 4 && nl + r < ncols; ++r) cp[r] = o[r];
      }
    }
  }
}

// ------------------------------------------------------------------ event profiling
struct ProfRec { hipEvent_t a, b; double flops; };
struct ProfState {
  std::mutex mu;
  bool on = false;
  std::vector<ProfRec> recs[4];
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
};
ProfState& prof() { static ProfState s; return s; }

}  // namespace

// shared with the other translation units
extern "C" int vm_prof_begin_(int kind, void* stream, void** tok) {
  ProfState& s = prof();
  if (!s.on) { *tok = nullptr; return 0; }
  std::lock_guard<std::mutex> lk(s.mu);
  ProfRec r;
  if (!s.pool.empty()) { r.a = s.pool.back().first; r.b = s.pool.back().second; s.pool.pop_back(); }
  else { if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { *tok = nullptr; return 0; } }
  r.flops = 0;
  (void)hipEventRecord(r.a, (hipStream_t)stream);
  s.recs[kind].push_back(r);
  *tok = (void*)(uintptr_t)(s.recs[kind].size());  // 1-based index
  return 0;
}
extern "C" int vm_prof_end_(int kind, void* stream, void* tok, double flops) {
  if (!tok) return 0;
  ProfState& s = prof();
  std::lock_guard<std::mutex> lk(s.mu);
  ProfRec& r = s.recs[kind][(size_t)(uintptr_t)tok - 1];
  r.flops = flops;
  (void)hipEventRecord(r.b, (hipStream_t)stream);
  return 0;
}

extern "C" {

int vm_version(void) { return 100; }

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

int vm_prof_enable(int on) { prof().on = on != 0; return VM_OK; }

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
  double ms = 0, fl = 0;
  for (auto& r : s.recs[kind]) {
    if (hipEventSynchronize(r.b) != hipSuccess) return VM_ERR_LAUNCH;
    float t = 0;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return VM_ERR_LAUNCH;
    ms += t; fl += r.flops;
  }
  if (total_ms_host) *total_ms_host = ms;
  if (total_flops_host) *total_flops_host = fl;
  if (launches_host) *launches_host = (int64_t)s.recs[kind].size();
  return VM_OK;
}

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
  if ((int64_t)BN * a->ldb * esz + (int64_t)a->K * esz >= (1ll << 31)) return VM_ERR_UNSUPPORTED;

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
  p.tiles_m = (a->M + BM - 1) / BM + (segmented ? 1 : 0);
  p.tiles_n = (a->N + BN - 1) / BN;
  const int grid = p.tiles_m * p.tiles_n;
  const int kind = esz == 2 ? VM_PROF_GEMM_BF16 : VM_PROF_GEMM_F32;

  void* tok = nullptr;
  vm_prof_begin_(kind, stream, &tok);
  if (esz == 4)
    hipLaunchKernelGGL((gemm_nt_k<4, true>), dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, p);
  else if (a->out_dtype == VM_F32)
    hipLaunchKernelGGL((gemm_nt_k<2, true>), dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL((gemm_nt_k<2, false>), dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, p);
  vm_prof_end_(kind, stream, tok, 2.0 * (double)a->M * (double)a->N * (double)(a->K + a->K2));
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_gemm_bf16(const vm_gemm_args* a, void* stream) { return gemm_launch(a, stream, 2); }
int vm_gemm_f32(const vm_gemm_args* a, void* stream) { return gemm_launch(a, stream, 4); }

}  // extern "C"
