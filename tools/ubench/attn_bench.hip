// Stand-alone A/B bench + check of the bf16 attention kernels (no torch): includes the library's translation unit, so the
// kernels timed here are the ones that ship. Variants are interleaved in ONE process (guide rule 24), random data (rule 25).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DVM_KEEP_DENORMS tools/ubench/attn_bench.hip -o tools/ubench/attn_bench
//   tools/ubench/attn_bench [rounds] [shape] [forward variant]
// Backward: the round-3 kernels (transposed reads through the builtin, variant 0) against the shipped ones (batched through assembly, 1):
// dQ / dK / dV must agree bit for bit (same MFMA order), dQ and dK/dV are timed separately.
#define VM_ATTN_BENCH_BUILD 1
#ifndef BV0
#define BV0 1          // backward A/B: variants BV0 and BV1 (0 = transposed reads through the builtin, 1 = assembly batches, 2 = 1 + the dS^T workspace)
#endif
#ifndef BV1
#define BV1 2
#endif
#include "../../mmmm_amd/csrc/attn_bf16.hip"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" int vm_prof_begin_(int, void*, void**) { return 0; }
extern "C" int vm_prof_end_(int, void*, void*, double) { return 0; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// reference: one workgroup per (position, head), fp32
__global__ void ref_fwd_k(const unsigned short* q, const unsigned short* k, const unsigned short* v, int64_t ld, const int* cu, int n_seq, int H, int hd,
                          float scale, int causal, float* out, float* lse, int total) {
  const int gpos = blockIdx.x, head = blockIdx.y;
  int seq = 0;
  while (seq + 1 < n_seq && cu[seq + 1] <= gpos) ++seq;
  const int s0 = cu[seq], L = cu[seq + 1] - s0, qp = gpos - s0;
  const int lim = causal ? qp : L - 1;
  __shared__ float sc[8192];
  __shared__ float red[256];
  const int tid = threadIdx.x;
  float mx = -1e30f;
  for (int j = tid; j <= lim; j += blockDim.x) {
    float a = 0.f;
    for (int d = 0; d < hd; ++d) a += bf2f(q[(int64_t)gpos * ld + head * hd + d]) * bf2f(k[(int64_t)(s0 + j) * ld + head * hd + d]);
    a *= scale;
    sc[j] = a;
    mx = fmaxf(mx, a);
  }
  red[tid] = mx; __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
  mx = red[0]; __syncthreads();
  float sum = 0.f;
  for (int j = tid; j <= lim; j += blockDim.x) { const float e = expf(sc[j] - mx); sc[j] = e; sum += e; }
  red[tid] = sum; __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  sum = red[0]; __syncthreads();
  for (int d = tid; d < hd; d += blockDim.x) {
    float a = 0.f;
    for (int j = 0; j <= lim; ++j) a += sc[j] * bf2f(v[(int64_t)(s0 + j) * ld + head * hd + d]);
    out[(int64_t)gpos * H * hd + head * hd + d] = a / sum;
  }
  if (tid == 0) lse[(int64_t)head * total + gpos] = mx + logf(sum);
}

struct Shape { const char* name; std::vector<int> lens; int H, hd, causal; };

static unsigned short f2bf_host(float f) { unsigned u; std::memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float bf2f_host(unsigned short b) { unsigned u = (unsigned)b << 16; float f; std::memcpy(&f, &u, 4); return f; }
static float gauss() { float a = (float)((rand() + 1.0) / (RAND_MAX + 2.0)), b = (float)((rand() + 1.0) / (RAND_MAX + 2.0)); return sqrtf(-2.f * logf(a)) * cosf(6.2831853f * b); }

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 10;
  const int only_shape = argc > 2 ? atoi(argv[2]) : -1;          // run one shape / one variant (profiler passes)
  const int only_variant = argc > 3 ? atoi(argv[3]) : -1;
  const int stress = argc > 4 ? atoi(argv[4]) : 0;                 // determinism stress: that many extra forward + backward launches, each compared bit for bit
  std::vector<Shape> shapes = {
    {"vit-e 8x785 h16 d112", std::vector<int>(8, 785), 16, 112, 0},
    {"decoder 8x456 h32 d128 causal", std::vector<int>(8, 456), 32, 128, 1},
    {"3d 4x4609 h16 d112", std::vector<int>(4, 4609), 16, 112, 0},
    {"grg 8x2049 h16 d112", std::vector<int>(8, 2049), 16, 112, 0},
    {"ragged h2 d128 causal", {130, 64, 1, 200, 456}, 2, 128, 1},
    {"ragged h3 d64", {33, 785, 7}, 3, 64, 0},
  };
  std::vector<int> variants = {8};
  if (only_variant >= 0) variants = {only_variant};
  for (size_t si = 0; si < shapes.size(); ++si) {
    if (only_shape >= 0 && (int)si != only_shape) continue;
    const Shape& sh = shapes[si];
    const int n_seq = (int)sh.lens.size();
    std::vector<int> cu(n_seq + 1, 0);
    int maxlen = 0;
    for (int i = 0; i < n_seq; ++i) { cu[i + 1] = cu[i] + sh.lens[i]; maxlen = std::max(maxlen, sh.lens[i]); }
    const int rows = cu[n_seq], H = sh.H, hd = sh.hd;
    const int64_t ld = 3LL * H * hd;
    std::vector<unsigned short> hq((size_t)rows * ld);
    srand(1234);
    for (auto& x : hq) x = f2bf_host(gauss());
    unsigned short *dqkv, *dout;
    float *dlse, *dref, *dlse_ref;
    int* dcu;
    CK(hipMalloc(&dqkv, hq.size() * 2)); CK(hipMalloc(&dout, (size_t)rows * H * hd * 2));
    CK(hipMalloc(&dlse, (size_t)H * rows * 4)); CK(hipMalloc(&dref, (size_t)rows * H * hd * 4)); CK(hipMalloc(&dlse_ref, (size_t)H * rows * 4));
    CK(hipMalloc(&dcu, (n_seq + 1) * 4));
    CK(hipMemcpy(dqkv, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcu, cu.data(), (n_seq + 1) * 4, hipMemcpyHostToDevice));
    const float scale = 1.0f / sqrtf((float)hd);
    const bool check = (double)rows * maxlen * H < 6e8;
    if (check) {
      hipLaunchKernelGGL(ref_fwd_k, dim3(rows, H), dim3(128), 0, 0, dqkv, dqkv + H * hd, dqkv + 2 * H * hd, ld, dcu, n_seq, H, hd, scale, sh.causal, dref, dlse_ref, rows);
      CK(hipDeviceSynchronize());
    }
    vm_attn_args a = {};
    a.q = dqkv; a.k = dqkv + H * hd; a.v = dqkv + 2 * H * hd; a.out = dout;
    a.ldq = a.ldk = a.ldv = ld; a.ldo = H * hd;
    a.lse = dlse; a.cu_seqlens = dcu; a.n_seq = n_seq; a.row_of_pos = nullptr;
    a.total_pos_max = rows; a.max_seqlen = maxlen; a.n_heads = H; a.head_dim = hd; a.scale = scale; a.causal = sh.causal;
    double flops = 0;
    for (int l : sh.lens) flops += 4.0 * l * l * hd * H * (sh.causal ? 0.5 : 1.0);
    printf("== %s  (%.2f GFLOP)\n", sh.name, flops / 1e9);
    std::vector<std::vector<float>> times(variants.size());
    std::vector<bool> ok(variants.size(), true);
    std::vector<float> href, hlse_ref;
    if (check) { href.resize((size_t)rows * H * hd); hlse_ref.resize((size_t)H * rows);
      CK(hipMemcpy(href.data(), dref, href.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hlse_ref.data(), dlse_ref, hlse_ref.size() * 4, hipMemcpyDeviceToHost)); }
    for (size_t vi = 0; vi < variants.size(); ++vi) {
      CK(hipMemset(dout, 0xFF, (size_t)rows * H * hd * 2));
      const int rc = fwd_launch(&a, 0, variants[vi]);
      if (rc != VM_OK) { ok[vi] = false; printf("  variant %2d: unsupported (rc %d)\n", variants[vi], rc); continue; }
      hipError_t e = hipDeviceSynchronize();
      if (e != hipSuccess) { printf("  variant %2d: launch failed: %s\n", variants[vi], hipGetErrorString(e)); return 1; }
      if (check) {
        std::vector<unsigned short> ho((size_t)rows * H * hd);
        std::vector<float> hl((size_t)H * rows);
        CK(hipMemcpy(ho.data(), dout, ho.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hl.data(), dlse, hl.size() * 4, hipMemcpyDeviceToHost));
        double num = 0, den = 0, mx = 0, lmx = 0;
        for (size_t i = 0; i < ho.size(); ++i) { const double d = bf2f_host(ho[i]) - href[i]; num += d * d; den += (double)href[i] * href[i]; mx = std::max(mx, std::fabs(d)); }
        for (size_t i = 0; i < hl.size(); ++i) lmx = std::max(lmx, (double)std::fabs(hl[i] - hlse_ref[i]));
        printf("  variant %2d: rel L2 err %.2e  max abs %.2e  lse max abs %.2e %s\n", variants[vi], std::sqrt(num / den), mx, lmx,
               (std::sqrt(num / den) < 6e-3 && lmx < 2e-3) ? "OK" : "**** MISMATCH ****");
      }
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < rounds; ++r)
      for (size_t vi = 0; vi < variants.size(); ++vi) {
        if (!ok[vi]) continue;
        const int reps = 5;
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) fwd_launch(&a, 0, variants[vi]);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) times[vi].push_back(ms / reps * 1e3f);
      }
#ifdef A32_STAMPS
    {
      const int nwv = only_variant == 4 ? 4 : 8;
      const int nb = 8 * (((H * n_seq + 7) / 8) * ((maxlen + nwv * 32 - 1) / (nwv * 32)));
      unsigned long long* dbg;
      CK(hipMalloc(&dbg, (size_t)nb * 8 * 12 * 8)); CK(hipMemset(dbg, 0, (size_t)nb * 8 * 12 * 8));
      g_a32_dbg = dbg;
      fwd_launch(&a, 0, nwv);
      CK(hipDeviceSynchronize());
      g_a32_dbg = nullptr;
      std::vector<unsigned long long> hd_((size_t)nb * 8 * 12);
      CK(hipMemcpy(hd_.data(), dbg, hd_.size() * 8, hipMemcpyDeviceToHost));
      double sum[2][12] = {}; double cnt[2] = {0, 0}, tiles[2] = {0, 0};
      for (int b = 0; b < nb; ++b) for (int w = 0; w < nwv; ++w) {
        const unsigned long long* d = &hd_[((size_t)b * nwv + w) * 12];
        if (d[9] == 0) continue;
        const int g = nwv == 8 && w >= 4;
        for (int i = 0; i < 9; ++i) sum[g][i] += (double)d[i];
        sum[g][10] += (double)d[10]; sum[g][11] += (double)d[11];
        cnt[g] += 1; tiles[g] += (double)d[9];
      }
      for (int g = 0; g < 2; ++g) {
        printf("  stamps %s waves (%d live, %.1f tiles each): per tile cycles", g ? "late " : "early", (int)cnt[g], tiles[g] / std::max(cnt[g], 1.0));
        for (int i = 0; i < 4; ++i) printf("  c%d work %5.0f wait %5.0f", i, sum[g][2 * i] / tiles[g], sum[g][2 * i + 1] / tiles[g]);
        printf("  | c2: reads done at %.0f, max + decision at %.0f | loop total %.0f per wave\n", sum[g][10] / tiles[g], sum[g][11] / tiles[g], sum[g][8] / cnt[g]);
      }
      CK(hipFree(dbg));
    }
#endif
    for (size_t vi = 0; vi < variants.size(); ++vi) {
      if (!ok[vi] || times[vi].empty()) continue;
      std::sort(times[vi].begin(), times[vi].end());
      const float med = times[vi][times[vi].size() / 2], mn = times[vi][0];
      printf("  variant %2d: fwd median %7.1f us (min %7.1f)  %6.0f TFLOP/s\n", variants[vi], med, mn, flops / (med * 1e-6) / 1e12);
    }
    {
      unsigned short *ddo, *dgrad[2];
      float* ddelta;
      const size_t no = (size_t)rows * H * hd;
      std::vector<unsigned short> hdo(no);
      for (auto& x : hdo) x = f2bf_host(gauss());
      CK(hipMalloc(&ddo, no * 2)); CK(hipMalloc(&ddelta, (size_t)H * rows * 4));
      CK(hipMemcpy(ddo, hdo.data(), no * 2, hipMemcpyHostToDevice));
      for (int v = 0; v < 2; ++v) CK(hipMalloc(&dgrad[v], hq.size() * 2));
      void* dws = nullptr;
      const int64_t ws_bytes = bwd_ds_bytes(&a);
      CK(hipMalloc(&dws, (size_t)ws_bytes));
      auto bwd = [&](int variant, int which) {
        a.workspace = variant == 2 ? dws : nullptr; a.workspace_bytes = variant == 2 ? ws_bytes : 0;
        return bwd_launch(&a, 0, variant == 2 ? 1 : variant, which);
      };
      fwd_launch(&a, 0, 8);
      a.dout = ddo; a.lddo = H * hd; a.delta = ddelta; a.lddq = a.lddk = a.lddv = ld;
      std::vector<unsigned short> hg[2];
      for (int v = 0; v < 2; ++v) {
        CK(hipMemset(dgrad[v], 0, hq.size() * 2));
        a.dq = dgrad[v]; a.dk = dgrad[v] + H * hd; a.dv = dgrad[v] + 2 * H * hd;
        const int rc = bwd((v ? BV1 : BV0), 7);
        if (rc != VM_OK) { printf("  bwd variant %d: rc %d\n", v, rc); return 1; }
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("  bwd variant %d: launch failed: %s\n", v, hipGetErrorString(e)); return 1; }
        hg[v].resize(hq.size());
        CK(hipMemcpy(hg[v].data(), dgrad[v], hq.size() * 2, hipMemcpyDeviceToHost));
      }
      size_t diff = 0; double nrm = 0, num = 0, den = 0;
      for (size_t i = 0; i < hq.size(); ++i) {
        const double x = bf2f_host(hg[0][i]), y = bf2f_host(hg[1][i]);
        diff += hg[0][i] != hg[1][i]; nrm += std::fabs(y); num += (x - y) * (x - y); den += x * x;
      }
      // (variants 0 / 1 are bit-identical; the workspace form sums dQ over the same bf16 dS in the same key order: identical as well)
      printf("  bwd: variant %d vs %d: %zu of %zu gradient elements differ, rel L2 %.2e (mean |g| %.3e) %s\n", BV1, BV0, diff, hq.size(), std::sqrt(num / std::max(den, 1e-30)),
             nrm / hq.size(), std::sqrt(num / std::max(den, 1e-30)) > 1e-3 ? "**** MISMATCH ****" : "OK");
      if (stress > 0) {
        std::vector<unsigned short> ho0(no), ho1(no), hg1(hq.size());
        CK(hipMemcpy(ho0.data(), dout, no * 2, hipMemcpyDeviceToHost));
        size_t bad_f = 0, bad_b = 0;
        a.dq = dgrad[1]; a.dk = dgrad[1] + H * hd; a.dv = dgrad[1] + 2 * H * hd;
        for (int i = 0; i < stress; ++i) {
          CK(hipMemset(dout, 0xFF, no * 2)); CK(hipMemset(dgrad[1], 0xFF, hq.size() * 2));
          fwd_launch(&a, 0, 8);
          bwd(BV1, 7);
          CK(hipDeviceSynchronize());
          CK(hipMemcpy(ho1.data(), dout, no * 2, hipMemcpyDeviceToHost));
          CK(hipMemcpy(hg1.data(), dgrad[1], hq.size() * 2, hipMemcpyDeviceToHost));
          bad_f += std::memcmp(ho0.data(), ho1.data(), no * 2) != 0;
          bad_b += std::memcmp(hg[1].data(), hg1.data(), hq.size() * 2) != 0;
        }
        printf("  stress: %d launches, forward differed %zu times, backward %zu times %s\n", stress, bad_f, bad_b, (bad_f || bad_b) ? "**** NONDETERMINISTIC ****" : "OK");
      }
      for (int which = 2; which <= 4; which += 2) {
        std::vector<float> tv[2];
        for (int r = 0; r < rounds; ++r)
          for (int v = 0; v < 2; ++v) {
            a.dq = dgrad[v]; a.dk = dgrad[v] + H * hd; a.dv = dgrad[v] + 2 * H * hd;
            const int reps = 5;
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) bwd((v ? BV1 : BV0), which);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) tv[v].push_back(ms / reps * 1e3f);
          }
        for (int v = 0; v < 2; ++v) {
          if (tv[v].empty()) continue;
          std::sort(tv[v].begin(), tv[v].end());
          printf("  bwd %s variant %d: median %7.1f us (min %7.1f)\n", which == 2 ? "dQ   " : "dK/dV", (v ? BV1 : BV0), tv[v][tv[v].size() / 2], tv[v][0]);
        }
      }
#ifdef A32_STAMPS
      for (int v = 0; v < 2; ++v) {
        const int nb = 8 * ((H * n_seq + 7) / 8) * ((maxlen + 127) / 128);
        unsigned long long* dbg;
        CK(hipMalloc(&dbg, (size_t)nb * 8 * 4 * 8)); CK(hipMemset(dbg, 0, (size_t)nb * 8 * 4 * 8));
        g_a32_dbg = dbg;
        bwd((v ? BV1 : BV0), 4);
        CK(hipDeviceSynchronize());
        g_a32_dbg = nullptr;
        std::vector<unsigned long long> hd_((size_t)nb * 8 * 4);
        CK(hipMemcpy(hd_.data(), dbg, hd_.size() * 8, hipMemcpyDeviceToHost));
        double sm[3] = {0, 0, 0}, tl = 0;
        for (size_t i = 0; i < hd_.size(); i += 4) { if (!hd_[i + 3]) continue; for (int k = 0; k < 3; ++k) sm[k] += (double)hd_[i + k]; tl += (double)hd_[i + 3]; }
        printf("  bwd dK/dV variant %d stamps per tile: issue %.0f  compute %.0f  wait+barrier %.0f\n", (v ? BV1 : BV0), sm[0] / tl, sm[1] / tl, sm[2] / tl);
        CK(hipFree(dbg));
      }
#endif
      CK(hipFree(dws));
      CK(hipFree(ddo)); CK(hipFree(ddelta)); CK(hipFree(dgrad[0])); CK(hipFree(dgrad[1]));
    }
    CK(hipFree(dqkv)); CK(hipFree(dout)); CK(hipFree(dlse)); CK(hipFree(dref)); CK(hipFree(dlse_ref)); CK(hipFree(dcu));
  }
  return 0;
}
