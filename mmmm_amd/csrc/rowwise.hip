// Row-wise / element-wise HBM-bound kernels of the VividMed training step.
// One wavefront (64 lanes) owns one row; 16-byte vector accesses; fp32 maths.
// Reference op sites are cited per entry point in include/vividmed_hip.h.
#include "vm_common.hpp"
#include <mutex>
#include <algorithm>

namespace {

constexpr int ROW_WAVES = 4;          // waves (= rows in flight) per workgroup
constexpr int ROW_THREADS = ROW_WAVES * 64;

template <typename T>
__device__ __forceinline__ typename Elem<T>::vec_t ldv(const T* p) {
  return *reinterpret_cast<const typename Elem<T>::vec_t*>(p);
}
template <typename T>
__device__ __forceinline__ void stv(T* p, typename Elem<T>::vec_t v) {
  *reinterpret_cast<typename Elem<T>::vec_t*>(p) = v;
}
// non-temporal forms (the lines are marked for early eviction). Measured (tools/bench_gelu_after_gemm.py, gpurun_out/s2_gelu_nt.log): the GELU
// backward — three [6280 x 15360] streams — 107 -> 90 us with both loads and the store non-temporal (95 with the loads only), -1 ms per step
// (A B A B: 312.8 / 312.2 vs 311.2 / 311.7); the forward (two streams) 67 -> 72 us: not used there; silu_mul_bwd and adamw_k: no change.
template <typename T>
__device__ __forceinline__ typename Elem<T>::vec_t ldv_nt(const T* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const typename Elem<T>::vec_t*>(p));
}
template <typename T>
__device__ __forceinline__ void stv_nt(T* p, typename Elem<T>::vec_t v) {
  __builtin_nontemporal_store(v, reinterpret_cast<typename Elem<T>::vec_t*>(p));
}

// ---------------------------------------------------------------- RMSNorm
template <typename T>
__global__ __launch_bounds__(ROW_THREADS) void rmsnorm_fwd_k(
    const T* __restrict__ x, const T* __restrict__ w, T* __restrict__ y, float* __restrict__ rstd_out,
    int rows, int cols, float eps, const int32_t* nrows_dev) {
  constexpr int V = Elem<T>::VEC;
  if (nrows_dev) rows = min(rows, *nrows_dev);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + (int64_t)row * cols;
  T* yr = y + (int64_t)row * cols;
  float ss = 0.f;
  for (int c = lane * V; c < cols; c += 64 * V) {
    auto v = ldv<T>(xr + c);
#pragma unroll
    for (int i = 0; i < V; ++i) { float f = Elem<T>::ld(v[i]); ss += f * f; }
  }
  ss = wave_sum(ss);
  const float r = rsqrtf(ss / (float)cols + eps);
  if (lane == 0 && rstd_out) rstd_out[row] = r;
  for (int c = lane * V; c < cols; c += 64 * V) {
    auto v = ldv<T>(xr + c);
    auto wv = ldv<T>(w + c);
    typename Elem<T>::vec_t o;
#pragma unroll
    for (int i = 0; i < V; ++i) o[i] = Elem<T>::st(Elem<T>::ld(wv[i]) * (Elem<T>::ld(v[i]) * r));
    stv<T>(yr + c, o);
  }
}

// dx only (one wave per row, full grid); the parameter gradients are column sums done by norm_bwd_dwdb_k
template <typename T>
__global__ __launch_bounds__(ROW_THREADS) void rmsnorm_bwd_k(
    const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ dy,
    const float* __restrict__ rstd, T* __restrict__ dx, int rows, int cols, const int32_t* nrows_dev,
    const T* __restrict__ dx_add = nullptr) {      // dx_add: gradient of the residual branch that forked off x, summed in (one rounding)
  constexpr int V = Elem<T>::VEC;
  if (nrows_dev) rows = min(rows, *nrows_dev);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + (int64_t)row * cols;
  const T* dyr = dy + (int64_t)row * cols;
  const T* addr = dx_add ? dx_add + (int64_t)row * cols : nullptr;
  T* dxr = dx + (int64_t)row * cols;
  const float r = rstd[row];
  float dot = 0.f;
  for (int c = lane * V; c < cols; c += 64 * V) {
    auto xv = ldv<T>(xr + c); auto gv = ldv<T>(dyr + c); auto wv = ldv<T>(w + c);
#pragma unroll
    for (int i = 0; i < V; ++i) dot += Elem<T>::ld(wv[i]) * Elem<T>::ld(gv[i]) * (Elem<T>::ld(xv[i]) * r);
  }
  dot = wave_sum(dot) / (float)cols;
  for (int c = lane * V; c < cols; c += 64 * V) {
    auto xv = ldv<T>(xr + c); auto gv = ldv<T>(dyr + c); auto wv = ldv<T>(w + c);
    typename Elem<T>::vec_t o, av;
    if (addr) av = ldv<T>(addr + c);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float xh = Elem<T>::ld(xv[i]) * r;
      o[i] = Elem<T>::st(r * (Elem<T>::ld(wv[i]) * Elem<T>::ld(gv[i]) - xh * dot) + (addr ? Elem<T>::ld(av[i]) : 0.f));
    }
    stv<T>(dxr + c, o);
  }
}

// dw[c] += sum_r dy[r,c] * xhat[r,c] ; db[c] += sum_r dy[r,c]   with xhat = (x - mean[r]) * rstd[r]  (mean == NULL -> 0).
// Grid: (cols / 256 column groups) x (row chunks of 32); a lane owns 4 consecutive columns (8- or 16-byte loads: a wave reads
// 512 B / 1 KiB of a row per instruction — the first form, one 2-byte column per lane, ran at 2.1 TB/s), the 4 waves of a block
// interleave rows and are summed through LDS before one atomic per column and block.
template <typename T>
__global__ __launch_bounds__(256) void norm_bwd_dwdb_k(
    const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ mean, const float* __restrict__ rstd,
    float* __restrict__ dw_accum, float* __restrict__ db_accum, int rows, int cols, const int32_t* nrows_dev) {
  __shared__ float red[2][4][256];
  if (nrows_dev) rows = min(rows, *nrows_dev);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const int r0 = blockIdx.y * 32;
  const int r1 = min(r0 + 32, rows);
  float aw[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
  typedef T vec4_t __attribute__((ext_vector_type(4)));
  if (c < cols) {          // (cols % 4 == 0: a lane's four columns are all inside or all outside)
    // a wave's eight rows of the chunk are requested together (a run-time row loop issued one pair of loads per round trip); rows past the
    // chunk's end re-read its first row and are not added. Same order of additions per lane as the loop.
    vec4_t gv[8], xv[8];
    float mu[8], rs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = r0 + wid + 4 * k < r1 ? r0 + wid + 4 * k : r0;
      gv[k] = *reinterpret_cast<const vec4_t*>(dy + (int64_t)r * cols + c);
      xv[k] = *reinterpret_cast<const vec4_t*>(x + (int64_t)r * cols + c);
      mu[k] = mean ? mean[r] : 0.f;
      rs[k] = rstd[r];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (r0 + wid + 4 * k >= r1) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float g = Elem<T>::ld(gv[k][i]);
        aw[i] += g * ((Elem<T>::ld(xv[k][i]) - mu[k]) * rs[k]);
        ab[i] += g;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { red[0][wid][lane * 4 + i] = aw[i]; red[1][wid][lane * 4 + i] = ab[i]; }
  __syncthreads();
  const int cc = blockIdx.x * 256 + threadIdx.x;          // one column per thread for the final sum
  if (cc < cols && r0 < rows) {
    const int t = threadIdx.x;
    if (dw_accum) atomicAdd(dw_accum + cc, red[0][0][t] + red[0][1][t] + red[0][2][t] + red[0][3][t]);
    if (db_accum) atomicAdd(db_accum + cc, red[1][0][t] + red[1][1][t] + red[1][2][t] + red[1][3][t]);
  }
}

// ---------------------------------------------------------------- LayerNorm
template <typename T>
__global__ __launch_bounds__(ROW_THREADS) void layernorm_fwd_k(
    const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ b, const T* __restrict__ res,
    T* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out,
    int rows, int cols, float eps) {
  constexpr int V = Elem<T>::VEC;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + (int64_t)row * cols;
  float s = 0.f;
  for (int c = lane * V; c < cols; c += 64 * V) {
    auto v = ldv<T>(xr + c);
#pragma unroll
    for (int i = 0; i < V; ++i) s += Elem<T>::ld(v[i]);
  }
  const float mu = wave_sum(s) / (float)cols;
  float ss = 0.f;
  for (int c = lane * V; c < cols; c += 64 * V) {
    auto v = ldv<T>(xr + c);
#pragma unroll
    for (int i = 0; i < V; ++i) { const float d = Elem<T>::ld(v[i]) - mu; ss += d * d; }
  }
  const float r = rsqrtf(wave_sum(ss) / (float)cols + eps);
  if (lane == 0) { if (mean_out) mean_out[row] = mu; if (rstd_out) rstd_out[row] = r; }
  T* yr = y + (int64_t)row * cols;
  const T* rr = res ? res + (int64_t)row * cols : nullptr;
  for (int c = lane * V; c < cols; c += 64 * V) {
    auto v = ldv<T>(xr + c);
    typename Elem<T>::vec_t o;
    typename Elem<T>::vec_t wv, bv, rv;
    if (w) wv = ldv<T>(w + c);
    if (b) bv = ldv<T>(b + c);
    if (rr) rv = ldv<T>(rr + c);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      float f = (Elem<T>::ld(v[i]) - mu) * r;
      if (w) f *= Elem<T>::ld(wv[i]);
      if (b) f += Elem<T>::ld(bv[i]);
      // torch rounds LN's output to the storage dtype before the residual add
      if (rr) f = Elem<T>::ld(Elem<T>::st(f)) + Elem<T>::ld(rv[i]);
      o[i] = Elem<T>::st(f);
    }
    stv<T>(yr + c, o);
  }
}

template <typename T>
__global__ __launch_bounds__(ROW_THREADS) void layernorm_bwd_k(
    const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ dy,
    const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dx, int rows, int cols,
    const T* __restrict__ dx_add = nullptr) {
  constexpr int V = Elem<T>::VEC;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + (int64_t)row * cols;
  const T* dyr = dy + (int64_t)row * cols;
  const T* addr = dx_add ? dx_add + (int64_t)row * cols : nullptr;
  T* dxr = dx + (int64_t)row * cols;
  const float mu = mean[row], r = rstd[row];
  float s1 = 0.f, s2 = 0.f;  // sum(w*dy), sum(w*dy*xhat)
  for (int c = lane * V; c < cols; c += 64 * V) {
    auto xv = ldv<T>(xr + c); auto gv = ldv<T>(dyr + c);
    typename Elem<T>::vec_t wv; if (w) wv = ldv<T>(w + c);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float xh = (Elem<T>::ld(xv[i]) - mu) * r, g = Elem<T>::ld(gv[i]);
      const float wg = w ? Elem<T>::ld(wv[i]) * g : g;
      s1 += wg; s2 += wg * xh;
    }
  }
  s1 = wave_sum(s1) / (float)cols; s2 = wave_sum(s2) / (float)cols;
  for (int c = lane * V; c < cols; c += 64 * V) {
    auto xv = ldv<T>(xr + c); auto gv = ldv<T>(dyr + c);
    typename Elem<T>::vec_t wv; if (w) wv = ldv<T>(w + c);
    typename Elem<T>::vec_t o, av;
    if (addr) av = ldv<T>(addr + c);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float xh = (Elem<T>::ld(xv[i]) - mu) * r, g = Elem<T>::ld(gv[i]);
      const float wg = w ? Elem<T>::ld(wv[i]) * g : g;
      o[i] = Elem<T>::st(r * (wg - s1 - xh * s2) + (addr ? Elem<T>::ld(av[i]) : 0.f));
    }
    stv<T>(dxr + c, o);
  }
}

// ---------------------------------------------------------------- norms with the row held in registers
// The kernels above walk a row two or three times with a run-time trip count, so every pass is a new trip to L1 / L2 (the rows of four waves
// plus their neighbours do not stay in a 32 KiB L1). Rows of up to NV x 64 vectors (bf16: 2 048 / 4 096 columns for NV = 4 / 8, fp32: 1 024 /
// 2 048) are loaded ONCE into NV vector registers per operand; the passes then run on registers. Same per-lane order of additions as the
// generic kernels (column chunks ascending), so the results are bit-identical (tests/test_kernels_gpu.py: *_register_rows_*).
#define VM_ROW_LOOP(j, c) _Pragma("unroll") for (int j = 0, c = lane * V; j < NV; ++j, c += 64 * V) if (c < cols)

template <typename T, int NV>
__global__ __launch_bounds__(ROW_THREADS) void rmsnorm_fwd_r_k(
    const T* __restrict__ x, const T* __restrict__ w, T* __restrict__ y, float* __restrict__ rstd_out,
    int rows, int cols, float eps, const int32_t* nrows_dev) {
  constexpr int V = Elem<T>::VEC;
  if (nrows_dev) rows = min(rows, *nrows_dev);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + (int64_t)row * cols;
  T* yr = y + (int64_t)row * cols;
  typename Elem<T>::vec_t xv[NV];
  VM_ROW_LOOP(j, c) xv[j] = ldv<T>(xr + c);
  float ss = 0.f;
  VM_ROW_LOOP(j, c) {
#pragma unroll
    for (int i = 0; i < V; ++i) { float f = Elem<T>::ld(xv[j][i]); ss += f * f; }
  }
  ss = wave_sum(ss);
  const float r = rsqrtf(ss / (float)cols + eps);
  if (lane == 0 && rstd_out) rstd_out[row] = r;
  VM_ROW_LOOP(j, c) {
    auto wv = ldv<T>(w + c);
    typename Elem<T>::vec_t o;
#pragma unroll
    for (int i = 0; i < V; ++i) o[i] = Elem<T>::st(Elem<T>::ld(wv[i]) * (Elem<T>::ld(xv[j][i]) * r));
    stv<T>(yr + c, o);
  }
}

template <typename T, int NV>
__global__ __launch_bounds__(ROW_THREADS) void rmsnorm_bwd_r_k(
    const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ dy,
    const float* __restrict__ rstd, T* __restrict__ dx, int rows, int cols, const int32_t* nrows_dev,
    const T* __restrict__ dx_add) {
  constexpr int V = Elem<T>::VEC;
  if (nrows_dev) rows = min(rows, *nrows_dev);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + (int64_t)row * cols;
  const T* dyr = dy + (int64_t)row * cols;
  const T* addr = dx_add ? dx_add + (int64_t)row * cols : nullptr;
  T* dxr = dx + (int64_t)row * cols;
  const float r = rstd[row];
  typename Elem<T>::vec_t xv[NV], gv[NV];
  VM_ROW_LOOP(j, c) { xv[j] = ldv<T>(xr + c); gv[j] = ldv<T>(dyr + c); }
  float dot = 0.f;
  VM_ROW_LOOP(j, c) {
    auto wv = ldv<T>(w + c);
#pragma unroll
    for (int i = 0; i < V; ++i) dot += Elem<T>::ld(wv[i]) * Elem<T>::ld(gv[j][i]) * (Elem<T>::ld(xv[j][i]) * r);
  }
  dot = wave_sum(dot) / (float)cols;
  VM_ROW_LOOP(j, c) {
    auto wv = ldv<T>(w + c);
    typename Elem<T>::vec_t o, av;
    if (addr) av = ldv<T>(addr + c);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float xh = Elem<T>::ld(xv[j][i]) * r;
      o[i] = Elem<T>::st(r * (Elem<T>::ld(wv[i]) * Elem<T>::ld(gv[j][i]) - xh * dot) + (addr ? Elem<T>::ld(av[i]) : 0.f));
    }
    stv<T>(dxr + c, o);
  }
}

template <typename T, int NV>
__global__ __launch_bounds__(ROW_THREADS) void layernorm_fwd_r_k(
    const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ b, const T* __restrict__ res,
    T* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out,
    int rows, int cols, float eps) {
  constexpr int V = Elem<T>::VEC;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + (int64_t)row * cols;
  typename Elem<T>::vec_t xv[NV];
  VM_ROW_LOOP(j, c) xv[j] = ldv<T>(xr + c);
  float s = 0.f;
  VM_ROW_LOOP(j, c) {
#pragma unroll
    for (int i = 0; i < V; ++i) s += Elem<T>::ld(xv[j][i]);
  }
  const float mu = wave_sum(s) / (float)cols;
  float ss = 0.f;
  VM_ROW_LOOP(j, c) {
#pragma unroll
    for (int i = 0; i < V; ++i) { const float d = Elem<T>::ld(xv[j][i]) - mu; ss += d * d; }
  }
  const float r = rsqrtf(wave_sum(ss) / (float)cols + eps);
  if (lane == 0) { if (mean_out) mean_out[row] = mu; if (rstd_out) rstd_out[row] = r; }
  T* yr = y + (int64_t)row * cols;
  const T* rr = res ? res + (int64_t)row * cols : nullptr;
  VM_ROW_LOOP(j, c) {
    typename Elem<T>::vec_t o;
    typename Elem<T>::vec_t wv, bv, rv;
    if (w) wv = ldv<T>(w + c);
    if (b) bv = ldv<T>(b + c);
    if (rr) rv = ldv<T>(rr + c);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      float f = (Elem<T>::ld(xv[j][i]) - mu) * r;
      if (w) f *= Elem<T>::ld(wv[i]);
      if (b) f += Elem<T>::ld(bv[i]);
      if (rr) f = Elem<T>::ld(Elem<T>::st(f)) + Elem<T>::ld(rv[i]);
      o[i] = Elem<T>::st(f);
    }
    stv<T>(yr + c, o);
  }
}

template <typename T, int NV>
__global__ __launch_bounds__(ROW_THREADS) void layernorm_bwd_r_k(
    const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ dy,
    const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dx, int rows, int cols,
    const T* __restrict__ dx_add) {
  constexpr int V = Elem<T>::VEC;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + (int64_t)row * cols;
  const T* dyr = dy + (int64_t)row * cols;
  const T* addr = dx_add ? dx_add + (int64_t)row * cols : nullptr;
  T* dxr = dx + (int64_t)row * cols;
  const float mu = mean[row], r = rstd[row];
  typename Elem<T>::vec_t xv[NV], gv[NV];
  VM_ROW_LOOP(j, c) { xv[j] = ldv<T>(xr + c); gv[j] = ldv<T>(dyr + c); }
  float s1 = 0.f, s2 = 0.f;
  VM_ROW_LOOP(j, c) {
    typename Elem<T>::vec_t wv; if (w) wv = ldv<T>(w + c);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float xh = (Elem<T>::ld(xv[j][i]) - mu) * r, g = Elem<T>::ld(gv[j][i]);
      const float wg = w ? Elem<T>::ld(wv[i]) * g : g;
      s1 += wg; s2 += wg * xh;
    }
  }
  s1 = wave_sum(s1) / (float)cols; s2 = wave_sum(s2) / (float)cols;
  VM_ROW_LOOP(j, c) {
    typename Elem<T>::vec_t wv; if (w) wv = ldv<T>(w + c);
    typename Elem<T>::vec_t o, av;
    if (addr) av = ldv<T>(addr + c);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float xh = (Elem<T>::ld(xv[j][i]) - mu) * r, g = Elem<T>::ld(gv[j][i]);
      const float wg = w ? Elem<T>::ld(wv[i]) * g : g;
      o[i] = Elem<T>::st(r * (wg - s1 - xh * s2) + (addr ? Elem<T>::ld(av[i]) : 0.f));
    }
    stv<T>(dxr + c, o);
  }
}
#undef VM_ROW_LOOP

// ---------------------------------------------------------------- RoPE
template <typename T>
__global__ __launch_bounds__(ROW_THREADS) void rope_k(
    T* __restrict__ qkv, int64_t ld, const int32_t* __restrict__ row_pos,
    const float* __restrict__ cos_tab, const float* __restrict__ sin_tab, int n_pos,
    int rows, int n_heads, int hd, int inverse, const int32_t* nrows_dev) {
  constexpr int V = Elem<T>::VEC;
  if (nrows_dev) rows = min(rows, *nrows_dev);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  int pos = row_pos[row];
  pos = max(0, min(pos, n_pos - 1));
  const float* cr = cos_tab + (int64_t)pos * hd;
  const float* sr = sin_tab + (int64_t)pos * hd;
  const int half = hd >> 1;
  const int cph = half / V;                    // chunks per half head
  const int items = 2 * n_heads * cph;         // q and k
  T* base = qkv + (int64_t)row * ld;
  // in place: the compiler must keep an iteration's loads behind the previous iteration's stores (same buffer), so a plain loop is one
  // memory round trip per item — 8 per lane at 32 heads x 128. Four items' operands are requested before the first is rotated.
  constexpr int U = 4;
  for (int it0 = lane; it0 < items; it0 += 64 * U) {
    typename Elem<T>::vec_t a[U], b[U];
    T* p1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int it = it0 + 64 * u;
      if (it >= items) continue;
      const int ch = it % cph;
      const int head = (it / cph) % n_heads;
      const int which = it / (cph * n_heads);    // 0 = q, 1 = k
      p1[u] = base + (int64_t)which * n_heads * hd + head * hd + ch * V;
      a[u] = ldv<T>(p1[u]); b[u] = ldv<T>(p1[u] + half);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int it = it0 + 64 * u;
      if (it >= items) continue;
      const int ch = it % cph;
      typename Elem<T>::vec_t oa, ob;
      // table holds cat(freqs, freqs): entry i and i+half are equal, but keep both lookups literal; 16-byte table loads
      float c1v[V], s1v[V], c2v[V], s2v[V];
#pragma unroll
      for (int i4 = 0; i4 < V; i4 += 4) {
        *reinterpret_cast<f32x4_t*>(c1v + i4) = *reinterpret_cast<const f32x4_t*>(cr + ch * V + i4);
        *reinterpret_cast<f32x4_t*>(s1v + i4) = *reinterpret_cast<const f32x4_t*>(sr + ch * V + i4);
        *reinterpret_cast<f32x4_t*>(c2v + i4) = *reinterpret_cast<const f32x4_t*>(cr + half + ch * V + i4);
        *reinterpret_cast<f32x4_t*>(s2v + i4) = *reinterpret_cast<const f32x4_t*>(sr + half + ch * V + i4);
      }
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const float c1 = c1v[i], s1 = s1v[i], c2 = c2v[i], s2 = s2v[i];
        const float x1 = Elem<T>::ld(a[u][i]), x2 = Elem<T>::ld(b[u][i]);
        if (!inverse) {
          // out = x*cos + rotate_half(x)*sin ; rotate_half = cat(-x2, x1)
          oa[i] = Elem<T>::st(x1 * c1 - x2 * s1);
          ob[i] = Elem<T>::st(x2 * c2 + x1 * s2);
        } else {
          // transposed map: dx1 = g1*c1 + g2*s2 ; dx2 = g2*c2 - g1*s1
          oa[i] = Elem<T>::st(x1 * c1 + x2 * s2);
          ob[i] = Elem<T>::st(x2 * c2 - x1 * s1);
        }
      }
      stv<T>(p1[u], oa); stv<T>(p1[u] + half, ob);
    }
  }
}

// ---------------------------------------------------------------- elementwise
enum { EW_SILU_MUL_F, EW_GELU_F, EW_GELU_B, EW_RELU_B, EW_ADD, EW_DROPOUT };

template <typename T, int OP>
__global__ __launch_bounds__(256) void ew_k(const T* __restrict__ a, const T* __restrict__ b,
                                            T* __restrict__ y, int64_t n, float p, uint64_t seed) {
  constexpr int V = Elem<T>::VEC;
  const int64_t nv = n / V;
  const float inv_keep = 1.0f / (1.0f - p);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    auto av = ldv<T>(a + i * V);
    typename Elem<T>::vec_t bv;
    if (OP == EW_SILU_MUL_F || OP == EW_GELU_B || OP == EW_RELU_B || OP == EW_ADD) bv = ldv<T>(b + i * V);
    typename Elem<T>::vec_t o;
    uint64_t h0 = 0, h1 = 0;
    if (OP == EW_DROPOUT) {   // V = 4 (f32): one hash group; V = 8 (bf16): two
      h0 = vm_hash4(seed, (uint64_t)(i * V) >> 2);
      if (V == 8) h1 = vm_hash4(seed, ((uint64_t)(i * V) >> 2) + 1);
    }
    const unsigned thr = vm_drop_threshold(p);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float x = Elem<T>::ld(av[j]);
      float r;
      if (OP == EW_SILU_MUL_F) {
        // reference rounds silu(gate) to the storage dtype before the product (modeling_cogvlm.py:55)
        const float s = Elem<T>::ld(Elem<T>::st(x / (1.0f + __expf(-x))));
        r = s * Elem<T>::ld(bv[j]);
      } else if (OP == EW_GELU_F) r = gelu_erf(x);
      else if (OP == EW_GELU_B) r = gelu_erf_grad(x) * Elem<T>::ld(bv[j]);
      else if (OP == EW_RELU_B) r = x > 0.f ? Elem<T>::ld(bv[j]) : 0.f;
      else if (OP == EW_ADD) r = x + Elem<T>::ld(bv[j]);
      else /* EW_DROPOUT */ r = vm_keep_bits(j < 4 ? h0 : h1, j & 3, thr) ? x * inv_keep : 0.f;
      o[j] = Elem<T>::st(r);
    }
    stv<T>(y + i * V, o);
  }
  // scalar tail
  const int64_t t0 = nv * V;
  for (int64_t i = t0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = Elem<T>::ld(a[i]);
    float r;
    if (OP == EW_SILU_MUL_F) r = Elem<T>::ld(Elem<T>::st(x / (1.0f + __expf(-x)))) * Elem<T>::ld(b[i]);
    else if (OP == EW_GELU_F) r = gelu_erf(x);
    else if (OP == EW_GELU_B) r = gelu_erf_grad(x) * Elem<T>::ld(b[i]);
    else if (OP == EW_RELU_B) r = x > 0.f ? Elem<T>::ld(b[i]) : 0.f;
    else if (OP == EW_ADD) r = x + Elem<T>::ld(b[i]);
    else r = vm_keep(seed, (uint64_t)i, p) ? x * inv_keep : 0.f;
    y[i] = Elem<T>::st(r);
  }
}

// ---------------------------------------------------------------- bf16 GELU / GELU' through a table
// erf-GELU on bf16 tensors is VALU-bound, not HBM-bound: erff is ~38 instructions per element (both branches of its range split
// execute in every wave) and the [6280 x 15360] activations of one ViT-E block took 84 us forward / 109 us backward in the step
// against a memory time of 60 / 89 us. A bf16 input has only 65 536 values: the 5120 with 2^-16 <= |x| < 2^4 (20 binades x 128
// mantissas x 2 signs) are filled ONCE per process by gelu_table_init_k with the very functions the arithmetic kernels use
// (gelu_erf / gelu_erf_grad), the rest have closed forms with the same bits (vm_common.hpp gelu_tab_fwd8 / gelu_tab_bwd8): the
// table kernels are bit-identical to ew_k on every input (tests/test_kernels_gpu.py walks all 65 536). Isolated 93 / 137 us ->
// 71 / 111 us (values inside the table), 84 / 115 us (9 % of |x| >= 16, the last blocks of the random-init benchmark model); step
// -2.1 ms. A full 65 536-entry table gathered from L1 / L2 instead of LDS + closed forms: 117 / 135 us (address-unit bound), removed.
// Tried on top and removed (profiles/r3_gelu_fusion.txt): the same look-ups inside the 256-column GEMM's epilogue, fc1 writing h and
// gelu(h), fc2's input-gradient GEMM multiplying by gelu'(h) — bit-identical, but the ~30 VALU operations per element run while the
// CU's matrix pipe idles (8-13 us per 256 x 256 tile), whereas the stand-alone pass finds its input in the Infinity Cache and
// costs 40 / 100 us: fused 392 / 536 us vs GEMM + kernel 396 / 530 us in isolation, +4.6 ms on the step.
__device__ unsigned short g_gelu_fwd_tab[GT_N];                   // bf16( gelu(x) )
__device__ float g_gelu_grad_tab[GT_N];                           // gelu'(x), fp32 as gelu_erf_grad returns it

__global__ void gelu_table_init_k() {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < GT_N; i += gridDim.x * blockDim.x) {
    const float x = bf2f((unsigned short)gt_bits_of(i));
    g_gelu_fwd_tab[i] = f2bf(gelu_erf(x));
    g_gelu_grad_tab[i] = gelu_erf_grad(x);
  }
}

template <bool BWD>
__global__ __launch_bounds__(512) void gelu_tab_k(const unsigned short* __restrict__ h, const unsigned short* __restrict__ dy,
                                                  unsigned short* __restrict__ y, int64_t n) {
  __shared__ __attribute__((aligned(16))) char tab_raw[BWD ? GT_N * 4 : GT_N * 2];
  {
    const f32x4_t* src = reinterpret_cast<const f32x4_t*>(BWD ? (const void*)g_gelu_grad_tab : (const void*)g_gelu_fwd_tab);
    f32x4_t* dst = reinterpret_cast<f32x4_t*>(tab_raw);
    for (int i = threadIdx.x; i < (int)sizeof(tab_raw) / 16; i += 512) dst[i] = src[i];
  }
  __syncthreads();
  const unsigned short* tf = reinterpret_cast<const unsigned short*>(tab_raw);
  const float* tg = reinterpret_cast<const float*>(tab_raw);
  const int64_t nv = n / 8;
  int64_t i = (int64_t)blockIdx.x * 512 + threadIdx.x;
  if constexpr (!BWD) {
    // the forward has ONE 16-byte load per lane and iteration: 24 waves per CU keep 24 KiB in flight where ~2 us of latency at the CU's share
    // of the memory rate wants ~45 (the backward, two loads per iteration, streams at 6.3 TB/s; this loop ran at 5.2). Two vectors per trip.
    const int64_t stride = (int64_t)gridDim.x * 512;
    for (; i + stride < nv; i += 2 * stride) {
      const u16x8_t h0 = ldv<unsigned short>(h + i * 8), h1 = ldv<unsigned short>(h + (i + stride) * 8);
      u16x8_t o0, o1;
      gelu_tab_fwd8(tf, h0, o0);
      gelu_tab_fwd8(tf, h1, o1);
      stv<unsigned short>(y + i * 8, o0);
      stv<unsigned short>(y + (i + stride) * 8, o1);
    }
  }
  for (; i < nv; i += (int64_t)gridDim.x * 512) {
    const u16x8_t hv = BWD ? ldv_nt<unsigned short>(h + i * 8) : ldv<unsigned short>(h + i * 8);
    u16x8_t dv;
    if (BWD) dv = ldv_nt<unsigned short>(dy + i * 8);
    u16x8_t o;
    if (BWD) gelu_tab_bwd8(tg, hv, dv, o);
    else gelu_tab_fwd8(tf, hv, o);
    if (BWD) stv_nt<unsigned short>(y + i * 8, o); else stv<unsigned short>(y + i * 8, o);
  }
  for (int64_t t = nv * 8 + (int64_t)blockIdx.x * 512 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 512) {
    const float x = bf2f(h[t]);
    y[t] = f2bf(BWD ? gelu_erf_grad(x) * bf2f(dy[t]) : gelu_erf(x));
  }
}

// fills the tables on first use (once per process; the wait makes them visible to every stream)
static int gelu_tables_ready(hipStream_t st) {
  static std::once_flag once;
  static int rc = VM_OK;
  std::call_once(once, [&] {
    hipLaunchKernelGGL(gelu_table_init_k, dim3(20), dim3(256), 0, st);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = VM_ERR_LAUNCH;
  });
  return rc;
}

template <typename T>
__global__ __launch_bounds__(256) void silu_mul_bwd_k(const T* __restrict__ g, const T* __restrict__ u,
                                                      const T* __restrict__ dout, T* __restrict__ dg,
                                                      T* __restrict__ du, int64_t n) {
  constexpr int V = Elem<T>::VEC;
  const int64_t nv = n / V;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    auto gv = ldv<T>(g + i * V); auto uv = ldv<T>(u + i * V); auto dv = ldv<T>(dout + i * V);
    typename Elem<T>::vec_t og, ou;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float x = Elem<T>::ld(gv[j]), uu = Elem<T>::ld(uv[j]), d = Elem<T>::ld(dv[j]);
      const float sig = 1.0f / (1.0f + __expf(-x));
      const float s = x * sig;
      ou[j] = Elem<T>::st(d * s);
      og[j] = Elem<T>::st(d * uu * (sig * (1.0f + x * (1.0f - sig))));
    }
    stv<T>(dg + i * V, og); stv<T>(du + i * V, ou);
  }
  const int64_t t0 = nv * V;
  for (int64_t i = t0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = Elem<T>::ld(g[i]), uu = Elem<T>::ld(u[i]), d = Elem<T>::ld(dout[i]);
    const float sig = 1.0f / (1.0f + __expf(-x));
    du[i] = Elem<T>::st(d * x * sig);
    dg[i] = Elem<T>::st(d * uu * (sig * (1.0f + x * (1.0f - sig))));
  }
}

template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_k(const S* __restrict__ x, D* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = Elem<D>::st(Elem<S>::ld(x[i]));
}

// ---------------------------------------------------------------- row gather / scatter
template <typename T, bool SCATTER>
__global__ __launch_bounds__(ROW_THREADS) void move_rows_k(
    const T* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ idx,
    T* __restrict__ out, int64_t ld_out, int rows, int cols, const int32_t* nrows_dev) {
  constexpr int V = Elem<T>::VEC;
  if (nrows_dev) rows = min(rows, *nrows_dev);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int j = idx[row];
  if (SCATTER) {
    if (j < 0) return;
    const T* s = src + (int64_t)row * ld_src;
    T* d = out + (int64_t)j * ld_out;
    for (int c = lane * V; c < cols; c += 64 * V) stv<T>(d + c, ldv<T>(s + c));
  } else {
    T* d = out + (int64_t)row * ld_out;
    if (j < 0) {
      typename Elem<T>::vec_t z;
#pragma unroll
      for (int i = 0; i < V; ++i) z[i] = Elem<T>::st(0.f);
      for (int c = lane * V; c < cols; c += 64 * V) stv<T>(d + c, z);
    } else {
      const T* s = src + (int64_t)j * ld_src;
      for (int c = lane * V; c < cols; c += 64 * V) stv<T>(d + c, ldv<T>(s + c));
    }
  }
}

// Embedding weight gradient: position p (in id-sorted order) that starts a segment sums the segment.
template <typename T>
__global__ __launch_bounds__(256) void embedding_bwd_k(
    const T* __restrict__ dout, int64_t ld, const int32_t* __restrict__ sorted_ids,
    const int32_t* __restrict__ sorted_rows, int n, T* __restrict__ dw, int64_t ld_w, int cols) {
  const int p = blockIdx.x;
  const int id = sorted_ids[p];
  if (id < 0) return;
  if (p > 0 && sorted_ids[p - 1] == id) return;  // not a segment head
  int e = p + 1;
  while (e < n && sorted_ids[e] == id) ++e;
  for (int c = threadIdx.x; c < cols; c += blockDim.x) {
    float acc = 0.f;
    for (int q = p; q < e; ++q) {
      const int r = sorted_rows[q];
      if (r >= 0) acc += Elem<T>::ld(dout[(int64_t)r * ld + c]);
    }
    dw[(int64_t)id * ld_w + c] = Elem<T>::st(acc);
  }
}

// ---------------------------------------------------------------- transpose (64x64 tiles through LDS)
template <typename T>
__global__ __launch_bounds__(256) void transpose_k(const T* __restrict__ in, int64_t ld_in,
                                                   T* __restrict__ out, int64_t ld_out,
                                                   int rows, int cols, const int32_t* nrows_dev,
                                                   const int32_t* range_dev, int segment, float* __restrict__ colsum = nullptr) {
  __shared__ T tile[64][64 + 2];
  __shared__ float csum[4][64];
  int rows_true = rows;
  if (nrows_dev) rows_true = min(rows, *nrows_dev);
  if (range_dev) {
    // rows [begin, end) of the token-routed layout: segment 0 = [0, counts[0]), 1 = [counts[0], counts[1])
    const int begin = segment ? range_dev[0] : 0;
    const int end = segment ? range_dev[1] : range_dev[0];
    in += (int64_t)begin * ld_in;
    rows_true = min(rows, max(end - begin, 0));
  }
  const int rows_out = (int)min(ld_out, (int64_t)((rows + 63) / 64) * 64);      // the output's pad columns up to the tile edge are zero-filled
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    T v = Elem<T>::st(0.f);
    if (r < rows_true && c < cols) v = in[(int64_t)r * ld_in + c];
    tile[i][tx] = v;
  }
  __syncthreads();
  if (colsum) {       // column sums of the input ride along (bias gradient = column sum of dy, which the weight gradient transposes anyway)
    float a = 0.f;
    for (int i = ty; i < 64; i += 4) a += Elem<T>::ld(tile[i][tx]);
    csum[ty][tx] = a;
  }
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows_out) out[(int64_t)c * ld_out + r] = tile[tx][i];      // (zeros beyond the true rows: the K padding)
  }
  if (colsum) {
    __syncthreads();
    if (ty == 0 && c0 + tx < cols) atomicAdd(colsum + c0 + tx, csum[0][tx] + csum[1][tx] + csum[2][tx] + csum[3][tx]);
  }
}

// fp32 fast path (the heads' weight gradients transpose dy and x of every unfrozen linear: 639 launches per step): 16-byte loads along the
// input rows, 16-byte stores along the output rows, the transposition in LDS. Needs ld_in, ld_out % 4 == 0, 16-byte aligned bases and
// an output pitch that holds whole 64-row tiles (callers pad rows to 64); same zero fill and optional column sums as transpose_k.
__global__ __launch_bounds__(256) void transpose_f32_vec_k(const float* __restrict__ in, int64_t ld_in, float* __restrict__ out, int64_t ld_out,
                                                           int rows, int cols, float* __restrict__ colsum) {
  __shared__ float tile[64][65];          // tile[c][r]
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int t = threadIdx.x, q = t & 15, p = t >> 4;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int r = r0 + p + 16 * it, c = c0 + 4 * q;
    f32x4_t v = {0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
      const float* src = in + (int64_t)r * ld_in + c;
      if (c + 3 < cols) v = *reinterpret_cast<const f32x4_t*>(src);
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (c + e < cols) v[e] = src[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[4 * q + e][p + 16 * it] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int oc = p + 16 * it;             // output row = input column
    if (c0 + oc < cols) {
      const f32x4_t v = {tile[oc][4 * q], tile[oc][4 * q + 1], tile[oc][4 * q + 2], tile[oc][4 * q + 3]};
      *reinterpret_cast<f32x4_t*>(out + (int64_t)(c0 + oc) * ld_out + r0 + 4 * q) = v;
    }
  }
  if (colsum && t < 64 && c0 + t < cols) {
    float a = 0.f;
#pragma unroll 8
    for (int r = 0; r < 64; ++r) a += tile[t][r];
    atomicAdd(colsum + c0 + t, a);
  }
}

// one launch for a whole table of small transposes (every LoRA factor of the model after an optimizer step):
// desc[i] = {src, dst, rows, cols, ld_src, ld_dst} as int64; blockIdx.y = table entry, blockIdx.x = 64x64 tile
// 2-byte entries whose rows, columns and pitches are multiples of 8 and whose bases are 16-byte aligned (every LoRA factor) take the
// vector path: 16-byte loads along the input rows, 16-byte stores along the output rows, the transposition in LDS (2 + 2 vector
// accesses per thread and tile instead of 16 + 16 two-byte ones: 1 152 factors = 604 MB moved per step, 0.74 -> ~3 TB/s).
template <typename T>
__global__ __launch_bounds__(256) void transpose_batched_k(const int64_t* __restrict__ desc) {
  constexpr int PITCH = sizeof(T) == 2 ? 72 : 66;          // 144-byte rows keep the 16-byte LDS writes of the vector path aligned
  __shared__ __attribute__((aligned(16))) T tile[64][PITCH];
  const int64_t* d = desc + (int64_t)blockIdx.y * 6;
  const T* in = reinterpret_cast<const T*>(d[0]);
  T* out = reinterpret_cast<T*>(d[1]);
  const int rows = (int)d[2], cols = (int)d[3];
  const int64_t ld_in = d[4], ld_out = d[5];
  const int tiles_c = (cols + 63) / 64, tiles_r = (rows + 63) / 64;
  const bool vec = sizeof(T) == 2 && ((rows | cols) & 7) == 0 && ((ld_in | ld_out) & 7) == 0 &&
                   ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  for (int t = blockIdx.x; t < tiles_c * tiles_r; t += gridDim.x) {
    const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
    if constexpr (sizeof(T) == 2) {
      if (vec) {
        typedef __attribute__((ext_vector_type(8))) unsigned short v8_t;
        const int p = threadIdx.x >> 3, q = threadIdx.x & 7;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int r = p + 32 * it;
          v8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
          if (r0 + r < rows && c0 + 8 * q < cols) v = *reinterpret_cast<const v8_t*>(in + (int64_t)(r0 + r) * ld_in + c0 + 8 * q);
          *reinterpret_cast<v8_t*>(&tile[r][8 * q]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int oc = p + 32 * it;                          // output row = input column
          if (c0 + oc < cols && r0 + 8 * q < rows) {
            v8_t v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = __builtin_bit_cast(unsigned short, tile[8 * q + e][oc]);
            *reinterpret_cast<v8_t*>(out + (int64_t)(c0 + oc) * ld_out + r0 + 8 * q) = v;
          }
        }
        __syncthreads();
        continue;
      }
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
      const int r = r0 + i, c = c0 + tx;
      T v = Elem<T>::st(0.f);
      if (r < rows && c < cols) v = in[(int64_t)r * ld_in + c];
      tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
      const int c = c0 + i, r = r0 + tx;
      if (c < cols && r < rows) out[(int64_t)c * ld_out + r] = tile[tx][i];
    }
    __syncthreads();
  }
}


// ---------------------------------------------------------------- per-row fp8 (OCP e4m3) quantisation
// x8[r][c] = e4m3(x[r][c] * 448 / amax_r), scale[r] = amax_r / 448 (1 for an all-zero row), inv_scale[r] = 1 / scale[r]: the operands of
// vm_gemm_fp8 (activations per token row; frozen weights per output channel, once). One wave per row, two passes over the row
// (the second one hits L2), 16 elements per lane and step.
template <typename T>
__global__ __launch_bounds__(ROW_THREADS) void quant_rows_fp8_k(const T* __restrict__ x, int64_t ldx, unsigned char* __restrict__ x8, int64_t ld8,
                                                                float* __restrict__ scale, float* __restrict__ inv_scale, int rows, int cols,
                                                                const int32_t* nrows_dev) {
  int rows_true = rows;
  if (nrows_dev) rows_true = min(rows, *nrows_dev);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + (int64_t)row * ldx;
  unsigned char* o = x8 + (int64_t)row * ld8;
  if (row >= rows_true) {           // rows beyond the device-side count: zeros, unit scale
    for (int c = lane * 16; c < cols; c += 64 * 16) *reinterpret_cast<i32x4_t*>(o + c) = (i32x4_t){0, 0, 0, 0};
    if (lane == 0) { scale[row] = 1.f; if (inv_scale) inv_scale[row] = 1.f; }
    return;
  }
  float amax = 0.f;
  for (int c = lane * 16; c < cols; c += 64 * 16) {
#pragma unroll
    for (int h = 0; h < 16; ++h) amax = fmaxf(amax, fabsf(Elem<T>::ld(xr[c + h])));
  }
  amax = wave_max(amax);
  const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.f;
  const float inv = amax > 0.f ? 448.0f / amax : 1.f;
  if (lane == 0) { scale[row] = sc; if (inv_scale) inv_scale[row] = 1.0f / sc; }
  for (int c = lane * 16; c < cols; c += 64 * 16) {
    i32x4_t pk;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float f0 = Elem<T>::ld(xr[c + 4 * q]) * inv, f1 = Elem<T>::ld(xr[c + 4 * q + 1]) * inv;
      const float f2 = Elem<T>::ld(xr[c + 4 * q + 2]) * inv, f3 = Elem<T>::ld(xr[c + 4 * q + 3]) * inv;
      int w = __builtin_amdgcn_cvt_pk_fp8_f32(f0, f1, 0, false);
      w = __builtin_amdgcn_cvt_pk_fp8_f32(f2, f3, w, true);
      pk[q] = w;
    }
    *reinterpret_cast<i32x4_t*>(o + c) = pk;
  }
}

// out[r][c] = x[r][c] * s[r] (bf16 in / out, fp32 factor): pre-division of the LoRA extension operands of vm_gemm_fp8 by the fp8 scales
__global__ __launch_bounds__(256) void scale_rows_bf16_k(const unsigned short* __restrict__ x, int64_t ldx, const float* __restrict__ s,
                                                         unsigned short* __restrict__ out, int64_t ldo, int rows, int cols) {
  const int groups = cols / 8;
  const int64_t total = (int64_t)rows * groups;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / groups), c = (int)(i % groups) * 8;
    const float f = s[r];
    u16x8_t v = *reinterpret_cast<const u16x8_t*>(x + (int64_t)r * ldx + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(v[e]) * f);
    *reinterpret_cast<u16x8_t*>(out + (int64_t)r * ldo + c) = v;
  }
}

// ---------------------------------------------------------------- fp32 side accumulators -> bf16 gradient slots
// desc[i] = {dst (bf16*), src (float*), count}: dst[j] = bf16(float(dst[j]) + float(bf16(src[j]))), src[j] = 0 — the rounding of
// AccumulateGrad's `grad += g.to(bf16)`. One launch per gradient bucket moves every column-sum gradient of the norm layers
// (atomically accumulated in fp32 by norm_bwd_dwdb_k) into its bf16 slot and leaves the accumulator zeroed for the next step.
__global__ __launch_bounds__(256) void accum_f32_table_k(const int64_t* __restrict__ desc) {
  const int64_t* d = desc + (int64_t)blockIdx.y * 3;
  unsigned short* dst = reinterpret_cast<unsigned short*>(d[0]);
  float* src = reinterpret_cast<float*>(d[1]);
  const int n = (int)d[2];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float g = bf2f(f2bf(src[i]));
    dst[i] = f2bf(bf2f(dst[i]) + g);
    src[i] = 0.f;
  }
}

// ---------------------------------------------------------------- fused gradient clip + AdamW over a flat bucket
// One pass over a flat parameter / gradient / moment bucket (torch.optim.AdamW semantics, decoupled weight decay, maths in
// fp32, states stored in the parameter dtype): g' = g * clip_coef[0] (device scalar: no host round trip for the norm),
// p *= 1 - lr*wd; m = b1 m + (1-b1) g'; v = b2 v + (1-b2) g'^2; p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps).
template <typename T>
__global__ __launch_bounds__(256) void adamw_k(T* __restrict__ p, const T* __restrict__ g, T* __restrict__ m, T* __restrict__ v,
                                               int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1,
                                               float rsqrt_bc2, const float* __restrict__ clip_coef) {
  constexpr int V = Elem<T>::VEC;
  const float coef = clip_coef ? clip_coef[0] : 1.0f;
  const float step = lr / bc1, decay = 1.0f - lr * wd;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V; i < n; i += (int64_t)gridDim.x * 256 * V) {
    auto pv = ldv<T>(p + i); auto gv = ldv<T>(g + i); auto mv = ldv<T>(m + i); auto vv = ldv<T>(v + i);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float gg = Elem<T>::ld(gv[e]) * coef;
      float pp = Elem<T>::ld(pv[e]) * decay;
      const float mm = b1 * Elem<T>::ld(mv[e]) + (1.0f - b1) * gg;
      const float ww = b2 * Elem<T>::ld(vv[e]) + (1.0f - b2) * gg * gg;
      pp -= step * mm / (sqrtf(ww) * rsqrt_bc2 + eps);
      pv[e] = Elem<T>::st(pp); mv[e] = Elem<T>::st(mm); vv[e] = Elem<T>::st(ww);
    }
    stv<T>(p + i, pv); stv<T>(m + i, mv); stv<T>(v + i, vv);
  }
}

// ---------------------------------------------------------------- sum of squares (gradient norm): one partial per workgroup
template <typename T>
__global__ __launch_bounds__(256) void sumsq_k(const T* __restrict__ x, int64_t n, float* __restrict__ partials) {
  constexpr int V = Elem<T>::VEC;
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V; i < n; i += (int64_t)gridDim.x * 256 * V) {
    auto v = ldv<T>(x + i);
#pragma unroll
    for (int e = 0; e < V; ++e) { const float f = Elem<T>::ld(v[e]); acc = __builtin_fmaf(f, f, acc); }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------- column sums (bias gradients)
template <typename T>
__global__ __launch_bounds__(256) void colsum_k(const T* __restrict__ x, int64_t ld, float* __restrict__ out,
                                                int rows, int cols, const int32_t* nrows_dev) {
  if (nrows_dev) rows = min(rows, *nrows_dev);
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  float acc = 0.f;
  for (int r = blockIdx.y; r < rows; r += gridDim.y) acc += Elem<T>::ld(x[(int64_t)r * ld + c]);
  atomicAdd(out + c, acc);
}

// ---------------------------------------------------------------- cross entropy
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_k(const T* __restrict__ logits, int64_t ld,
                                                const int64_t* __restrict__ labels,
                                                float* __restrict__ row_loss, float* __restrict__ lse_out,
                                                int rows, int vocab, const int32_t* nrows_dev) {
  constexpr int V = Elem<T>::VEC;
  __shared__ float red[16];
  if (nrows_dev) rows = min(rows, *nrows_dev);
  const int row = blockIdx.x;
  if (row >= rows) return;
  const T* lr = logits + (int64_t)row * ld;
  const int nv = vocab / V;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < nv; i += blockDim.x) {
    auto v = ldv<T>(lr + i * V);
#pragma unroll
    for (int j = 0; j < V; ++j) m = fmaxf(m, Elem<T>::ld(v[j]));
  }
  for (int i = nv * V + threadIdx.x; i < vocab; i += blockDim.x) m = fmaxf(m, Elem<T>::ld(lr[i]));
  m = block_max(m, red);
  float s = 0.f;
  for (int i = threadIdx.x; i < nv; i += blockDim.x) {
    auto v = ldv<T>(lr + i * V);
#pragma unroll
    for (int j = 0; j < V; ++j) s += __expf(Elem<T>::ld(v[j]) - m);
  }
  for (int i = nv * V + threadIdx.x; i < vocab; i += blockDim.x) s += __expf(Elem<T>::ld(lr[i]) - m);
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    const float lse = m + __logf(s);
    lse_out[row] = lse;
    const int64_t lab = labels[row];
    row_loss[row] = (lab >= 0 && lab < vocab) ? lse - Elem<T>::ld(lr[lab]) : 0.f;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_k(const T* __restrict__ logits, int64_t ld,
                                                const int64_t* __restrict__ labels,
                                                const float* __restrict__ lse, const float* __restrict__ row_scale,
                                                T* __restrict__ dlogits, int64_t ld_d,
                                                int rows, int vocab, const int32_t* nrows_dev) {
  constexpr int V = Elem<T>::VEC;
  int rows_true = rows;
  if (nrows_dev) rows_true = min(rows, *nrows_dev);
  const int row = blockIdx.x;
  if (row >= rows) return;
  T* dr = dlogits + (int64_t)row * ld_d;
  const int nvd = (int)(ld_d / V);
  if (row >= rows_true) {  // keep padded rows exactly zero: they are contracted over by the lm_head wgrad
    typename Elem<T>::vec_t z;
#pragma unroll
    for (int j = 0; j < V; ++j) z[j] = Elem<T>::st(0.f);
    for (int i = threadIdx.x; i < nvd; i += blockDim.x) stv<T>(dr + i * V, z);
    return;
  }
  const T* lr = logits + (int64_t)row * ld;
  const float l = lse[row];
  const float sc = row_scale[row];
  const int64_t lab = labels[row];
  const int nv = vocab / V;
  for (int i = threadIdx.x; i < nvd; i += blockDim.x) {
    typename Elem<T>::vec_t o;
    if (i < nv && sc != 0.f) {
      auto v = ldv<T>(lr + i * V);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float p = __expf(Elem<T>::ld(v[j]) - l);
        if ((int64_t)(i * V + j) == lab) p -= 1.0f;
        o[j] = Elem<T>::st(p * sc);
      }
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const int c = i * V + j;
        float p = 0.f;
        if (c < vocab && sc != 0.f) { p = __expf(Elem<T>::ld(lr[c]) - l); if (c == lab) p -= 1.0f; p *= sc; }
        o[j] = Elem<T>::st(p);
      }
    }
    stv<T>(dr + i * V, o);
  }
}

// ---------------------------------------------------------------- im2col (patch embedding)
template <typename T>
__global__ __launch_bounds__(256) void im2col3d_k(const T* __restrict__ img, int C, int D, int H, int W,
                                                  int pz, int py, int px, T* __restrict__ cols, int64_t ld) {
  const int gd = D / pz, gh = H / py, gw = W / px;
  const int patch = blockIdx.x;
  if (patch >= gd * gh * gw) return;
  const int pd = patch / (gh * gw), ph = (patch / gw) % gh, pw = patch % gw;
  const int K = C * pz * py * px;
  T* out = cols + (int64_t)patch * ld;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const int x = k % px, y = (k / px) % py, z = (k / (px * py)) % pz, c = k / (px * py * pz);
    out[k] = img[(((int64_t)c * D + pd * pz + z) * H + ph * py + y) * W + pw * px + x];
  }
}

inline int ew_grid(int64_t n, int vec) {
  int64_t b = (n / vec + 255) / 256;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}
inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

#define DISPATCH_DTYPE(dtype, ...)                                    \
  if ((dtype) == VM_BF16) { typedef unsigned short T; __VA_ARGS__; }  \
  else if ((dtype) == VM_F32) { typedef float T; __VA_ARGS__; }       \
  else return VM_ERR_BAD_ARG;

// rows that fit NV = 4 / 8 vectors per lane take the register form (internal switch for A/B runs and the equality tests)
static int& norm_regs_on() { static int v = 1; return v; }
extern "C" int vm_norm_register_rows_(int on) { norm_regs_on() = on ? 1 : 0; return VM_OK; }
static inline int norm_nv(int cols, int vec) {
  if (!norm_regs_on()) return 0;
  const int n = (cols + 64 * vec - 1) / (64 * vec);
  return n <= 4 ? 4 : (n <= 8 ? 8 : 0);
}

// bf16 tensors of >= 1 M elements go through the table kernels
static int& gelu_table_on() { static int on = 1; return on; }
extern "C" int vm_gelu_table_(int on) { gelu_table_on() = on ? 1 : 0; return VM_OK; }      // internal (tools/bench_gelu.py): 0 = always the arithmetic kernel
static bool gelu_use_table(int64_t n, int dtype) { return gelu_table_on() && dtype == VM_BF16 && n >= (1 << 20); }
template <bool BWD>
static int gelu_tab_launch(const void* x, const void* dy, void* y, int64_t n, void* stream) {
  if (!aligned16(x) || !aligned16(y) || (BWD && !aligned16(dy))) return VM_ERR_BAD_ARG;
  if (int rc = gelu_tables_ready((hipStream_t)stream)) return rc;
  const int64_t wgs = (n / 8 + 511) / 512;
  hipLaunchKernelGGL((gelu_tab_k<BWD>), dim3((unsigned)std::min<int64_t>(wgs, 768)), dim3(512), 0, (hipStream_t)stream,
                     (const unsigned short*)x, (const unsigned short*)dy, (unsigned short*)y, n);
  VM_LAUNCH_CHECK();
  return VM_OK;
}

extern "C" {

int vm_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int rows, int cols, float eps,
                   int dtype, const int32_t* nrows_dev, void* stream) {
  if (rows <= 0) return VM_OK;
  const int vec = dtype == VM_BF16 ? 8 : 4;
  if (cols % vec) return VM_ERR_BAD_ARG;
  dim3 grid((rows + ROW_WAVES - 1) / ROW_WAVES);
  const int nv = norm_nv(cols, vec);
#define VM_RMS_FWD(K) hipLaunchKernelGGL(K, grid, dim3(ROW_THREADS), 0, (hipStream_t)stream, (const T*)x, (const T*)w, (T*)y, rstd, rows, cols, eps, nrows_dev)
  DISPATCH_DTYPE(dtype, if (nv == 4) VM_RMS_FWD((rmsnorm_fwd_r_k<T, 4>)); else if (nv == 8) VM_RMS_FWD((rmsnorm_fwd_r_k<T, 8>)); else VM_RMS_FWD(rmsnorm_fwd_k<T>));
#undef VM_RMS_FWD
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_rmsnorm_bwd(const void* x, const void* w, const void* dy, const float* rstd, void* dx, float* dw_accum,
                   int rows, int cols, int dtype, const int32_t* nrows_dev, void* stream) {
  return vm_rmsnorm_bwd_res(x, w, dy, rstd, nullptr, dx, dw_accum, rows, cols, dtype, nrows_dev, stream);
}

int vm_rmsnorm_bwd_res(const void* x, const void* w, const void* dy, const float* rstd, const void* dx_add, void* dx, float* dw_accum,
                       int rows, int cols, int dtype, const int32_t* nrows_dev, void* stream) {
  if (rows <= 0) return VM_OK;
  if (dx_add && !dx) return VM_ERR_BAD_ARG;
  const int vec = dtype == VM_BF16 ? 8 : 4;
  if (cols % vec) return VM_ERR_BAD_ARG;
  dim3 grid((rows + ROW_WAVES - 1) / ROW_WAVES);
  dim3 gridw((cols + 255) / 256, (rows + 31) / 32);
  const int nv = norm_nv(cols, vec);
#define VM_RMS_BWD(K) hipLaunchKernelGGL(K, grid, dim3(ROW_THREADS), 0, (hipStream_t)stream, (const T*)x, (const T*)w, (const T*)dy, rstd, (T*)dx, rows, cols, nrows_dev, (const T*)dx_add)
  DISPATCH_DTYPE(dtype,
                 if (dx) {
                   if (nv == 4) VM_RMS_BWD((rmsnorm_bwd_r_k<T, 4>)); else if (nv == 8) VM_RMS_BWD((rmsnorm_bwd_r_k<T, 8>)); else VM_RMS_BWD(rmsnorm_bwd_k<T>);
                 }
                 if (dw_accum) hipLaunchKernelGGL(norm_bwd_dwdb_k<T>, gridw, dim3(256), 0, (hipStream_t)stream, (const T*)x,
                                                  (const T*)dy, (const float*)nullptr, rstd, dw_accum, (float*)nullptr, rows, cols,
                                                  nrows_dev));
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_layernorm_fwd(const void* x, const void* w, const void* b, const void* residual, void* y, float* mean,
                     float* rstd, int rows, int cols, float eps, int dtype, void* stream) {
  if (rows <= 0) return VM_OK;
  const int vec = dtype == VM_BF16 ? 8 : 4;
  if (cols % vec) return VM_ERR_BAD_ARG;
  dim3 grid((rows + ROW_WAVES - 1) / ROW_WAVES);
  const int nv = norm_nv(cols, vec);
#define VM_LN_FWD(K) hipLaunchKernelGGL(K, grid, dim3(ROW_THREADS), 0, (hipStream_t)stream, (const T*)x, (const T*)w, (const T*)b, (const T*)residual, (T*)y, mean, rstd, rows, cols, eps)
  DISPATCH_DTYPE(dtype, if (nv == 4) VM_LN_FWD((layernorm_fwd_r_k<T, 4>)); else if (nv == 8) VM_LN_FWD((layernorm_fwd_r_k<T, 8>)); else VM_LN_FWD(layernorm_fwd_k<T>));
#undef VM_LN_FWD
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_layernorm_bwd(const void* x, const void* w, const void* dy, const float* mean, const float* rstd, void* dx,
                     float* dw_accum, float* db_accum, int rows, int cols, int dtype, void* stream) {
  return vm_layernorm_bwd_res(x, w, dy, mean, rstd, nullptr, dx, dw_accum, db_accum, rows, cols, dtype, stream);
}

int vm_layernorm_bwd_res(const void* x, const void* w, const void* dy, const float* mean, const float* rstd, const void* dx_add, void* dx,
                         float* dw_accum, float* db_accum, int rows, int cols, int dtype, void* stream) {
  if (rows <= 0) return VM_OK;
  if (dx_add && !dx) return VM_ERR_BAD_ARG;
  const int vec = dtype == VM_BF16 ? 8 : 4;
  if (cols % vec) return VM_ERR_BAD_ARG;
  dim3 grid((rows + ROW_WAVES - 1) / ROW_WAVES);
  dim3 gridw((cols + 255) / 256, (rows + 31) / 32);
  const int nv = norm_nv(cols, vec);
#define VM_LN_BWD(K) hipLaunchKernelGGL(K, grid, dim3(ROW_THREADS), 0, (hipStream_t)stream, (const T*)x, (const T*)w, (const T*)dy, mean, rstd, (T*)dx, rows, cols, (const T*)dx_add)
  DISPATCH_DTYPE(dtype,
                 if (dx) {
                   if (nv == 4) VM_LN_BWD((layernorm_bwd_r_k<T, 4>)); else if (nv == 8) VM_LN_BWD((layernorm_bwd_r_k<T, 8>)); else VM_LN_BWD(layernorm_bwd_k<T>);
                 }
                 if (dw_accum || db_accum) hipLaunchKernelGGL(norm_bwd_dwdb_k<T>, gridw, dim3(256), 0, (hipStream_t)stream,
                                                              (const T*)x, (const T*)dy, mean, rstd, dw_accum, db_accum, rows, cols,
                                                              (const int32_t*)nullptr));
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_rope_inplace(void* qkv, int64_t ld, const int32_t* row_pos, const float* cos_tab, const float* sin_tab,
                    int n_pos, int rows, int n_heads, int head_dim, int dtype, int inverse,
                    const int32_t* nrows_dev, void* stream) {
  if (rows <= 0) return VM_OK;
  const int vec = dtype == VM_BF16 ? 8 : 4;
  if ((head_dim / 2) % vec || ld % vec) return VM_ERR_BAD_ARG;
  dim3 grid((rows + ROW_WAVES - 1) / ROW_WAVES);
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(rope_k<T>, grid, dim3(ROW_THREADS), 0, (hipStream_t)stream, (T*)qkv, ld,
                                           row_pos, cos_tab, sin_tab, n_pos, rows, n_heads, head_dim, inverse,
                                           nrows_dev));
  VM_LAUNCH_CHECK();
  return VM_OK;
}

#define EW_LAUNCH(OP, a, b, y, n, p, seed)                                                                       \
  {                                                                                                              \
    if (n <= 0) return VM_OK;                                                                                    \
    if (!aligned16(a) || !aligned16(y) || ((b) && !aligned16(b))) return VM_ERR_BAD_ARG;                         \
    const int vec = dtype == VM_BF16 ? 8 : 4;                                                                    \
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ew_k<T, OP>), dim3(ew_grid(n, vec)), dim3(256), 0,                 \
                                             (hipStream_t)stream, (const T*)(a), (const T*)(b), (T*)(y), n, p, seed)); \
    VM_LAUNCH_CHECK();                                                                                           \
    return VM_OK;                                                                                                \
  }

int vm_silu_mul_fwd(const void* gate, const void* up, void* out, int64_t n, int dtype, void* stream)
  EW_LAUNCH(EW_SILU_MUL_F, gate, up, out, n, 0.f, 0ull)
int vm_gelu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream) {
  if (gelu_use_table(n, dtype)) return gelu_tab_launch<false>(x, nullptr, y, n, stream);
  EW_LAUNCH(EW_GELU_F, x, (const void*)nullptr, y, n, 0.f, 0ull)
}
int vm_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, int dtype, void* stream) {
  if (gelu_use_table(n, dtype)) return gelu_tab_launch<true>(x, dy, dx, n, stream);
  EW_LAUNCH(EW_GELU_B, x, dy, dx, n, 0.f, 0ull)
}
int vm_relu_bwd(const void* yv, const void* dy, void* dx, int64_t n, int dtype, void* stream)
  EW_LAUNCH(EW_RELU_B, yv, dy, dx, n, 0.f, 0ull)
int vm_add(const void* a, const void* b, void* y, int64_t n, int dtype, void* stream)
  EW_LAUNCH(EW_ADD, a, b, y, n, 0.f, 0ull)
int vm_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int dtype, void* stream) {
  if (p < 0.f || p >= 1.f) return VM_ERR_BAD_ARG;
  EW_LAUNCH(EW_DROPOUT, x, (const void*)nullptr, y, n, p, seed)
}

int vm_silu_mul_bwd(const void* gate, const void* up, const void* dout, void* dgate, void* dup, int64_t n,
                    int dtype, void* stream) {
  if (n <= 0) return VM_OK;
  const int vec = dtype == VM_BF16 ? 8 : 4;
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(silu_mul_bwd_k<T>, dim3(ew_grid(n, vec)), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)gate, (const T*)up, (const T*)dout, (T*)dgate, (T*)dup, n));
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_cast(const void* x, int src_dtype, void* y, int dst_dtype, int64_t n, void* stream) {
  if (n <= 0) return VM_OK;
  dim3 grid(ew_grid(n, 1));
  hipStream_t s = (hipStream_t)stream;
  if (src_dtype == VM_BF16 && dst_dtype == VM_F32)
    hipLaunchKernelGGL((cast_k<unsigned short, float>), grid, dim3(256), 0, s, (const unsigned short*)x, (float*)y, n);
  else if (src_dtype == VM_F32 && dst_dtype == VM_BF16)
    hipLaunchKernelGGL((cast_k<float, unsigned short>), grid, dim3(256), 0, s, (const float*)x, (unsigned short*)y, n);
  else if (src_dtype == VM_F32 && dst_dtype == VM_F32)
    hipLaunchKernelGGL((cast_k<float, float>), grid, dim3(256), 0, s, (const float*)x, (float*)y, n);
  else if (src_dtype == VM_BF16 && dst_dtype == VM_BF16)
    hipLaunchKernelGGL((cast_k<unsigned short, unsigned short>), grid, dim3(256), 0, s, (const unsigned short*)x,
                       (unsigned short*)y, n);
  else return VM_ERR_BAD_ARG;
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_gather_rows(const void* src, int64_t ld_src, const int32_t* idx, void* out, int64_t ld_out, int rows,
                   int cols, int dtype, const int32_t* nrows_dev, void* stream) {
  if (rows <= 0) return VM_OK;
  const int vec = dtype == VM_BF16 ? 8 : 4;
  if (cols % vec || ld_src % vec || ld_out % vec) return VM_ERR_BAD_ARG;
  dim3 grid((rows + ROW_WAVES - 1) / ROW_WAVES);
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((move_rows_k<T, false>), grid, dim3(ROW_THREADS), 0, (hipStream_t)stream,
                                           (const T*)src, ld_src, idx, (T*)out, ld_out, rows, cols, nrows_dev));
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_scatter_rows(const void* src, int64_t ld_src, const int32_t* idx, void* out, int64_t ld_out, int rows,
                    int cols, int dtype, const int32_t* nrows_dev, void* stream) {
  if (rows <= 0) return VM_OK;
  const int vec = dtype == VM_BF16 ? 8 : 4;
  if (cols % vec || ld_src % vec || ld_out % vec) return VM_ERR_BAD_ARG;
  dim3 grid((rows + ROW_WAVES - 1) / ROW_WAVES);
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((move_rows_k<T, true>), grid, dim3(ROW_THREADS), 0, (hipStream_t)stream,
                                           (const T*)src, ld_src, idx, (T*)out, ld_out, rows, cols, nrows_dev));
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_embedding_bwd(const void* dout, int64_t ld, const int32_t* sorted_ids, const int32_t* sorted_rows, int n,
                     void* dweight, int64_t ld_w, int cols, int dtype, void* stream) {
  if (n <= 0) return VM_OK;
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(embedding_bwd_k<T>, dim3(n), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)dout, ld, sorted_ids, sorted_rows, n, (T*)dweight, ld_w, cols));
  VM_LAUNCH_CHECK();
  return VM_OK;
}

static bool transpose_vec_ok(const void* in, int64_t ld_in, const void* out, int64_t ld_out, int rows) {
  return ld_in % 4 == 0 && ld_out % 4 == 0 && aligned16(in) && aligned16(out) && ld_out >= (int64_t)((rows + 63) / 64) * 64;
}

int vm_transpose(const void* in, int64_t ld_in, void* out, int64_t ld_out, int rows, int cols, int dtype,
                 const int32_t* nrows_dev, void* stream) {
  if (rows <= 0 || cols <= 0) return VM_OK;
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  if (dtype == VM_F32 && !nrows_dev && transpose_vec_ok(in, ld_in, out, ld_out, rows)) {
    hipLaunchKernelGGL(transpose_f32_vec_k, grid, dim3(256), 0, (hipStream_t)stream, (const float*)in, ld_in, (float*)out, ld_out, rows, cols,
                       (float*)nullptr);
    VM_LAUNCH_CHECK();
    return VM_OK;
  }
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(transpose_k<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)in,
                                           ld_in, (T*)out, ld_out, rows, cols, nrows_dev, (const int32_t*)nullptr, 0));
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_transpose_colsum(const void* in, int64_t ld_in, void* out, int64_t ld_out, int rows, int cols, int dtype, float* colsum_accum,
                        void* stream) {
  if (rows <= 0 || cols <= 0) return VM_OK;
  if (!colsum_accum) return VM_ERR_BAD_ARG;
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  if (dtype == VM_F32 && transpose_vec_ok(in, ld_in, out, ld_out, rows)) {
    hipLaunchKernelGGL(transpose_f32_vec_k, grid, dim3(256), 0, (hipStream_t)stream, (const float*)in, ld_in, (float*)out, ld_out, rows, cols,
                       colsum_accum);
    VM_LAUNCH_CHECK();
    return VM_OK;
  }
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(transpose_k<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)in,
                                           ld_in, (T*)out, ld_out, rows, cols, (const int32_t*)nullptr, (const int32_t*)nullptr, 0, colsum_accum));
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_transpose_batched(const int64_t* desc_dev, int n, int tiles_per_entry, int dtype, void* stream) {
  if (n <= 0) return VM_OK;
  if (!desc_dev || tiles_per_entry <= 0 || n > 65535) return VM_ERR_BAD_ARG;
  dim3 grid(tiles_per_entry, n);
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(transpose_batched_k<T>, grid, dim3(256), 0, (hipStream_t)stream, desc_dev));
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_quant_rows_fp8(const void* x, int64_t ldx, void* x8, int64_t ld8, float* scale, float* inv_scale, int rows, int cols, int dtype,
                      const int32_t* nrows_dev, void* stream) {
  if (rows <= 0 || cols <= 0) return VM_OK;
  if (!x || !x8 || !scale || cols % 16 || ld8 % 16 || ldx % 8 || !aligned16(x) || !aligned16(x8)) return VM_ERR_BAD_ARG;
  dim3 grid((rows + ROW_WAVES - 1) / ROW_WAVES);
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(quant_rows_fp8_k<T>, grid, dim3(ROW_THREADS), 0, (hipStream_t)stream, (const T*)x, ldx,
                                           (unsigned char*)x8, ld8, scale, inv_scale, rows, cols, nrows_dev));
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_scale_rows_bf16(const void* x, int64_t ldx, const float* s, void* out, int64_t ldo, int rows, int cols, void* stream) {
  if (rows <= 0 || cols <= 0) return VM_OK;
  if (!x || !s || !out || cols % 8 || ldx % 8 || ldo % 8 || !aligned16(x) || !aligned16(out)) return VM_ERR_BAD_ARG;
  const int64_t total = (int64_t)rows * (cols / 8);
  hipLaunchKernelGGL(scale_rows_bf16_k, dim3((unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)x, ldx, s, (unsigned short*)out, ldo, rows, cols);
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_accum_f32_table(const int64_t* desc_dev, int n, int blocks_per_entry, void* stream) {
  if (n <= 0) return VM_OK;
  if (!desc_dev || blocks_per_entry <= 0 || n > 65535) return VM_ERR_BAD_ARG;
  hipLaunchKernelGGL(accum_f32_table_k, dim3(blocks_per_entry, n), dim3(256), 0, (hipStream_t)stream, desc_dev);
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_transpose_segment(const void* in, int64_t ld_in, void* out, int64_t ld_out, int rows, int cols, int dtype,
                         const int32_t* counts_dev, int segment, void* stream) {
  if (rows <= 0 || cols <= 0) return VM_OK;
  if (!counts_dev || (segment != 0 && segment != 1)) return VM_ERR_BAD_ARG;
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(transpose_k<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)in,
                                           ld_in, (T*)out, ld_out, rows, cols, (const int32_t*)nullptr, counts_dev, segment));
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_adamw(void* p, const void* g, void* m, void* v, int64_t n, float lr, float beta1, float beta2, float eps,
             float weight_decay, int step, const float* clip_coef_dev, int dtype, void* stream) {
  if (n <= 0) return VM_OK;
  if (!p || !g || !m || !v || step < 1) return VM_ERR_BAD_ARG;
  const int vec = dtype == VM_BF16 ? 8 : 4;
  if (n % vec || !aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v)) return VM_ERR_BAD_ARG;
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float rsqrt_bc2 = 1.0f / sqrtf(1.0f - powf(beta2, (float)step));
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(adamw_k<T>, dim3(ew_grid(n, vec)), dim3(256), 0, (hipStream_t)stream, (T*)p, (const T*)g,
                                           (T*)m, (T*)v, n, lr, beta1, beta2, eps, weight_decay, bc1, rsqrt_bc2, clip_coef_dev));
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_sumsq_partials(const void* x, int64_t n, int dtype, float* partials_out, int n_partials, void* stream) {
  if (!x || !partials_out || n < 0 || n_partials <= 0) return VM_ERR_BAD_ARG;
  const int vec = dtype == VM_BF16 ? 8 : 4;
  if (n % vec || !aligned16(x)) return VM_ERR_BAD_ARG;
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(sumsq_k<T>, dim3(n_partials), dim3(256), 0, (hipStream_t)stream, (const T*)x, n, partials_out));
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_colsum(const void* x, int64_t ld, float* out_accum, int rows, int cols, int dtype,
              const int32_t* nrows_dev, void* stream) {
  if (rows <= 0 || cols <= 0) return VM_OK;
  dim3 grid((cols + 255) / 256, min((rows + 63) / 64, 128));
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(colsum_k<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, ld,
                                           out_accum, rows, cols, nrows_dev));
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_ce_fwd(const void* logits, int64_t ld, const int64_t* labels, float* row_loss, float* lse, int rows,
              int vocab, int dtype, const int32_t* nrows_dev, void* stream) {
  if (rows <= 0) return VM_OK;
  const int vec = dtype == VM_BF16 ? 8 : 4;
  if (ld % vec) return VM_ERR_BAD_ARG;
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(ce_fwd_k<T>, dim3(rows), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)logits, ld, labels, row_loss, lse, rows, vocab, nrows_dev));
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_ce_bwd(const void* logits, int64_t ld, const int64_t* labels, const float* lse, const float* row_scale,
              void* dlogits, int64_t ld_d, int rows, int vocab, int dtype, const int32_t* nrows_dev, void* stream) {
  if (rows <= 0) return VM_OK;
  const int vec = dtype == VM_BF16 ? 8 : 4;
  if (ld % vec || ld_d % vec) return VM_ERR_BAD_ARG;
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(ce_bwd_k<T>, dim3(rows), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)logits, ld, labels, lse, row_scale, (T*)dlogits, ld_d, rows,
                                           vocab, nrows_dev));
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_im2col3d(const void* image, int C, int D, int H, int W, int pz, int py, int px, void* cols, int64_t ld,
                int dtype, void* stream) {
  if (D % pz || H % py || W % px) return VM_ERR_BAD_ARG;
  const int n = (D / pz) * (H / py) * (W / px);
  if (n <= 0) return VM_OK;
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(im2col3d_k<T>, dim3(n), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)image, C, D, H, W, pz, py, px, (T*)cols, ld));
  VM_LAUNCH_CHECK();
  return VM_OK;
}

}  // extern "C"
