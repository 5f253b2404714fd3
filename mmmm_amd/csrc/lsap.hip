// Rectangular linear sum assignment on the device: the Hungarian matching of InstanceSamLoss._match_instances
// (/root/reference/mmmm/models/segvol/modeling/sam.py:243: `linear_sum_assignment(cost.float().cpu().numpy())`).
//
// The reference moves every cost matrix to the host and calls SciPy — one device->host synchronisation per target; batching
// them still leaves ONE synchronisation in the middle of the step, after which the host has to rebuild its launch lead while
// the GPU starves on the small kernels of the loss and head backward (measured: 12 ms of a 388 ms step). The matrices are tiny
// (6 queries x a handful of targets), so the assignment is solved where the costs are: one thread per problem runs the same
// shortest-augmenting-path algorithm SciPy uses (D. F. Crouse, "On implementing 2D rectangular assignment algorithms", IEEE
// TAES 52(4), 2016 — scipy.optimize.linear_sum_assignment since 1.4), in double precision like SciPy, with its scan order and
// tie rule (columns scanned from the highest index down on first touch, ties resolved towards an unassigned column), so the
// assignment is the one SciPy returns, ties included (tests/test_kernels_gpu.py::test_lsap_matches_scipy).
#include "vm_common.hpp"

namespace {

constexpr int LSAP_MAX = 64;      // rows <= cols <= LSAP_MAX

__global__ void lsap_k(const float* __restrict__ cost, int64_t ld_prob, int64_t ld_row, const int32_t* __restrict__ dims,
                       int32_t* __restrict__ col4row_out, int64_t ld_out, int n_prob) {
  const int pidx = blockIdx.x * blockDim.x + threadIdx.x;
  if (pidx >= n_prob) return;
  const int nr = dims[2 * pidx], nc = dims[2 * pidx + 1];
  int32_t* out = col4row_out + pidx * ld_out;
  if (nr <= 0 || nc < nr || nc > LSAP_MAX) {            // not a problem (padding entry) or outside the supported size
    for (int i = 0; i < (nr > 0 ? nr : 0); ++i) out[i] = -1;
    return;
  }
  const float* c = cost + pidx * ld_prob;
  double u[LSAP_MAX], v[LSAP_MAX], spc[LSAP_MAX];
  int path[LSAP_MAX], col4row[LSAP_MAX], row4col[LSAP_MAX], remaining[LSAP_MAX];
  unsigned long long SR, SC;                             // visited rows / columns of the current search
  for (int i = 0; i < nr; ++i) { u[i] = 0.0; col4row[i] = -1; }
  for (int j = 0; j < nc; ++j) { v[j] = 0.0; row4col[j] = -1; path[j] = -1; }
  const double inf = __longlong_as_double(0x7FF0000000000000ll);
  for (int cur = 0; cur < nr; ++cur) {
    // ---- shortest augmenting path from row `cur`
    double min_val = 0.0;
    int i = cur, num_remaining = nc, sink = -1;
    for (int it = 0; it < nc; ++it) { remaining[it] = nc - it - 1; spc[it] = inf; }
    SR = 0ull; SC = 0ull;
    while (sink == -1) {
      int index = -1;
      double lowest = inf;
      SR |= 1ull << i;
      for (int it = 0; it < num_remaining; ++it) {
        const int j = remaining[it];
        const double r = min_val + (double)c[i * ld_row + j] - u[i] - v[j];
        if (r < spc[j]) { path[j] = i; spc[j] = r; }
        if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
      }
      min_val = lowest;
      if (min_val == inf) break;                          // infeasible (only with infinite costs)
      const int j = remaining[index];
      if (row4col[j] == -1) sink = j; else i = row4col[j];
      SC |= 1ull << j;
      remaining[index] = remaining[--num_remaining];
    }
    if (sink < 0) { for (int k = 0; k < nr; ++k) out[k] = -1; return; }
    // ---- dual update
    u[cur] += min_val;
    for (int k = 0; k < nr; ++k)
      if (((SR >> k) & 1ull) && k != cur) u[k] += min_val - spc[col4row[k]];
    for (int j = 0; j < nc; ++j)
      if ((SC >> j) & 1ull) v[j] -= min_val - spc[j];
    // ---- augment
    int j = sink;
    while (true) {
      const int r = path[j];
      row4col[j] = r;
      const int prev = col4row[r];
      col4row[r] = j;
      j = prev;
      if (r == cur) break;
    }
  }
  for (int i = 0; i < nr; ++i) out[i] = col4row[i];
}

}  // namespace

extern "C" int vm_lsap_f32(const float* cost, int64_t ld_prob, int64_t ld_row, const int32_t* dims_dev, int32_t* col4row,
                           int64_t ld_out, int n_prob, int max_cols, void* stream) {
  if (n_prob < 0 || (n_prob > 0 && (!cost || !dims_dev || !col4row))) return VM_ERR_BAD_ARG;
  if (n_prob == 0) return VM_OK;
  if (max_cols > LSAP_MAX) return VM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(lsap_k, dim3((n_prob + 63) / 64), dim3(64), 0, (hipStream_t)stream, cost, ld_prob, ld_row, dims_dev, col4row,
                     ld_out, n_prob);
  return hipGetLastError() == hipSuccess ? VM_OK : VM_ERR_LAUNCH;
}
