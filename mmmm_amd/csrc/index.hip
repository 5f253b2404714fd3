// Token routing metadata built on the device (no host sync).
// Bit-exact restatement of get_expert_mask (reference mmmm/models/cogvlm/modeling_cogvlm.py:58-70)
// plus the packed, expert-sorted row layout used by every LM kernel (DESIGN.md "LM layout").
#include "vm_common.hpp"

namespace {

__global__ __launch_bounds__(1024) void expert_index_k(
    const int64_t* __restrict__ tt, const int64_t* __restrict__ am, int B, int L,
    int32_t* __restrict__ counts, int32_t* __restrict__ row_of_tok, int32_t* __restrict__ tok_of_row,
    int32_t* __restrict__ cu_seqlens, int32_t* __restrict__ row_of_pos, uint8_t* __restrict__ expert_mask) {
  extern __shared__ int32_t sh[];      // [3*B + 4]: n_valid, n_vis per sample, then prefix arrays
  int32_t* s_valid = sh;
  int32_t* s_vis = sh + B;
  int32_t* s_maxlen = sh + 2 * B;
  const int tid = threadIdx.x;
  const int total = B * L;
  // pass 1: masks
  for (int t = tid; t < total; t += blockDim.x) {
    const int b = t / L, l = t % L;
    bool vis = false;
    if (l + 1 < L) vis = (tt[t] == 1) && (tt[t + 1] == 1);
    bool lang = !vis;
    bool valid = true;
    if (L > 1) { valid = am[t] != 0; vis = vis && valid; lang = lang && valid; }
    expert_mask[t] = (uint8_t)((vis ? 1 : 0) | (lang ? 2 : 0));
    row_of_tok[t] = -1;
    tok_of_row[t] = -1;
    row_of_pos[t] = -1;
    (void)b;
  }
  __syncthreads();
  // pass 2: per-sample counts (one thread per sample; B is small)
  for (int b = tid; b < B; b += blockDim.x) {
    int nv = 0, nvis = 0;
    for (int l = 0; l < L; ++l) {
      const uint8_t m = expert_mask[b * L + l];
      nv += (m != 0); nvis += (m & 1);
    }
    s_valid[b] = nv; s_vis[b] = nvis;
  }
  __syncthreads();
  if (tid == 0) {
    int tv = 0, tvis = 0, mx = 0;
    cu_seqlens[0] = 0;
    for (int b = 0; b < B; ++b) {
      tv += s_valid[b]; tvis += s_vis[b]; mx = max(mx, s_valid[b]);
      cu_seqlens[b + 1] = tv;
    }
    counts[0] = tvis; counts[1] = tv; counts[2] = mx; counts[3] = 0;
    s_maxlen[0] = tvis;
  }
  __syncthreads();
  const int n_vis_total = s_maxlen[0];
  // pass 3: assign rows; thread b walks its sample in order
  for (int b = tid; b < B; b += blockDim.x) {
    int vis_before = 0, lang_before = 0, pos0 = 0;
    for (int bb = 0; bb < b; ++bb) { vis_before += s_vis[bb]; lang_before += s_valid[bb] - s_vis[bb]; pos0 += s_valid[bb]; }
    int iv = vis_before, il = n_vis_total + lang_before, ip = pos0;
    for (int l = 0; l < L; ++l) {
      const int t = b * L + l;
      const uint8_t m = expert_mask[t];
      if (m == 0) continue;
      const int row = (m & 1) ? iv++ : il++;
      row_of_tok[t] = row;
      tok_of_row[row] = t;
      row_of_pos[ip++] = row;
    }
  }
}

}  // namespace

extern "C" int vm_expert_index_build(const int64_t* token_type_ids, const int64_t* attention_mask, int B, int L,
                                     int32_t* counts, int32_t* row_of_tok, int32_t* tok_of_row,
                                     int32_t* cu_seqlens, int32_t* row_of_pos, uint8_t* expert_mask, void* stream) {
  if (B <= 0 || L <= 0) return VM_ERR_BAD_ARG;
  const size_t shmem = (size_t)(3 * B + 4) * sizeof(int32_t);
  if (shmem > 60000) return VM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(expert_index_k, dim3(1), dim3(1024), shmem, (hipStream_t)stream, token_type_ids, attention_mask,
                     B, L, counts, row_of_tok, tok_of_row, cu_seqlens, row_of_pos, expert_mask);
  VM_LAUNCH_CHECK();
  return VM_OK;
}
