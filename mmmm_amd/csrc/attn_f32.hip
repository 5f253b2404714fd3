// fp32 attention for the fp32 islands (SAM ViT-B encoder, two-way transformer). See attn_f32 section of DESIGN.md.
#include "vm_common.hpp"
extern "C" {
int vm_attn_fwd_f32(const vm_attn_f32_args* a, void* stream) { (void)a; (void)stream; return VM_ERR_UNSUPPORTED; }
int vm_attn_bwd_f32(const vm_attn_f32_args* a, void* stream) { (void)a; (void)stream; return VM_ERR_UNSUPPORTED; }
}
