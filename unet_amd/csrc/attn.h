// Launchers of the bfloat16 shifted-window attention kernels (attn.hip), called by the C-ABI entry
// points unetk_window_attn_fwd / unetk_window_attn_bwd in swin.hip.
#pragma once
#include <hip/hip_runtime.h>

struct AttnLaunch {
  const void* qkv;           // (B, He, We, 3C) bf16
  const void* dout;          // (B, He, We, C) bf16, backward only
  void* out;                 // (B, He, We, C) bf16, forward only
  void* dqkv;                // (B, He, We, 3C) bf16, backward only
  const float* table;        // (225, heads)
  float* dtable_partials;    // [B * windows * heads][225], backward only
  int B, He, We, C, heads, shift, use_mask;
  float scale;
};

int attn_bf16_fwd(const AttnLaunch& a, hipStream_t stream);
int attn_bf16_bwd(const AttnLaunch& a, hipStream_t stream);   // fills dtable_partials; the caller reduces them
