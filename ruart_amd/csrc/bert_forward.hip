// Host-side orchestration of the BERT encoder forward over a packed token stream.
// Reference path: Models/Bert/modeling.py:585-614 (BertModel.forward) -> :326-334 (all layer outputs kept,
// because Models/Bert/Bert.py:137 concatenates every layer).  Seven launches per layer, no host sync,
// no allocation: capturable into a hipGraph by the caller.
#include "common.h"
#include "ruart_hip.h"

extern int ruart_prof_real_rows;

extern "C" const char* ruart_version(void) { return "ruart_hip 0.1 gfx950"; }

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

namespace {
struct Carve {
  char* base;
  size_t off;
  void* take(size_t bytes) {
    void* p = base + off;
    off += align_up(bytes, 256);
    return p;
  }
};
}  // namespace

// RUART_DT_F16C ("f16 + fp8 correction", m->corr8 != 0; common.h): the residual stream, the layer outputs and the QKV rows are fp32;
// every GEMM input exists as an f16 matrix plus a fp8 matrix of the same row pitch (two e4m3 bytes per element).
static size_t corr_workspace_bytes(size_t R, size_t H, size_t I) {
  size_t t = 0;
  t += align_up(R * H * 4, 256);        // x32   embedding output (residual of layer 0)
  t += 2 * align_up(R * H * 2, 256);    // x16, x8   current layer input as GEMM operand
  t += align_up(R * 3 * H * 4, 256);    // qkv (fp32)
  t += 2 * align_up(R * H * 2, 256);    // ctx16, ctx8
  t += align_up(R * H * 4, 256);        // pre-LN rows (fp32)
  t += align_up(R * H * 4, 256);        // mid32
  t += 2 * align_up(R * H * 2, 256);    // mid16, mid8
  t += 2 * align_up(R * I * 2, 256);    // ffn16, ffn8
  return t;
}

extern "C" size_t ruart_bert_workspace_bytes(const ruart_bert_model* m, int n_rows) {
  const size_t es = m->dtype == RUART_DT_F32 ? 4 : 2;
  const size_t R = (size_t)n_rows, H = (size_t)m->hidden, I = (size_t)m->intermediate;
  if (m->corr8) return corr_workspace_bytes(R, H, I);
  size_t t = 0;
  t += align_up(R * H * es, 256);       // x0   embedding output
  t += align_up(R * 3 * H * es, 256);   // qkv
  t += align_up(R * H * es, 256);       // ctx
  t += align_up(R * H * 4, 256);        // pre-LN rows (fp32)
  t += align_up(R * H * es, 256);       // mid  (post-attention LN)
  t += align_up(R * I * es, 256);       // ffn
  return t;
}

// The encoder in the f16 + fp8-correction mode.  layers_out: [n_layers][n_rows][hidden] fp32.
static int bert_forward_corr(const ruart_bert_model* m, const ruart_bert_batch* b, void* layers_out, void* workspace, void* stream) {
  const int H = m->hidden, I = m->intermediate, R = b->n_rows;
  if (R % 256 || H % 256 || I % 256 || b->n_long_blocks != 0 || b->n_blocks <= 0) return (int)hipErrorInvalidValue;
  if (!m->w8_qkv || !m->w8_ao || !m->w8_ff1 || !m->w8_ff2) return (int)hipErrorInvalidValue;
  Carve c{(char*)workspace, 0};
  float* x32 = (float*)c.take((size_t)R * H * 4);
  void* x16 = c.take((size_t)R * H * 2);
  void* x8 = c.take((size_t)R * H * 2);
  float* qkv = (float*)c.take((size_t)R * 3 * H * 4);
  void* ctx16 = c.take((size_t)R * H * 2);
  void* ctx8 = c.take((size_t)R * H * 2);
  float* pre = (float*)c.take((size_t)R * H * 4);
  float* mid32 = (float*)c.take((size_t)R * H * 4);
  void* mid16 = c.take((size_t)R * H * 2);
  void* mid8 = c.take((size_t)R * H * 2);
  void* ffn16 = c.take((size_t)R * I * 2);
  void* ffn8 = c.take((size_t)R * I * 2);
  int rc = ruart_bert_embed_ln_split(b->ids, b->pos_ids, m->word_emb, m->pos_emb, m->type_emb, m->emb_ln_g, m->emb_ln_b, m->ln_eps, x32,
                                     x16, x8, H, R, H, stream);
  if (rc) return rc;
  ruart_prof_real_rows = b->n_tokens;
  const float* res = x32;
  for (int l = 0; l < m->n_layers; ++l) {
    float* out = (float*)layers_out + (size_t)l * R * H;
    if ((rc = ruart_gemm_16c_nt(x16, x8, H, m->w_qkv[l], m->w8_qkv[l], H, m->b_qkv[l], nullptr, 0, qkv, 3 * H, nullptr, R, 3 * H, H,
                                RUART_ACT_NONE, stream)))
      return rc;
    if ((rc = ruart_bert_attention_split(qkv, 3 * H, ctx16, ctx8, H, H, m->n_heads, b->n_blocks, b->blk_q0, b->blk_q1, b->blk_k0, b->blk_k1,
                                         b->tok_lo, b->tok_hi, b->key_bias, stream)))
      return rc;
    if ((rc = ruart_gemm_16c_nt(ctx16, ctx8, H, m->w_ao[l], m->w8_ao[l], H, m->b_ao[l], res, H, pre, H, nullptr, R, H, H, RUART_ACT_NONE,
                                stream)))
      return rc;
    if ((rc = ruart_rows_layernorm_split(pre, H, m->ln1_g[l], m->ln1_b[l], m->ln_eps, mid32, mid16, mid8, H, R, H, stream))) return rc;
    if ((rc = ruart_gemm_16c_nt(mid16, mid8, H, m->w_ff1[l], m->w8_ff1[l], H, m->b_ff1[l], nullptr, 0, ffn16, I, ffn8, R, I, H,
                                RUART_ACT_GELU, stream)))
      return rc;
    if ((rc = ruart_gemm_16c_nt(ffn16, ffn8, I, m->w_ff2[l], m->w8_ff2[l], I, m->b_ff2[l], mid32, H, pre, H, nullptr, R, H, I,
                                RUART_ACT_NONE, stream)))
      return rc;
    if ((rc = ruart_rows_layernorm_split(pre, H, m->ln2_g[l], m->ln2_b[l], m->ln_eps, out, x16, x8, H, R, H, stream))) return rc;
    res = out;
  }
  ruart_prof_real_rows = 0;
  return 0;
}

extern "C" int ruart_bert_forward(const ruart_bert_model* m, const ruart_bert_batch* b, void* layers_out, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  RUART_ENTRY();
  const int H = m->hidden, I = m->intermediate, R = b->n_rows, dt = m->dtype;
  if (R % 128 || b->n_tokens > R || b->n_tokens <= 0 || H % 64 || m->n_heads * 64 != H) return (int)hipErrorInvalidValue;
  if (dt != RUART_DT_F32 && (H % 128 || I % 128)) return (int)hipErrorInvalidValue;
  if (dt != RUART_DT_F32 && dt != RUART_DT_BF16 && dt != RUART_DT_F16) return (int)hipErrorInvalidValue;
  if (workspace_bytes < ruart_bert_workspace_bytes(m, R)) return (int)hipErrorInvalidValue;
  if (m->corr8) {
    if (dt != RUART_DT_F16) return (int)hipErrorInvalidValue;
    return bert_forward_corr(m, b, layers_out, workspace, stream);
  }
  const size_t es = dt == RUART_DT_F32 ? 4 : 2;
  Carve c{(char*)workspace, 0};
  void* x0 = c.take((size_t)R * H * es);
  void* qkv = c.take((size_t)R * 3 * H * es);
  void* ctx = c.take((size_t)R * H * es);
  float* pre = (float*)c.take((size_t)R * H * 4);
  void* mid = c.take((size_t)R * H * es);
  void* ffn = c.take((size_t)R * I * es);

  int rc = ruart_bert_embed_ln(b->ids, b->pos_ids, m->word_emb, m->pos_emb, m->type_emb, m->emb_ln_g, m->emb_ln_b, m->ln_eps, x0, H,
                               dt, R, H, stream);
  if (rc) return rc;

  auto gemm = [&](const void* A, int K, const void* W, const float* bias, const void* res, void* C, int out_dt, int N, int act) {
    if (dt != RUART_DT_F32)
      return ruart_gemm_16_nt(A, K, W, K, bias, res, N, dt, C, N, out_dt, R, N, K, act, dt, stream);
    if (m->f32_gemm == 1)                                        // fp32 storage, split-bf16 products (no K split at these sizes)
      return ruart_gemm_x3((const float*)A, K, 1, (const float*)W, 1, K, bias, (const float*)res, N, act, (float*)C, N, R, N, K,
                           nullptr, 0, nullptr, nullptr, nullptr, 1, stream);
    return ruart_gemm_f32_nt((const float*)A, K, (const float*)W, K, bias, (const float*)res, N, (float*)C, N, R, N, K, act, stream);
  };

  ruart_prof_real_rows = b->n_tokens;
  const void* in = x0;
  for (int l = 0; l < m->n_layers; ++l) {
    void* out = (char*)layers_out + (size_t)l * R * H * es;
    if ((rc = gemm(in, H, m->w_qkv[l], m->b_qkv[l], nullptr, qkv, dt, 3 * H, RUART_ACT_NONE))) return rc;
    if ((rc = ruart_bert_attention(qkv, 3 * H, ctx, H, dt, H, m->n_heads, b->n_blocks, b->blk_q0, b->blk_q1, b->blk_k0, b->blk_k1,
                                   b->tok_lo, b->tok_hi, b->key_bias, b->n_long_blocks, b->lblk_q0, b->lblk_q1, b->lblk_k0, b->lblk_k1, stream)))
      return rc;
    if ((rc = gemm(ctx, H, m->w_ao[l], m->b_ao[l], in, pre, RUART_DT_F32, H, RUART_ACT_NONE))) return rc;
    if ((rc = ruart_rows_layernorm(pre, H, m->ln1_g[l], m->ln1_b[l], m->ln_eps, mid, H, dt, R, H, stream))) return rc;
    if ((rc = gemm(mid, H, m->w_ff1[l], m->b_ff1[l], nullptr, ffn, dt, I, RUART_ACT_GELU))) return rc;
    if ((rc = gemm(ffn, I, m->w_ff2[l], m->b_ff2[l], mid, pre, RUART_DT_F32, H, RUART_ACT_NONE))) return rc;
    if ((rc = ruart_rows_layernorm(pre, H, m->ln2_g[l], m->ln2_b[l], m->ln_eps, out, H, dt, R, H, stream))) return rc;
    in = out;
  }
  ruart_prof_real_rows = 0;
  return 0;
}
