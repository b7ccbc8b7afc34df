// C handles over the header-only pieces of the host mirror (texturefusion_amd/host/tf_chisel.hpp) that
// oracle/_ref can check against the compiled reference: SparseMat, QuadraticTruncator, ConstantWeighter.
// Same entry-point shapes as oracle/ref_shim.cpp (prefix tfmir_ instead of tfref_); tests/test_ref_pin.py
// drives both with the same operation streams.
#include <cstdint>

#include "../../texturefusion_amd/host/tf_chisel.hpp"

extern "C" {
void tfmir_truncation_n(float q, float l, float c, float s, const float* z, float* out, int64_t n) {
  chisel::QuadraticTruncator t(q, l, c, s);
  for (int64_t i = 0; i < n; i++) out[i] = t.GetTruncationDistance(z[i]);
}
void tfmir_weight_n(float w, const float* trunc, float* out, int64_t n) {
  chisel::ConstantWeighter wt(w);
  for (int64_t i = 0; i < n; i++) out[i] = wt.GetWeight(0.0f, trunc[i]);
}
void* tfmir_sm_new(void) { return new chisel::SparseMat(); }
void tfmir_sm_free(void* m) { delete (chisel::SparseMat*)m; }
uint64_t tfmir_sm_cols(void* m) { return ((chisel::SparseMat*)m)->cols(); }
uint64_t tfmir_sm_rows(void* m) { return ((chisel::SparseMat*)m)->rows(); }
uint64_t tfmir_sm_nnz(void* m) { return ((chisel::SparseMat*)m)->get_nnz(); }
int tfmir_sm_add_value(void* m, uint64_t c, uint64_t r, float v) { return ((chisel::SparseMat*)m)->add_value(c, r, v) ? 1 : 0; }
void tfmir_sm_set_value(void* m, uint64_t c, uint64_t r, float v) { ((chisel::SparseMat*)m)->set_value(c, r, v); }
void tfmir_sm_resize(void* m, uint64_t c) { ((chisel::SparseMat*)m)->resize(c); }
void tfmir_sm_clear(void* m) { ((chisel::SparseMat*)m)->clear(); }
void tfmir_sm_remove_node(void* m, uint64_t c) { ((chisel::SparseMat*)m)->remove_node(c); }
void tfmir_sm_remove_observation(void* m, uint64_t c, uint64_t r) { ((chisel::SparseMat*)m)->remove_observation(c, r); }
uint64_t tfmir_sm_col(void* m, uint64_t c, uint64_t* rows, float* vals, uint64_t cap) {
  const chisel::SparseMat::Column& col = ((chisel::SparseMat*)m)->col(c);
  uint64_t k = 0;
  for (const auto& e : col) {
    if (k < cap) { rows[k] = e.first; vals[k] = e.second; }
    k++;
  }
  return k;
}
}
