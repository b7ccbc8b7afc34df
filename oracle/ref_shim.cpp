// oracle/_ref: the part of the reference that compiles in this image, compiled UNMODIFIED from /root/reference.
//
// TEST INFRASTRUCTURE ONLY (like everything under oracle/): built by `make -C oracle ref` when /root/reference is
// present, output oracle/_ref/libtf_ref.so (git-ignored, travels with gpurun).  Nothing under texturefusion_amd/
// may load it.
//
// What is in it -- every file on the hot path that needs none of Eigen / OpenCV / Sophus / TBB:
//   3rd_party/open_chisel/truncation/{Truncator,QuadraticTruncator,ConstantTruncator}.h   
//   3rd_party/open_chisel/weighting/{Weighter,ConstantWeighter}.h                         (row a8)
//   3rd_party/open_chisel/threading/Threading.h    chisel::parallel_for, the thread policy of rows a6 and f-1
//   Structure/sparse_matrix.{h,cpp}                SparseMat = the data-cost matrix of row f-4 (TexMap.cpp:63-105)
// This file holds NO reference code: it includes those headers from where they lie (-I/root/reference/...) and
// wraps them in a C ABI the tests bind with ctypes.  K-A, the selection, the mesher and the atlas need Eigen (and
// OpenCV / Sophus) and stay unpinned -- see DESIGN.md s.5.
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <numeric>
#include <thread>
#include <vector>

#include "truncation/Truncator.h"
#include "truncation/QuadraticTruncator.h"
#include "truncation/ConstantTruncator.h"
#include "weighting/Weighter.h"
#include "weighting/ConstantWeighter.h"
#include "threading/Threading.h"
#include "sparse_matrix.h"

extern "C" {

// chisel::QuadraticTruncator::GetTruncationDistance through the base-class pointer, as ProjectionIntegrator.cpp:91 calls it
float tfref_truncation(float q, float l, float c, float s, float z) {
  chisel::TruncatorPtr t = std::make_shared<chisel::QuadraticTruncator>(q, l, c, s);
  return t->GetTruncationDistance(z);
}

void tfref_truncation_n(float q, float l, float c, float s, const float* z, float* out, int64_t n) {
  chisel::TruncatorPtr t = std::make_shared<chisel::QuadraticTruncator>(q, l, c, s);
  for (int64_t i = 0; i < n; i++) out[i] = t->GetTruncationDistance(z[i]);
}

float tfref_constant_truncation(float v, float z) {
  chisel::TruncatorPtr t = std::make_shared<chisel::ConstantTruncator>(v);
  return t->GetTruncationDistance(z);
}

// chisel::ConstantWeighter::GetWeight (ProjectionIntegrator.cpp:92)
void tfref_weight_n(float w, const float* trunc, float* out, int64_t n) {
  chisel::WeighterPtr wt = std::make_shared<chisel::ConstantWeighter>(w);
  for (int64_t i = 0; i < n; i++) out[i] = wt->GetWeight(0.0f, trunc[i]);
}

// chisel::parallel_for over n items exactly as Chisel.h:234 calls it (an index vector, default threshold):
// worker[i] = ordinal of the thread that ran item i (0 = first group ... ; the calling thread runs the last
// stretch), returns the number of distinct threads.  *nthreads_out = the function-static thread budget the
// reference took from this machine (hardware_concurrency() - 2).
int tfref_parallel_for_groups(int64_t n, int32_t* worker, int* nthreads_out) {
  std::vector<int> index((size_t)n);
  std::iota(index.begin(), index.end(), 0);
  std::vector<std::thread::id> who((size_t)n);
  chisel::parallel_for(index.begin(), index.end(), [&](const int& i) { who[(size_t)i] = std::this_thread::get_id(); });
  std::vector<std::thread::id> seen;
  for (int64_t i = 0; i < n; i++) {
    size_t k = 0;
    while (k < seen.size() && seen[k] != who[(size_t)i]) k++;
    if (k == seen.size()) seen.push_back(who[(size_t)i]);
    worker[i] = (int32_t)k;
  }
  if (nthreads_out) *nthreads_out = (int)std::thread::hardware_concurrency() - 2;
  return (int)seen.size();
}

// SparseMat (Structure/sparse_matrix.{h,cpp}) behind handles
void* tfref_sm_new(void) { return new SparseMat(); }
void* tfref_sm_new2(uint64_t cols, uint64_t rows) { return new SparseMat(cols, rows); }
void tfref_sm_free(void* m) { delete (SparseMat*)m; }
uint64_t tfref_sm_cols(void* m) { return ((SparseMat*)m)->cols(); }
uint64_t tfref_sm_rows(void* m) { return ((SparseMat*)m)->rows(); }
uint64_t tfref_sm_nnz(void* m) { return ((SparseMat*)m)->get_nnz(); }
int tfref_sm_add_value(void* m, uint64_t c, uint64_t r, float v) { return ((SparseMat*)m)->add_value(c, r, v) ? 1 : 0; }
void tfref_sm_set_value(void* m, uint64_t c, uint64_t r, float v) { ((SparseMat*)m)->set_value(c, r, v); }
void tfref_sm_resize(void* m, uint64_t c) { ((SparseMat*)m)->resize(c); }
void tfref_sm_clear(void* m) { ((SparseMat*)m)->clear(); }
void tfref_sm_remove_node(void* m, uint64_t c) { ((SparseMat*)m)->remove_node(c); }
void tfref_sm_remove_observation(void* m, uint64_t c, uint64_t r) { ((SparseMat*)m)->remove_observation(c, r); }
// column c as (row, value) pairs in map order; returns the column's size (may exceed cap)
uint64_t tfref_sm_col(void* m, uint64_t c, uint64_t* rows, float* vals, uint64_t cap) {
  const SparseMat::Column& col = ((SparseMat*)m)->col(c);
  uint64_t k = 0;
  for (const auto& e : col) {
    if (k < cap) { rows[k] = e.first; vals[k] = e.second; }
    k++;
  }
  return k;
}

}  // extern "C"
