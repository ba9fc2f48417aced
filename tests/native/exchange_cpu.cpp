// CPU slice of the multi-GPU layouts' arithmetic for tests/test_distributed_cpu.py (world_size 2 over gloo): the SAME header the
// device kernels and cdae_multi.hip compile (cdae_amd/csrc/cdae_exchange_algebra.h), looped over host arrays.  Test infrastructure.
#include <cstddef>
#include <cstdint>

#include "../../cdae_amd/csrc/cdae_exchange_algebra.h"

extern "C" {

// one boundary pass of the pipelined exchange over a flat block of n floats (delta_pipe_kernel without the pad-column compaction)
void xa_pipe(int mode, float* cur, float* base, float* snap, float* send, float* recv, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    if (mode == cdae_xa::STAGE) cdae_xa::pipe_elem<cdae_xa::STAGE>(cur[i], base[i], snap[i], send[i], recv[i]);
    else if (mode == cdae_xa::MERGE) cdae_xa::pipe_elem<cdae_xa::MERGE>(cur[i], base[i], snap[i], send[i], recv[i]);
    else cdae_xa::pipe_elem<cdae_xa::MERGE_STAGE>(cur[i], base[i], snap[i], send[i], recv[i]);
  }
}

// the same pass under the global-accumulator combine rule, over a (parameter, accumulator) pair of n floats each (delta_pipe_pair_kernel)
void xa_pipe_pair(int mode, float* cp, float* ca, float* Ap, float* Aa, float* snp, float* sna, float* sp, float* sa, float* rp, float* ra,
                  float beta, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    if (mode == cdae_xa::STAGE) cdae_xa::pipe_pair<cdae_xa::STAGE>(cp[i], ca[i], Ap[i], Aa[i], snp[i], sna[i], sp[i], sa[i], rp[i], ra[i], beta);
    else if (mode == cdae_xa::MERGE) cdae_xa::pipe_pair<cdae_xa::MERGE>(cp[i], ca[i], Ap[i], Aa[i], snp[i], sna[i], sp[i], sa[i], rp[i], ra[i], beta);
    else cdae_xa::pipe_pair<cdae_xa::MERGE_STAGE>(cp[i], ca[i], Ap[i], Aa[i], snp[i], sna[i], sp[i], sa[i], rp[i], ra[i], beta);
  }
}

// own_rows_stage_kernel: rows of users [u0, u0 + n) from a table that starts at user own_u0, zeros for users owned elsewhere
void xa_stage_own_rows(const float* table, uint64_t own_u0, uint64_t own_u1, uint64_t u0, uint32_t n, uint32_t width, float* out) {
  for (uint32_t s = 0; s < n; ++s) {
    const uint64_t uid = u0 + s;
    const bool own = cdae_xa::owns_user(uid, own_u0, own_u1);
    for (uint32_t k = 0; k < width; ++k)
      out[(size_t)s * width + k] = cdae_xa::own_row_contribution(own, own ? table[(size_t)(uid - own_u0) * width + k] : 0.f);
  }
}

void xa_balanced_cuts(const int64_t* prefix, uint64_t n, uint64_t S, int at_least_one, uint64_t* cuts) {
  cdae_xa::balanced_cuts(prefix, n, S, at_least_one != 0, cuts);
}

}
