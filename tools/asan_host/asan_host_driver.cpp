// Host side of the C-ABI under AddressSanitizer (tests/test_asan_host.py; built by tools/asan_host/build_and_run.sh).
// No GPU is needed or touched: every call below must return from the host-side argument / shape / workspace checks --
// the code that runs on every call of the product and that a GPU sanitizer build (not available on the pool) would not
// cover any better.  Prints one line per section and "asan driver ok" at the end; any ASan report aborts the process.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mi_critic.h"

static int fails = 0;
#define EXPECT(cond)                                                     \
  do {                                                                   \
    if (!(cond)) {                                                       \
      std::printf("FAILED line %d: %s (last error: %s)\n", __LINE__, #cond, mi_last_error()); \
      ++fails;                                                           \
    }                                                                    \
  } while (0)

int main() {
  EXPECT(mi_abi_version() == 4);
  (void)mi_last_error();

  // ---- workspace / path / record queries over a sweep of shapes, including ragged, tiny, huge and invalid ones
  const int64_t bs[] = {-1, 0, 1, 31, 32, 33, 96, 128, 500, 512, 4096, 8192, 65536};
  const int64_t ds[] = {-4, 0, 1, 8, 24, 64, 128, 256, 512, 768, 1024, 4096};
  const int precs[] = {-1, 0, 1, 2, 3, 4, 5, 6, 99};
  size_t acc = 0;
  for (int64_t b : bs)
    for (int64_t d : ds)
      for (int p : precs) {
        acc += mi_bilinear_workspace_bytes(b, b, d, d, p);
        acc += mi_bilinear_workspace_bytes(b / 4, b, d, d / 2, p);
        acc += mi_separable_workspace_bytes(b, b, d, d, d, p);
        acc += (size_t)(mi_bilinear_path(b, b, d, d, p) + 8);
        acc += (size_t)(mi_separable_path(b, b, d, d, 256, p) + 8);
        size_t off = 0;
        acc += mi_bilinear_raw_records(b, b, d, d, p, &off);
        acc += mi_bilinear_raw_records(b / 8, b, d, d, p, nullptr);
        for (int ng = 0; ng < 2; ++ng) acc += mi_concat_mlp_workspace_bytes(b, b, d, d, 2 * d, d, p, ng);
        acc += mi_concat_mlp_workspace_bytes(b, b, d, d, 1024, 512, p, 1);
      }
  for (int64_t b : bs) acc += mi_bound_workspace_bytes(b) + mi_matrix_bound_workspace_bytes(b) + mi_pair_index_workspace_bytes(b);
  std::printf("queries ok (%zu)\n", acc);

  // ---- null pointers
  EXPECT(mi_bound_fwd(nullptr, 4, 2, 0, nullptr, nullptr, nullptr, 0, nullptr) == MI_EINVAL);
  EXPECT(std::strlen(mi_last_error()) > 0);
  EXPECT(mi_bound_bwd(nullptr, 4, 2, nullptr, nullptr, nullptr, nullptr) == MI_EINVAL);
  EXPECT(mi_matrix_bound_fwd(nullptr, nullptr, 4, 0, nullptr, nullptr, nullptr, 0, nullptr) == MI_EINVAL);
  EXPECT(mi_matrix_bound_bwd(nullptr, nullptr, 4, nullptr, nullptr, nullptr, nullptr) == MI_EINVAL);
  EXPECT(mi_pairs_count_host(nullptr, 4, nullptr, nullptr, 0, nullptr) == MI_EINVAL);
  EXPECT(mi_pair_index(nullptr, 4, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr) == MI_EINVAL);
  EXPECT(mi_create_pairs(nullptr, nullptr, nullptr, nullptr, 4, 8, 8, nullptr, nullptr) == MI_EINVAL);
  EXPECT(mi_create_pairs_bwd(nullptr, nullptr, 4, 8, 8, nullptr, nullptr, nullptr) == MI_EINVAL);
  EXPECT(mi_bilinear_fwd(nullptr, nullptr, nullptr, nullptr, nullptr, 64, 64, 0, 128, 128, 1, 1, 1, nullptr, nullptr, nullptr,
                         nullptr, nullptr, 0, nullptr) == MI_EINVAL);
  EXPECT(mi_bilinear_bwd(nullptr, nullptr, nullptr, nullptr, nullptr, 64, 64, 0, 128, 128, 1, nullptr, nullptr, nullptr, nullptr,
                         nullptr, nullptr, 0, 0, nullptr) == MI_EINVAL);
  EXPECT(mi_bilinear_prep_local(nullptr, nullptr, 64, 64, 128, 128, 1, nullptr, 0, nullptr) == MI_EINVAL);
  EXPECT(mi_bilinear_bwd_records(nullptr, nullptr, nullptr, nullptr, nullptr, 64, 64, 0, 128, 128, 1, 1, nullptr, 0, 0, nullptr,
                                 nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr) == MI_EINVAL);
  EXPECT(mi_bilinear_bwd_dw(64, 64, 128, 128, 1, nullptr, nullptr, 0, nullptr) == MI_EINVAL);
  EXPECT(mi_bilinear_fp8_stage(nullptr, nullptr, nullptr, 64, 64, 128, 128, 0, nullptr, nullptr, 0, nullptr) == MI_EINVAL);
  EXPECT(mi_bilinear_step(nullptr, nullptr, nullptr, nullptr, 64, 128, 128, 1, 1, nullptr, nullptr, nullptr, nullptr, nullptr,
                          nullptr, nullptr, nullptr, 0, nullptr) == MI_EINVAL);
  EXPECT(mi_bilinear_step_bf16(nullptr, nullptr, nullptr, nullptr, 64, 128, 128, 1, nullptr, nullptr, nullptr, nullptr, nullptr,
                               nullptr, 1, nullptr, nullptr, 0, nullptr) == MI_EINVAL);
  EXPECT(mi_separable_fwd(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 64, 64, 0, 128, 128, 128, 1, 1, 1, nullptr,
                          nullptr, nullptr, nullptr, 0, nullptr) == MI_EINVAL);
  EXPECT(mi_separable_bwd(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 64, 64, 0, 128, 128, 128, 1, nullptr, nullptr,
                          nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr) == MI_EINVAL);
  EXPECT(mi_separable_step(nullptr, nullptr, nullptr, nullptr, nullptr, 64, 128, 128, 128, 1, 1, nullptr, nullptr, nullptr,
                           nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr) == MI_EINVAL);
  EXPECT(mi_concat_mlp_fwd(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 64, 64, 0,
                           128, 128, 128, 256, 1, 1, 1, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr) == MI_EINVAL);
  EXPECT(mi_concat_mlp_bwd(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 64, 64, 0,
                           128, 128, 128, 256, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, nullptr, nullptr, nullptr, 0, nullptr) == MI_EINVAL);
  EXPECT(mi_merge_partials(nullptr, 1, 1, 0, nullptr, nullptr, nullptr) == MI_EINVAL);
  std::printf("null-pointer checks ok\n");

  // ---- real (host) pointers, shapes and precisions the entry points must refuse, and workspaces that are too small: the
  // shape checks and the planners run on real addresses, nothing may be read or written through them
  const int64_t b = 64, d = 128, h1 = 128, h2 = 256;
  std::vector<float> x(b * d), y(b * d), w(d * d), big(4 * h1 * h2 + b * b), out(b * d), gw(d * d);
  std::vector<int64_t> sid(b);
  for (int64_t i = 0; i < b; ++i) sid[i] = i;
  mi_stats st;
  std::memset(&st, 0, sizeof st);
  float loss = 0.0f, go = 1.0f, rec[8] = {0};
  std::vector<char> ws(4096);  // far too small for any of the calls below
  for (int p : precs) {
    for (int est : {-1, 0, 1, 7}) {
      const bool bad_enum = p < 0 || p > 5 || est < 0 || est > 1;
      int rc = mi_bilinear_step(x.data(), y.data(), w.data(), sid.data(), b, d, d, est, p, &go, &loss, &st, rec, out.data(),
                                out.data(), gw.data(), ws.data(), ws.size(), nullptr);
      EXPECT(rc != MI_OK);
      if (bad_enum) EXPECT(rc == MI_EINVAL);
      rc = mi_bilinear_fwd(x.data(), y.data(), w.data(), sid.data(), sid.data(), b, b, 0, d, d, est, p, 1, &loss, &st, rec,
                           nullptr, ws.data(), ws.size(), nullptr);
      EXPECT(rc != MI_OK);
      rc = mi_separable_fwd(x.data(), y.data(), w.data(), w.data(), sid.data(), sid.data(), b, b, 0, d, d, d, est, p, 1, &loss,
                            &st, rec, ws.data(), ws.size(), nullptr);
      EXPECT(rc != MI_OK);
      rc = mi_concat_mlp_fwd(x.data(), y.data(), big.data(), big.data(), big.data(), big.data(), big.data(), big.data(),
                             sid.data(), sid.data(), b, b, 0, d, d, h1, h2, est, p, 1, &loss, &st, rec, big.data(), ws.data(),
                             ws.size(), nullptr);
      EXPECT(rc != MI_OK);
    }
  }
  // shapes: zero / negative sizes, a row block that overruns the batch, widths the concat kernels do not take
  EXPECT(mi_bilinear_fwd(x.data(), y.data(), w.data(), sid.data(), sid.data(), 0, b, 0, d, d, 1, 1, 1, &loss, &st, rec, nullptr,
                         ws.data(), ws.size(), nullptr) != MI_OK);
  EXPECT(mi_bilinear_fwd(x.data(), y.data(), w.data(), sid.data(), sid.data(), b, b, 1, d, d, 1, 1, 1, &loss, &st, rec, nullptr,
                         ws.data(), ws.size(), nullptr) != MI_OK);
  EXPECT(mi_bilinear_fwd(x.data(), y.data(), w.data(), sid.data(), sid.data(), b, b, 0, -d, d, 1, 1, 1, &loss, &st, rec, nullptr,
                         ws.data(), ws.size(), nullptr) != MI_OK);
  EXPECT(mi_concat_mlp_fwd(x.data(), y.data(), big.data(), big.data(), big.data(), big.data(), big.data(), big.data(),
                           sid.data(), sid.data(), b, b, 0, d, d, 100, 300, 1, 1, 1, &loss, &st, rec, big.data(), ws.data(),
                           ws.size(), nullptr) != MI_OK);
  EXPECT(mi_bound_fwd(x.data(), 16, 17, 0, &loss, &st, ws.data(), ws.size(), nullptr) != MI_OK);   // pos_size > n
  EXPECT(mi_bound_fwd(x.data(), 16, 4, 9, &loss, &st, ws.data(), ws.size(), nullptr) != MI_OK);    // estimator
  EXPECT(mi_bound_fwd(x.data(), 1 << 20, 4, 0, &loss, &st, ws.data(), 8, nullptr) != MI_OK);       // workspace
  EXPECT(mi_merge_partials(rec, 0, 1, 0, &loss, &st, nullptr) != MI_OK);
  std::printf("shape / enum / workspace checks ok\n");

  // ---- the profiling hook's host bookkeeping
  char names[256];
  float ms[8];
  int n = -1;
  EXPECT(mi_profile_end(names, sizeof names, ms, 8, &n) != MI_OK || n == 0);  // end without begin
  std::printf(fails ? "asan driver FAILED (%d)\n" : "asan driver ok\n", fails);
  return fails ? 1 : 0;
}
