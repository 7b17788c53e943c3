// smx_dataset.hip -- the resident cells x genes matrix: uploads (float32 / uint16 / CSR), library statistics, corruption, read-back
// (host side of smx_data.hip).
#include "smx_model.h"

extern "C" {

static int dataset_upload_impl(smx_model* m, const void* X, bool u16, int64_t n_cells, const float* const* labels,
                               const float* library, const uint8_t* label_mask, int64_t cell_id_base);

int smx_dataset_upload(smx_model* m, const float* X, int64_t n_cells, const float* const* labels, const float* library,
                       const uint8_t* label_mask, int64_t cell_id_base) {
  return dataset_upload_impl(m, X, false, n_cells, labels, library, label_mask, cell_id_base);
}

int smx_dataset_upload_u16(smx_model* m, const uint16_t* X, int64_t n_cells, const float* const* labels, const float* library,
                           const uint8_t* label_mask, int64_t cell_id_base) {
  return dataset_upload_impl(m, X, true, n_cells, labels, library, label_mask, cell_id_base);
}

static int upload_side_arrays(smx_model* m, int64_t n_cells, const float* const* labels, const float* library, const uint8_t* label_mask) {
  int rc;
  for (int j = 0; j < m->cfg.n_labels; ++j) {
    const int P = m->cfg.label_dim[j], Pp = m->lab_Pp[j];
    if ((rc = dmalloc(&m->Y[j], (size_t)n_cells * Pp))) return rc;
    SMX_HIP(hipMemcpy2D(m->Y[j], (size_t)Pp * sizeof(float), labels[j], (size_t)P * sizeof(float), (size_t)P * sizeof(float),
                        (size_t)n_cells, hipMemcpyHostToDevice));
  }
  if (library) {
    if ((rc = dmalloc(&m->library, (size_t)n_cells * 2))) return rc;
    SMX_HIP(hipMemcpy(m->library, library, (size_t)n_cells * 2 * sizeof(float), hipMemcpyHostToDevice));
  }
  if (label_mask) {
    if ((rc = dmalloc(&m->mask, (size_t)n_cells))) return rc;
    SMX_HIP(hipMemcpy(m->mask, label_mask, (size_t)n_cells, hipMemcpyHostToDevice));
  }
  return SMX_OK;
}

// Compact sparse store: the counts as CSR (indptr [n_cells + 1], column indices and values of the non-zeros, rows in
// order, columns < n_genes) -- 8 bytes per non-zero instead of 4 per entry (7-12 % non-zeros in the named datasets).
// Every pass expands its minibatch's rows into a dense float32 tile first (csr_stage), so results are bit-identical to
// the float32 store; the resident-matrix kernels (library statistics, corruption) stay with the dense stores.
int smx_dataset_upload_csr(smx_model* m, const int64_t* indptr, const int32_t* cols, const float* vals, int64_t n_cells,
                           const float* const* labels, const float* library, const uint8_t* label_mask, int64_t cell_id_base) {
  SMX_REQUIRE(m && indptr && n_cells > 0, "bad dataset");
  SMX_REQUIRE(n_cells < (int64_t)1 << 31, "row ids are int32");
  SMX_REQUIRE(!m->scvi || library, "scvi needs the library prior (scvi.py:100-105)");
  for (int j = 0; j < m->cfg.n_labels; ++j) SMX_REQUIRE(labels && labels[j], "missing label matrix");
  const int64_t nnz = indptr[n_cells];
  SMX_REQUIRE(indptr[0] == 0 && nnz >= 0 && (nnz == 0 || (cols && vals)), "bad CSR arrays");
  for (int64_t r = 0; r < n_cells; ++r) SMX_REQUIRE(indptr[r + 1] >= indptr[r], "CSR indptr must not decrease");
  for (int64_t i = 0; i < nnz; ++i) SMX_REQUIRE(cols[i] >= 0 && cols[i] < m->G, "CSR column index out of range");
  SMX_HIP(hipStreamSynchronize(m->st));
  drop_graphs(m);
  auto fr = [](void* p) { if (p) hipFree(p); };
  release_csr(m);
  fr(m->X); fr(m->library); fr(m->mask); fr(m->lgx1);
  m->X = nullptr; m->library = nullptr; m->mask = nullptr; m->lgx1 = nullptr;
  for (int j = 0; j < SMX_MAX_LABELS; ++j) { fr(m->Y[j]); m->Y[j] = nullptr; }
  m->N = n_cells; m->cell_base = cell_id_base; m->x_u16 = false; m->staged_steps = 0;   // (row ids staged against the matrix before are void)
  int rc;
  m->x_csr = true;
  if ((rc = dmalloc(&m->csr_indptr, (size_t)n_cells + 1)) || (rc = dmalloc(&m->csr_cols, (size_t)std::max<int64_t>(nnz, 1))) ||
      (rc = dmalloc(&m->csr_vals, (size_t)std::max<int64_t>(nnz, 1))) || (rc = dmalloc(&m->xbatch, (size_t)m->Bmax * m->Gp)) ||
      (rc = dmalloc(&m->lgx1, (size_t)n_cells)))
    return rc;
  m->X = m->xbatch;   // (non-null: "a dataset is resident"; csr_stage fills it per pass)
  SMX_HIP(hipMemcpy(m->csr_indptr, indptr, ((size_t)n_cells + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  if (nnz) {
    SMX_HIP(hipMemcpy(m->csr_cols, cols, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice));
    SMX_HIP(hipMemcpy(m->csr_vals, vals, (size_t)nnz * sizeof(float), hipMemcpyHostToDevice));
  }
  SMX_CHECK(launch_csr_row_stats(m->st, m->csr_indptr, m->csr_vals, m->N, m->lgx1));
  return upload_side_arrays(m, n_cells, labels, library, label_mask);
}

static int dataset_upload_impl(smx_model* m, const void* X, bool u16, int64_t n_cells, const float* const* labels,
                               const float* library, const uint8_t* label_mask, int64_t cell_id_base) {
  SMX_REQUIRE(m && X && n_cells > 0, "bad dataset");
  SMX_REQUIRE(n_cells < (int64_t)1 << 31, "row ids are int32");
  SMX_REQUIRE(!m->scvi || library, "scvi needs the library prior (scvi.py:100-105)");
  for (int j = 0; j < m->cfg.n_labels; ++j) SMX_REQUIRE(labels && labels[j], "missing label matrix");
  SMX_HIP(hipStreamSynchronize(m->st));
  drop_graphs(m);
  auto fr = [](void* p) { if (p) hipFree(p); };
  release_csr(m);
  fr(m->X); fr(m->library); fr(m->mask); fr(m->lgx1);
  m->X = nullptr; m->library = nullptr; m->mask = nullptr; m->lgx1 = nullptr;
  for (int j = 0; j < SMX_MAX_LABELS; ++j) { fr(m->Y[j]); m->Y[j] = nullptr; }
  m->N = n_cells; m->cell_base = cell_id_base; m->staged_steps = 0;
  int rc;
  m->x_u16 = u16;
  if (u16) {   // compact store: uint16 counts, same row pitch in ELEMENTS (Gp), half the bytes
    uint16_t* xh = nullptr;
    if ((rc = dmalloc(&xh, (size_t)n_cells * m->Gp)) || (rc = dmalloc(&m->lgx1, (size_t)n_cells))) return rc;
    m->X = reinterpret_cast<float*>(xh);
    SMX_HIP(hipMemcpy2D(xh, (size_t)m->Gp * sizeof(uint16_t), X, (size_t)m->G * sizeof(uint16_t), (size_t)m->G * sizeof(uint16_t),
                        (size_t)n_cells, hipMemcpyHostToDevice));
  } else {
    if ((rc = dmalloc(&m->X, (size_t)n_cells * m->Gp)) || (rc = dmalloc(&m->lgx1, (size_t)n_cells))) return rc;
    SMX_HIP(hipMemcpy2D(m->X, (size_t)m->Gp * sizeof(float), X, (size_t)m->G * sizeof(float), (size_t)m->G * sizeof(float),
                        (size_t)n_cells, hipMemcpyHostToDevice));
  }
  // per-row constant sum_g lgamma(x+1) of the likelihood, on the device (one wave per row)
  SMX_CHECK(launch_row_stats(m->st, m->X, m->x_u16 ? 1 : 0, m->Gp, m->N, m->G, m->lgx1, nullptr));
  return upload_side_arrays(m, n_cells, labels, library, label_mask);
}

// BASELINE.json configs[4] (synthetic 1e6 cells x 20k genes log-normal counts): the rank's shard generated ON the device
//  -- 40 GB as uint16 at the full size, nothing crosses PCIe.  Rows are those of one virtual matrix keyed by
// the global cell id rank * n_cells + row (smx_data.hip: generate_lognormal_kernel).
int smx_dataset_generate_lognormal(smx_model* m, uint64_t seed, int32_t rank, int64_t n_cells, int32_t storage_u16, double density) {
  SMX_REQUIRE(m && n_cells > 0 && rank >= 0, "bad arguments");
  SMX_REQUIRE(n_cells < (int64_t)1 << 31 && (int64_t)rank * n_cells + n_cells <= (int64_t)0xFFFFFFFF, "cell ids are 32-bit");
  SMX_REQUIRE(density > 0.0 && density <= 1.0, "density must be in (0, 1]");
  SMX_REQUIRE(!m->scvi && m->cfg.n_labels == 0, "the generator makes counts only (no library prior, no labels)");
  SMX_HIP(hipStreamSynchronize(m->st));
  drop_graphs(m);
  auto fr = [](void* p) { if (p) hipFree(p); };
  release_csr(m);
  fr(m->X); fr(m->library); fr(m->mask); fr(m->lgx1);
  m->X = nullptr; m->library = nullptr; m->mask = nullptr; m->lgx1 = nullptr;
  for (int j = 0; j < SMX_MAX_LABELS; ++j) { fr(m->Y[j]); m->Y[j] = nullptr; }
  m->N = n_cells; m->cell_base = (int64_t)rank * n_cells; m->x_u16 = storage_u16 != 0; m->staged_steps = 0;
  const size_t bytes = (size_t)n_cells * (size_t)m->Gp * (m->x_u16 ? sizeof(uint16_t) : sizeof(float));
  hipError_t e = hipMalloc((void**)&m->X, bytes);   // (no memset of tens of GB: the generator writes every element of every padded row)
  if (e != hipSuccess) { m->X = nullptr; m->N = 0; set_error(std::string("hipMalloc of the resident matrix failed: ") + hipGetErrorString(e)); return SMX_ERR_NOMEM; }
  int rc;
  float* mu = nullptr;
  if ((rc = dmalloc(&m->lgx1, (size_t)n_cells)) || (rc = dmalloc(&mu, (size_t)m->Gp))) return rc;
  // (Gp is a multiple of 32: the kernel covers the padded columns too, writing zeros beyond G)
  rc = launch_generate_lognormal(m->st, m->X, m->x_u16 ? 1 : 0, m->Gp, m->N, m->G, seed, (uint32_t)m->cell_base, (float)density, mu);
  if (rc == SMX_OK) rc = launch_row_stats(m->st, m->X, m->x_u16 ? 1 : 0, m->Gp, m->N, m->G, m->lgx1, nullptr);
  hipError_t es = hipStreamSynchronize(m->st);
  hipFree(mu);
  if (rc == SMX_OK && es != hipSuccess) { set_error(std::string("generator failed: ") + hipGetErrorString(es)); rc = SMX_ERR_HIP; }
  return rc;
}

int64_t smx_dataset_size(const smx_model* m) { return m ? m->N : 0; }

int smx_dataset_library(smx_model* m, float stats[2]) {
  SMX_REQUIRE(m && m->X && m->N > 0, "no resident dataset");
  SMX_REQUIRE(!m->x_csr, "the resident-matrix kernels take a dense store (float32 / uint16), not the sparse one");
  SMX_HIP(hipStreamSynchronize(m->st));
  double* work = nullptr;   // [N] log counts + [2] moments
  int rc;
  if ((rc = dmalloc(&work, (size_t)m->N + 2))) return rc;
  if (!m->library && (rc = dmalloc(&m->library, (size_t)m->N * 2))) { hipFree(work); return rc; }
  drop_graphs(m);   // a captured step may hold the old (null) library pointer
  rc = launch_row_stats(m->st, m->X, m->x_u16 ? 1 : 0, m->Gp, m->N, m->G, m->lgx1, work);
  if (rc == SMX_OK) rc = launch_library_stats(m->st, work, m->N, work + m->N, m->library);
  double h[2] = {0.0, 0.0};
  if (rc == SMX_OK) {
    hipError_t e = hipMemcpyAsync(h, work + m->N, sizeof(h), hipMemcpyDeviceToHost, m->st);
    if (e == hipSuccess) e = hipStreamSynchronize(m->st);
    if (e != hipSuccess) { set_error(std::string("dataset_library failed: ") + hipGetErrorString(e)); rc = SMX_ERR_HIP; }
  }
  hipFree(work);
  if (rc == SMX_OK && stats) { stats[0] = (float)h[0]; stats[1] = (float)h[1]; }
  return rc;
}

int smx_dataset_corrupt(smx_model* m, double dropout, double retain_rate, uint64_t seed, int64_t* n_corrupted) {
  SMX_REQUIRE(m && m->X && m->N > 0, "no resident dataset");
  SMX_REQUIRE(!m->x_csr, "the resident-matrix kernels take a dense store (float32 / uint16), not the sparse one");
  SMX_REQUIRE(dropout >= 0.0 && dropout < 1.0, "dropout value must be >= 0 and < 1");   // utils.py:184-185
  SMX_REQUIRE(retain_rate >= 0.0 && retain_rate <= 1.0, "retain_rate must be in [0, 1]");
  if (n_corrupted) *n_corrupted = 0;
  if (!((dropout > 0.0 && dropout < 1.0) || (retain_rate > 0.0 && retain_rate < 1.0))) return SMX_OK;   // utils.py:188-189
  SMX_HIP(hipStreamSynchronize(m->st));
  unsigned long long* hist = nullptr;
  int rc;
  if ((rc = dmalloc(&hist, 256))) return rc;
  CorruptArgs a;
  a.X = m->X; a.ld = m->Gp; a.N = m->N; a.G = m->G; a.u16 = m->x_u16 ? 1 : 0;
  a.k0 = (uint32_t)(seed & 0xFFFFFFFFu); a.k1 = (uint32_t)(seed >> 32); a.cell_base = (uint32_t)m->cell_base;
  a.hist = hist;
  a.thr_binom = (uint64_t)floor(retain_rate * 4294967296.0);
  unsigned long long h[256];
  unsigned long long rank = 0;   // 1-based rank of the threshold key among the keys that share the prefix
  bool nothing = false;
  for (int pass = 0; pass < 8 && rc == SMX_OK && !nothing; ++pass) {
    hipError_t e = hipMemsetAsync(hist, 0, sizeof(h), m->st);
    if (e == hipSuccess) rc = launch_corrupt_hist(m->st, a, pass);
    if (rc == SMX_OK && e == hipSuccess) e = hipMemcpyAsync(h, hist, sizeof(h), hipMemcpyDeviceToHost, m->st);
    if (rc == SMX_OK && e == hipSuccess) e = hipStreamSynchronize(m->st);
    if (e != hipSuccess) { set_error(std::string("dataset_corrupt failed: ") + hipGetErrorString(e)); rc = SMX_ERR_HIP; }
    if (rc != SMX_OK) break;
    if (pass == 0) {
      unsigned long long nnz = 0;
      for (int d = 0; d < 256; ++d) nnz += h[d];
      rank = (unsigned long long)floor(dropout * (double)nnz);   // int(np.floor(dropout * len(i))), utils.py:213-215
      if (rank == 0) { nothing = true; break; }
    }
    unsigned long long cum = 0;
    int digit = 255;
    for (int d = 0; d < 256; ++d) {
      if (cum + h[d] >= rank) { digit = d; break; }
      cum += h[d];
    }
    rank -= cum;
    a.prefix |= (uint64_t)digit << (56 - 8 * pass);
  }
  if (rc == SMX_OK && !nothing) {
    hipError_t e = hipMemsetAsync(hist, 0, sizeof(unsigned long long), m->st);
    if (e == hipSuccess) rc = launch_corrupt_apply(m->st, a);
    // the per-row constant sum lgamma(x+1) follows the matrix
    if (rc == SMX_OK) rc = launch_row_stats(m->st, m->X, m->x_u16 ? 1 : 0, m->Gp, m->N, m->G, m->lgx1, nullptr);
    if (rc == SMX_OK && e == hipSuccess) e = hipMemcpyAsync(h, hist, sizeof(unsigned long long), hipMemcpyDeviceToHost, m->st);
    if (rc == SMX_OK && e == hipSuccess) e = hipStreamSynchronize(m->st);
    if (e != hipSuccess) { set_error(std::string("dataset_corrupt failed: ") + hipGetErrorString(e)); rc = SMX_ERR_HIP; }
    if (rc == SMX_OK && n_corrupted) *n_corrupted = (int64_t)h[0];
  }
  hipFree(hist);
  return rc;
}

int smx_dataset_read(smx_model* m, int64_t row0, int64_t n_rows, float* X, float* row_const, float* library) {
  SMX_REQUIRE(m && m->X, "no resident dataset");
  SMX_REQUIRE(row0 >= 0 && n_rows > 0 && row0 + n_rows <= m->N, "rows out of range");
  SMX_HIP(hipStreamSynchronize(m->st));
  if (X && m->x_csr) {   // the sparse store's rows, expanded a tile at a time
    for (int64_t r = 0; r < n_rows; r += m->Bmax) {
      const int B = (int)std::min<int64_t>(m->Bmax, n_rows - r);
      SMX_CHECK(launch_csr_expand(m->st, m->csr_indptr, m->csr_cols, m->csr_vals, nullptr, (long)(row0 + r), B, m->Gp, m->xbatch));
      SMX_HIP(hipMemcpy2DAsync(X + (size_t)r * m->G, (size_t)m->G * sizeof(float), m->xbatch, (size_t)m->Gp * sizeof(float),
                               (size_t)m->G * sizeof(float), (size_t)B, hipMemcpyDeviceToHost, m->st));
      SMX_HIP(hipStreamSynchronize(m->st));
    }
  } else if (X && m->x_u16) {
    std::vector<uint16_t> tmp((size_t)n_rows * m->G);
    SMX_HIP(hipMemcpy2D(tmp.data(), (size_t)m->G * sizeof(uint16_t), reinterpret_cast<const uint16_t*>(m->X) + (size_t)row0 * m->Gp,
                        (size_t)m->Gp * sizeof(uint16_t), (size_t)m->G * sizeof(uint16_t), (size_t)n_rows, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < tmp.size(); ++i) X[i] = (float)tmp[i];
  } else if (X)
    SMX_HIP(hipMemcpy2D(X, (size_t)m->G * sizeof(float), m->X + (size_t)row0 * m->Gp, (size_t)m->Gp * sizeof(float),
                        (size_t)m->G * sizeof(float), (size_t)n_rows, hipMemcpyDeviceToHost));
  if (row_const) SMX_HIP(hipMemcpy(row_const, m->lgx1 + row0, (size_t)n_rows * sizeof(float), hipMemcpyDeviceToHost));
  if (library) {
    SMX_REQUIRE(m->library, "no library prior resident");
    SMX_HIP(hipMemcpy(library, m->library + 2 * row0, (size_t)n_rows * 2 * sizeof(float), hipMemcpyDeviceToHost));
  }
  return SMX_OK;
}

}  // extern "C"
