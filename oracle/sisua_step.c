/* sisua_step.c -- C / OpenMP fp32 port of the VAE training step: the timed CPU baseline.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  This is the "C++/OpenMP fp32 restatement of the
 * identical step" that BASELINE.md section 3 names as the CPU baseline: the same formulas as
 * oracle/sisua_oracle.py (SURVEY.md section 8 rows a-6 ... a-16; reference call sites
 * sisua/models/single_cell_model.py:119-151, configs/base.yaml:45-50), fp32 storage, Philox4x32-10 noise
 * identical to the oracle / the HIP kernels, OpenMP over cells (elementwise work) and over output rows
 * (products).  Scope: the VAE family of the benchmark (model 'vae', nb / zinb / nbd / zinbd, any MLP depth,
 * BatchNorm on or off).  It is validated against the NumPy oracle in tests/test_oracle_cport.py and is
 * never the checker itself.   Build: oracle/build_c.py (gcc -O3 -march=x86-64-v3 -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define OST_MAX_LAYERS 8
#define SP_INV1 0.5413248546129181f

typedef struct {
  int32_t G, D, n_enc, enc[OST_MAX_LAYERS], n_dec, dec[OST_MAX_LAYERS];
  int32_t likelihood; /* 0 nb, 1 zinb, 2 nbd, 3 zinbd */
  int32_t batchnorm, log_norm;
  float dropout_enc, dropout_dec, input_dropout, beta, bn_momentum, bn_eps, lr, b1, b2, adam_eps, clipnorm;
  uint64_t seed;
} ost_config;

typedef struct { int in, out; float *W, *gamma, *beta, *bias; float *mm, *mv;          /* parameters, moving stats */
                 float *gW, *ggamma, *gbeta, *gbias;                                   /* gradients */
                 float *pre, *xhat, *act, *mask, *inv; int stream; float drop; } layer_t;

typedef struct {
  ost_config c; int k, Bcap;
  layer_t enc[OST_MAX_LAYERS], dec[OST_MAX_LAYERS];
  float *Wlat, *blat, *gWlat, *gblat, *Wout, *bout, *gWout, *gbout;
  /* flat views for the optimiser */
  int n_tensors; float* tp[64]; float* tg[64]; float* tm[64]; float* tv[64]; size_t tn[64];
  int t;
  /* every buffer of the model (freed by ost_destroy: found by the AddressSanitizer build, tools/asan_build.sh) */
  float* owned[640]; int n_owned;
  /* activations */
  float *h0, *in_mask, *lat, *z, *sig, *eps, *P, *dP, *dd, *dz, *dlat, *dh, *tmp;
} ost_t;

/* ---- Philox4x32-10, same counter layout as oracle/sisua_oracle.py ------------------------------ */
static inline void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t o[4]) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
static void dropout_row(const ost_t* m, int stream, int step, int64_t cell, int n, float p, float* out) {
  const float scale = 1.0f / (1.0f - p);
  for (int q = 0; q < (n + 3) / 4; ++q) {
    uint32_t w[4];
    philox((uint32_t)q, (uint32_t)cell, (uint32_t)step, (uint32_t)stream, (uint32_t)m->c.seed, (uint32_t)(m->c.seed >> 32), w);
    for (int e = 0; e < 4 && 4 * q + e < n; ++e) out[4 * q + e] = ((float)(w[e] >> 8) * 5.9604644775390625e-08f >= p) ? scale : 0.f;
  }
}
static void normal_row(const ost_t* m, int stream, int step, int64_t cell, int n, float* out) {
  for (int q = 0; q < (n + 3) / 4; ++q) {
    uint32_t w[4];
    float v[4];
    philox((uint32_t)q, (uint32_t)cell, (uint32_t)step, (uint32_t)stream, (uint32_t)m->c.seed, (uint32_t)(m->c.seed >> 32), w);
    for (int pr = 0; pr < 2; ++pr) {
      const double u1 = ((double)(w[2 * pr] >> 8) + 1.0) * 5.9604644775390625e-08, u2 = (double)(w[2 * pr + 1] >> 8) * 5.9604644775390625e-08;
      const double r = sqrt(-2.0 * log(u1));
      v[2 * pr] = (float)(r * cos(6.283185307179586 * u2)); v[2 * pr + 1] = (float)(r * sin(6.283185307179586 * u2));
    }
    for (int e = 0; e < 4 && 4 * q + e < n; ++e) out[4 * q + e] = v[e];
  }
}

/* ---- products -------------------------------------------------------------------------------------
 * Dense products go through the BLAS NumPy itself links when the caller hands its path over (ost_use_blas: OpenBLAS's ILP64 cblas_sgemm from
 * numpy.libs) -- a CPU baseline whose three large products run at library speed on every core is the fair bar; the loops
 * below remain for the products whose left operand is the SPARSE input (log1p of 93 % zeros: they skip the zeros, 14x fewer flops than a
 * dense product) and as the fallback without a BLAS. */
#include <dlfcn.h>
typedef void (*ost_sgemm_fn)(int, int, int, int64_t, int64_t, int64_t, float, const float*, int64_t, const float*, int64_t, float, float*, int64_t);
static ost_sgemm_fn g_sgemm = NULL;
static void (*g_blas_set_threads)(int) = NULL;
int ost_use_blas(const char* path) {   /* 1: dense products through <path>'s scipy_cblas_sgemm64_ / cblas_sgemm64_; 0: not available */
  g_sgemm = NULL; g_blas_set_threads = NULL;
  if (!path) return 0;
  void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return 0;
  const char* names[] = {"scipy_cblas_sgemm64_", "cblas_sgemm64_"};
  for (int i = 0; i < 2 && !g_sgemm; ++i) g_sgemm = (ost_sgemm_fn)dlsym(h, names[i]);
  const char* tn[] = {"scipy_openblas_set_num_threads64_", "openblas_set_num_threads64_"};
  for (int i = 0; i < 2 && !g_blas_set_threads; ++i) g_blas_set_threads = (void (*)(int))dlsym(h, tn[i]);
  if (g_sgemm && g_blas_set_threads) g_blas_set_threads(1);
  return g_sgemm != NULL;
}
#define OST_BLAS_MIN_WORK 200000L   /* below this a product is faster in the loops */
static int use_blas(long M, long K, long N) { return g_sgemm != NULL && M * K * N >= OST_BLAS_MIN_WORK; }
/* ONE thread pool: the library runs single-threaded (ost_set_threads) and the OpenMP threads each take a slice of the product's larger
 * output axis.  (Two pools -- the library's own threads beside OpenMP's -- spin against each other inside a CPU quota: on the GPU boxes'
 * 16-CPU share of a 256-core host the step got SLOWER with every thread added, 7.5 k cells/s at 8 + 8 threads, 1.2 k at 64 + 64.)
 * row-major C[M,N] = op(A) op(B); ta / tb: 111 no transpose, 112 transpose */
static void blas_sliced(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc) {
#pragma omp parallel
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
    const int t = 0, nt = 1;
#endif
    if (N >= M) {   /* slices of columns: multiples of 16 floats */
      const int per = ((N + nt - 1) / nt + 15) / 16 * 16, n0 = t * per, n1 = n0 + per < N ? n0 + per : N;
      if (n0 < n1) g_sgemm(101, ta, tb, M, n1 - n0, K, 1.f, A, lda, tb == 111 ? B + n0 : B + (size_t)n0 * ldb, ldb, 0.f, C + n0, ldc);
    } else {        /* slices of rows */
      const int per = (M + nt - 1) / nt, m0 = t * per, m1 = m0 + per < M ? m0 + per : M;
      if (m0 < m1) g_sgemm(101, ta, tb, m1 - m0, N, K, 1.f, ta == 111 ? A + (size_t)m0 * lda : A + m0, lda, B, ldb, 0.f, C + (size_t)m0 * ldc, ldc);
    }
  }
}

/* (row-parallel, i-k-j so the inner loop vectorises; zeros of A skipped) */
static void gemm_nn_sp(const float* A, const float* B, float* C, int M, int K, int N) { /* C[M,N] = A[M,K] B[K,N] */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < M; ++i) {
    float* c = C + (size_t)i * N;
    memset(c, 0, sizeof(float) * N);
    for (int k = 0; k < K; ++k) { const float a = A[(size_t)i * K + k]; if (a == 0.f) continue; const float* b = B + (size_t)k * N; for (int j = 0; j < N; ++j) c[j] += a * b[j]; }
  }
}
static void gemm_tn_sp(const float* A, const float* B, float* C, int K, int M, int N) { /* C[M,N] = A[K,M]^T B[K,N] */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < M; ++i) {
    float* c = C + (size_t)i * N;
    memset(c, 0, sizeof(float) * N);
    for (int k = 0; k < K; ++k) { const float a = A[(size_t)k * M + i]; if (a == 0.f) continue; const float* b = B + (size_t)k * N; for (int j = 0; j < N; ++j) c[j] += a * b[j]; }
  }
}
static void gemm_nn(const float* A, const float* B, float* C, int M, int K, int N) {
  if (use_blas(M, K, N)) blas_sliced(111, 111, M, N, K, A, K, B, N, C, N); else gemm_nn_sp(A, B, C, M, K, N);
}
static void gemm_tn(const float* A, const float* B, float* C, int K, int M, int N) {
  if (use_blas(M, K, N)) blas_sliced(112, 111, M, N, K, A, M, B, N, C, N); else gemm_tn_sp(A, B, C, K, M, N);
}
static void gemm_nt(const float* A, const float* B, float* C, int M, int K, int N) { /* C[M,N] = A[M,K] B[N,K]^T */
  if (use_blas(M, K, N)) { blas_sliced(111, 112, M, N, K, A, K, B, K, C, N); return; }
#pragma omp parallel for schedule(static) collapse(2)
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < N; ++j) {
      const float *a = A + (size_t)i * K, *b = B + (size_t)j * K;
      float s = 0.f;
      for (int k = 0; k < K; ++k) s += a[k] * b[k];
      C[(size_t)i * N + j] = s;
    }
}

static inline float softplusf_(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
static double digamma_(double x) {
  double r = 0.0;
  while (x < 6.0) { r -= 1.0 / x; x += 1.0; }
  const double f = 1.0 / (x * x);
  return r + log(x) - 0.5 / x - f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f / 132))));
}

/* lgamma(x + r) - lgamma(r) - lgamma(x + 1) and digamma(x + r) - digamma(r).  93 % of the counts are 0 (both exactly 0) and nearly all of
 * the rest small integers, for which both differences are finite sums over the rising factorial: log prod_{i < x} (r + i) and
 * sum_{i < x} 1 / (r + i) -- one log instead of three libm lgamma calls and two digamma series per element (the likelihood was 62 % of a
 * single-threaded step, profiled when the CPU baseline was made a fair bar: VERDICT r03 item 8); libm for everything else. */
static inline void lgdg_diff(double x, double r, double* lg, double* dg) {
  static const double LOGFACT[17] = {0.0, 0.0, 0.693147180559945, 1.7917594692280554, 3.178053830347945, 4.787491742782047, 6.579251212010102, 8.525161361065415, 10.604602902745249, 12.801827480081467, 15.104412573075514, 17.502307845873887, 19.987214495661885, 22.55216385312342, 25.191221182738683, 27.89927138384089, 30.671860106080672};
  if (x == 0.0) { *lg = 0.0; *dg = 0.0; return; }
  if (x <= 16.0 && x == floor(x)) {
    double P = 1.0, s = 0.0;
    for (int i = 0; i < (int)x; ++i) { P *= r + i; s += 1.0 / (r + i); }
    *lg = log(P) - LOGFACT[(int)x]; *dg = s;
    return;
  }
  *lg = lgamma(x + r) - lgamma(r) - lgamma(x + 1.0); *dg = digamma_(x + r) - digamma_(r);
}

/* elementwise count log-likelihood and gradients wrt the raw planes (oracle count_llk) */
static inline float count_elem(int lik, float x, float p0, float p1, float p2, float* d0, float* d1, float* d2) {
  double ell, lg, dg;
  if (lik <= 1) {
    const double r = exp((double)p0), sp = softplusf_(p1);
    lgdg_diff(x, r, &lg, &dg);
    ell = lg + x * (p1 - sp) - r * sp;
    *d0 = (float)(r * (dg - sp));
    *d1 = (float)(x - (x + r) * sigmoidf_(p1));
  } else {
    const double mu = softplusf_(p0), th = softplusf_(p1 + SP_INV1), e = 1e-8, lt = log(th + mu + e);
    lgdg_diff(x, th, &lg, &dg);
    ell = th * (log(th + e) - lt) + x * (log(mu + e) - lt) + lg;
    const double dmu = -th / (th + mu + e) + x / (mu + e) - x / (th + mu + e);
    const double dth = log(th + e) - lt + th / (th + e) - th / (th + mu + e) - x / (th + mu + e) + dg;
    *d0 = (float)(dmu * sigmoidf_(p0)); *d1 = (float)(dth * sigmoidf_(p1 + SP_INV1));
  }
  if (lik == 0 || lik == 2) { *d2 = 0.f; return (float)ell; }
  const double spg = softplusf_(p2), sg = sigmoidf_(p2);
  if (x == 0.f) {
    const double mx = fmax((double)p2, ell), lse = mx + log(exp(p2 - mx) + exp(ell - mx)), w = exp(ell - lse);
    *d0 *= (float)w; *d1 *= (float)w; *d2 = (float)((1.0 - w) - sg);
    return (float)(lse - spg);
  }
  *d2 = (float)(-sg);
  return (float)(ell - spg);
}

static float* falloc(ost_t* m, size_t n) {
  float* p = (float*)calloc(n ? n : 1, sizeof(float));
  if (m->n_owned < (int)(sizeof(m->owned) / sizeof(m->owned[0]))) m->owned[m->n_owned++] = p;
  return p;
}
static void reg(ost_t* m, float* p, float* g, size_t n) { int i = m->n_tensors++; m->tp[i] = p; m->tg[i] = g; m->tn[i] = n; m->tm[i] = falloc(m, n); m->tv[i] = falloc(m, n); }

static void init_mlp(ost_t* m, layer_t* L, int n, const int32_t* units, int n_in, int stream0, float drop, const float* const** pp) {
  for (int i = 0; i < n; ++i) {
    layer_t* l = &L[i];
    l->in = n_in; l->out = units[i]; l->stream = stream0 + i; l->drop = drop;
    const size_t nw = (size_t)n_in * units[i];
    l->W = falloc(m, nw); memcpy(l->W, *(*pp)++, nw * sizeof(float)); l->gW = falloc(m, nw); reg(m, l->W, l->gW, nw);
    if (m->c.batchnorm) {
      l->gamma = falloc(m, units[i]); memcpy(l->gamma, *(*pp)++, units[i] * sizeof(float)); l->ggamma = falloc(m, units[i]); reg(m, l->gamma, l->ggamma, units[i]);
      l->beta = falloc(m, units[i]); memcpy(l->beta, *(*pp)++, units[i] * sizeof(float)); l->gbeta = falloc(m, units[i]); reg(m, l->beta, l->gbeta, units[i]);
      l->mm = falloc(m, units[i]); l->mv = falloc(m, units[i]); for (int j = 0; j < units[i]; ++j) l->mv[j] = 1.f;
    } else {
      l->bias = falloc(m, units[i]); memcpy(l->bias, *(*pp)++, units[i] * sizeof(float)); l->gbias = falloc(m, units[i]); reg(m, l->bias, l->gbias, units[i]);
    }
    l->inv = falloc(m, units[i]);
    n_in = units[i];
  }
}

void* ost_create(const ost_config* c, const float* const* params) {
  ost_t* m = (ost_t*)calloc(1, sizeof(ost_t));
  m->c = *c; m->k = (c->likelihood == 1 || c->likelihood == 3) ? 3 : 2;
  const float* const* pp = params;
  init_mlp(m, m->enc, c->n_enc, c->enc, c->G, 16, c->dropout_enc, &pp);
  const int H = c->enc[c->n_enc - 1], D = c->D;
  m->Wlat = falloc(m, (size_t)H * 2 * D); memcpy(m->Wlat, *pp++, (size_t)H * 2 * D * sizeof(float)); m->gWlat = falloc(m, (size_t)H * 2 * D); reg(m, m->Wlat, m->gWlat, (size_t)H * 2 * D);
  m->blat = falloc(m, 2 * D); memcpy(m->blat, *pp++, 2 * D * sizeof(float)); m->gblat = falloc(m, 2 * D); reg(m, m->blat, m->gblat, 2 * D);
  init_mlp(m, m->dec, c->n_dec, c->dec, D, 48, c->dropout_dec, &pp);
  const int Hd = c->dec[c->n_dec - 1]; const size_t kg = (size_t)m->k * c->G;
  m->Wout = falloc(m, Hd * kg); memcpy(m->Wout, *pp++, Hd * kg * sizeof(float)); m->gWout = falloc(m, Hd * kg); reg(m, m->Wout, m->gWout, Hd * kg);
  m->bout = falloc(m, kg); memcpy(m->bout, *pp++, kg * sizeof(float)); m->gbout = falloc(m, kg); reg(m, m->bout, m->gbout, kg);
  return m;
}

static void ensure(ost_t* m, int B) {
  if (B <= m->Bcap) return;
  m->Bcap = B;
  const int G = m->c.G, D = m->c.D; int maxw = 2 * D;
  for (int i = 0; i < m->c.n_enc; ++i) if (m->c.enc[i] > maxw) maxw = m->c.enc[i];
  for (int i = 0; i < m->c.n_dec; ++i) if (m->c.dec[i] > maxw) maxw = m->c.dec[i];
  layer_t* Ls[2] = {m->enc, m->dec}; const int ns[2] = {m->c.n_enc, m->c.n_dec};
  for (int q = 0; q < 2; ++q) for (int i = 0; i < ns[q]; ++i) { layer_t* l = &Ls[q][i]; const size_t n = (size_t)B * l->out; l->pre = falloc(m, n); l->xhat = falloc(m, n); l->act = falloc(m, n); l->mask = falloc(m, n); }
  m->h0 = falloc(m, (size_t)B * G); m->in_mask = falloc(m, (size_t)B * G); m->lat = falloc(m, (size_t)B * 2 * D); m->z = falloc(m, (size_t)B * D); m->sig = falloc(m, (size_t)B * D); m->eps = falloc(m, (size_t)B * D);
  m->P = falloc(m, (size_t)B * m->k * G); m->dP = falloc(m, (size_t)B * m->k * G); m->dd = falloc(m, (size_t)B * maxw); m->dz = falloc(m, (size_t)B * maxw); m->dlat = falloc(m, (size_t)B * 2 * D); m->dh = falloc(m, (size_t)B * maxw); m->tmp = falloc(m, (size_t)B * maxw);
}

static void mlp_fwd(ost_t* m, layer_t* L, int n, const float* in, int B, int step, const int64_t* cells) {
  for (int i = 0; i < n; ++i) {
    layer_t* l = &L[i]; const int N = l->out;
    if (L == m->enc && i == 0) gemm_nn_sp(in, l->W, l->pre, B, l->in, N);   /* the sparse counts: the zero-skipping loop */
    else gemm_nn(in, l->W, l->pre, B, l->in, N);
    if (m->c.batchnorm) {
#pragma omp parallel for schedule(static)
      for (int j = 0; j < N; ++j) {
        double s = 0, s2 = 0;
        for (int b = 0; b < B; ++b) s += l->pre[(size_t)b * N + j];
        const double mean = s / B;
        for (int b = 0; b < B; ++b) { const double d = l->pre[(size_t)b * N + j] - mean; s2 += d * d; }
        const double var = s2 / B; const float inv = (float)(1.0 / sqrt(var + m->c.bn_eps));
        l->inv[j] = inv;
        l->mm[j] = l->mm[j] * m->c.bn_momentum + (float)mean * (1.f - m->c.bn_momentum);
        l->mv[j] = l->mv[j] * m->c.bn_momentum + (float)var * (1.f - m->c.bn_momentum);
        for (int b = 0; b < B; ++b) l->xhat[(size_t)b * N + j] = (l->pre[(size_t)b * N + j] - (float)mean) * inv;
      }
    }
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
      float* mk = l->mask + (size_t)b * N;
      if (l->drop > 0.f) dropout_row(m, l->stream, step, cells[b], N, l->drop, mk); else for (int j = 0; j < N; ++j) mk[j] = 1.f;
      for (int j = 0; j < N; ++j) {
        const float y = m->c.batchnorm ? l->gamma[j] * l->xhat[(size_t)b * N + j] + l->beta[j] : l->pre[(size_t)b * N + j] + l->bias[j];
        l->act[(size_t)b * N + j] = fmaxf(y, 0.f) * mk[j];
      }
    }
    in = l->act;
  }
}

/* dh: d loss / d output of the last layer [B][out]; returns d input in m->tmp-sized buffer `din` */
static void mlp_bwd(ost_t* m, layer_t* L, int n, const float* in0, float* dh, float* din, int B) {
  for (int i = n - 1; i >= 0; --i) {
    layer_t* l = &L[i]; const int N = l->out; const float* in = i == 0 ? in0 : L[i - 1].act;
    float* dpre = l->pre; /* reuse */
#pragma omp parallel for schedule(static)
    for (int j = 0; j < N; ++j) {
      double s1 = 0, s2 = 0;
      for (int b = 0; b < B; ++b) {
        const size_t o = (size_t)b * N + j;
        const float dy = (l->act[o] > 0.f) ? dh[o] * l->mask[o] : 0.f;
        dh[o] = dy; s1 += dy; if (m->c.batchnorm) s2 += dy * l->xhat[o];
      }
      if (m->c.batchnorm) {
        l->ggamma[j] = (float)s2; l->gbeta[j] = (float)s1;
        for (int b = 0; b < B; ++b) { const size_t o = (size_t)b * N + j; dpre[o] = l->gamma[j] * l->inv[j] * (dh[o] - (float)((s1 + l->xhat[o] * s2) / B)); }
      } else {
        l->gbias[j] = (float)s1;
        for (int b = 0; b < B; ++b) dpre[(size_t)b * N + j] = dh[(size_t)b * N + j];
      }
    }
    if (L == m->enc && i == 0) gemm_tn_sp(in, dpre, l->gW, B, l->in, N);
    else gemm_tn(in, dpre, l->gW, B, l->in, N);
    if (i > 0 || din) { float* dst = i > 0 ? dh : din; gemm_nt(dpre, l->W, dst, B, N, l->in); }
  }
}

float ost_train_step(void* h, const float* x, const int64_t* cells, int B, int step) {
  ost_t* m = (ost_t*)h; ensure(m, B);
  const ost_config* c = &m->c; const int G = c->G, D = c->D, k = m->k; const size_t kg = (size_t)k * G;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b) {
    float* mk = m->in_mask + (size_t)b * G;
    if (c->input_dropout > 0.f) dropout_row(m, 0, step, cells[b], G, c->input_dropout, mk); else for (int g = 0; g < G; ++g) mk[g] = 1.f;
    for (int g = 0; g < G; ++g) { const float v = x[(size_t)b * G + g]; m->h0[(size_t)b * G + g] = (c->log_norm ? log1pf(v) : v) * mk[g]; }
  }
  mlp_fwd(m, m->enc, c->n_enc, m->h0, B, step, cells);
  const layer_t* eL = &m->enc[c->n_enc - 1]; const int H = eL->out;
  gemm_nn(eL->act, m->Wlat, m->lat, B, H, 2 * D);
  double kl_sum = 0;
#pragma omp parallel for schedule(static) reduction(+ : kl_sum)
  for (int b = 0; b < B; ++b) {
    normal_row(m, 64, step, cells[b], D, m->eps + (size_t)b * D);
    for (int d = 0; d < D; ++d) {
      const float mu = m->lat[(size_t)b * 2 * D + d] + m->blat[d], sg = softplusf_(m->lat[(size_t)b * 2 * D + D + d] + m->blat[D + d] + SP_INV1);
      m->lat[(size_t)b * 2 * D + d] = mu; m->sig[(size_t)b * D + d] = sg;
      m->z[(size_t)b * D + d] = mu + sg * m->eps[(size_t)b * D + d];
      kl_sum += 0.5 * ((double)sg * sg + (double)mu * mu - 1.0 - 2.0 * log(sg));
    }
  }
  mlp_fwd(m, m->dec, c->n_dec, m->z, B, step, cells);
  const layer_t* dL = &m->dec[c->n_dec - 1]; const int Hd = dL->out;
  gemm_nn(dL->act, m->Wout, m->P, B, Hd, (int)kg);
  double llk_sum = 0; const float cx = -1.0f / B;
#pragma omp parallel for schedule(static) reduction(+ : llk_sum)
  for (int b = 0; b < B; ++b) {
    float* p = m->P + (size_t)b * kg; float* dp = m->dP + (size_t)b * kg;
    for (int g = 0; g < G; ++g) {
      float d0, d1, d2;
      const float p2 = k == 3 ? p[2 * G + g] + m->bout[2 * G + g] : 0.f;
      llk_sum += count_elem(c->likelihood, x[(size_t)b * G + g], p[g] + m->bout[g], p[G + g] + m->bout[G + g], p2, &d0, &d1, &d2);
      dp[g] = d0 * cx; dp[G + g] = d1 * cx; if (k == 3) dp[2 * G + g] = d2 * cx;
    }
  }
  const float loss = (float)(-(llk_sum - c->beta * kl_sum) / B);
  /* ---- backward ---- */
  gemm_tn(dL->act, m->dP, m->gWout, B, Hd, (int)kg);
#pragma omp parallel for schedule(static)
  for (size_t j = 0; j < kg; ++j) { double s = 0; for (int b = 0; b < B; ++b) s += m->dP[(size_t)b * kg + j]; m->gbout[j] = (float)s; }
  gemm_nt(m->dP, m->Wout, m->dd, B, (int)kg, Hd);
  mlp_bwd(m, m->dec, c->n_dec, m->z, m->dd, m->dz, B);
  const float ckl = c->beta / B;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b)
    for (int d = 0; d < D; ++d) {
      const float mu = m->lat[(size_t)b * 2 * D + d], sraw = m->lat[(size_t)b * 2 * D + D + d] + m->blat[D + d], sg = m->sig[(size_t)b * D + d], dz = m->dz[(size_t)b * D + d];
      m->dlat[(size_t)b * 2 * D + d] = dz + ckl * mu;
      m->dlat[(size_t)b * 2 * D + D + d] = (dz * m->eps[(size_t)b * D + d] + ckl * (sg - 1.f / sg)) * sigmoidf_(sraw + SP_INV1);
    }
  gemm_tn(eL->act, m->dlat, m->gWlat, B, H, 2 * D);
  for (int j = 0; j < 2 * D; ++j) { double s = 0; for (int b = 0; b < B; ++b) s += m->dlat[(size_t)b * 2 * D + j]; m->gblat[j] = (float)s; }
  gemm_nt(m->dlat, m->Wlat, m->dh, B, 2 * D, H);
  mlp_bwd(m, m->enc, c->n_enc, m->h0, m->dh, NULL, B);
  /* ---- per-tensor clipnorm + Adam ---- */
  m->t += 1;
  const double lr_t = c->lr * sqrt(1.0 - pow(c->b2, m->t)) / (1.0 - pow(c->b1, m->t));
  for (int i = 0; i < m->n_tensors; ++i) {
    double n2 = 0; const size_t n = m->tn[i]; float *g = m->tg[i], *p = m->tp[i], *mm = m->tm[i], *vv = m->tv[i];
#pragma omp parallel for schedule(static) reduction(+ : n2)
    for (size_t j = 0; j < n; ++j) n2 += (double)g[j] * g[j];
    const double nrm = sqrt(n2); const float clip = (c->clipnorm > 0 && nrm > c->clipnorm) ? (float)(c->clipnorm / nrm) : 1.f;
#pragma omp parallel for schedule(static)
    for (size_t j = 0; j < n; ++j) {
      const float gj = g[j] * clip;
      mm[j] = c->b1 * mm[j] + (1.f - c->b1) * gj; vv[j] = c->b2 * vv[j] + (1.f - c->b2) * gj * gj;
      p[j] -= (float)lr_t * mm[j] / (sqrtf(vv[j]) + c->adam_eps);
    }
  }
  return loss;
}

int ost_num_tensors(void* h) { return ((ost_t*)h)->n_tensors; }
long ost_tensor_size(void* h, int i) { return (long)((ost_t*)h)->tn[i]; }
void ost_get_param(void* h, int i, float* out) { ost_t* m = (ost_t*)h; memcpy(out, m->tp[i], m->tn[i] * sizeof(float)); }
void ost_get_grad(void* h, int i, float* out) { ost_t* m = (ost_t*)h; memcpy(out, m->tg[i], m->tn[i] * sizeof(float)); }
int ost_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void ost_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
  if (g_blas_set_threads) g_blas_set_threads(1);   /* (one pool: see blas_sliced) */
}
void ost_destroy(void* h) {
  ost_t* m = (ost_t*)h;
  if (!m) return;
  for (int i = 0; i < m->n_owned; ++i) free(m->owned[i]);
  free(m);
}
