/* cstep_driver.c -- the C / OpenMP port of the step (oracle/sisua_step.c, the timed CPU baseline) under AddressSanitizer + UBSan
 * (tools/asan_build.sh): three optimiser steps of every likelihood with and without BatchNorm on a small synthetic matrix, then the
 * accessors.  A sanitizer report aborts; "CSTEP DRIVER OK" is printed last.  TEST INFRASTRUCTURE ONLY. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define OST_MAX_LAYERS 8
typedef struct {
  int32_t G, D, n_enc, enc[OST_MAX_LAYERS], n_dec, dec[OST_MAX_LAYERS];
  int32_t likelihood;
  int32_t batchnorm, log_norm;
  float dropout_enc, dropout_dec, input_dropout, beta, bn_momentum, bn_eps, lr, b1, b2, adam_eps, clipnorm;
  uint64_t seed;
} ost_config;
void* ost_create(const ost_config* c, const float* const* params);
float ost_train_step(void* h, const float* x, const int64_t* cells, int B, int step);
int ost_num_tensors(void* h);
long ost_tensor_size(void* h, int i);
void ost_get_param(void* h, int i, float* out);
void ost_get_grad(void* h, int i, float* out);
void ost_set_threads(int n);
void ost_destroy(void* h);

static uint64_t rng = 88172645463325252ull;
static float unif(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (float)((rng >> 11) * (1.0 / 9007199254740992.0)); }

int main(void) {
  const int G = 53, D = 5, B = 24, N = 96;
  int fails = 0;
  for (int lik = 0; lik < 4; ++lik) for (int bn = 0; bn < 2; ++bn) {
    ost_config c;
    memset(&c, 0, sizeof(c));
    c.G = G; c.D = D; c.n_enc = 2; c.enc[0] = 20; c.enc[1] = 12; c.n_dec = 1; c.dec[0] = 16; c.likelihood = lik; c.batchnorm = bn; c.log_norm = 1;
    c.dropout_enc = 0.15f; c.dropout_dec = 0.1f; c.input_dropout = 0.2f; c.beta = 1.f; c.bn_momentum = 0.99f; c.bn_eps = 1e-3f; c.lr = 1e-3f;
    c.b1 = 0.9f; c.b2 = 0.999f; c.adam_eps = 1e-7f; c.clipnorm = 100.f; c.seed = 8;
    /* tensors in manifest order: enc{i}/W (+ gamma, beta | b), lat/W, lat/b, dec{i}/W (...), out/W, out/b */
    const int k = (lik == 1 || lik == 3) ? 3 : 2;
    long sizes[32]; int nt = 0, in = G;
    for (int i = 0; i < c.n_enc; ++i) { sizes[nt++] = (long)in * c.enc[i]; if (bn) { sizes[nt++] = c.enc[i]; sizes[nt++] = c.enc[i]; } else sizes[nt++] = c.enc[i]; in = c.enc[i]; }
    sizes[nt++] = (long)in * 2 * D; sizes[nt++] = 2 * D;
    in = D;
    for (int i = 0; i < c.n_dec; ++i) { sizes[nt++] = (long)in * c.dec[i]; if (bn) { sizes[nt++] = c.dec[i]; sizes[nt++] = c.dec[i]; } else sizes[nt++] = c.dec[i]; in = c.dec[i]; }
    sizes[nt++] = (long)in * k * G; sizes[nt++] = (long)k * G;
    float* bufs[32]; const float* ptrs[32];
    for (int t = 0; t < nt; ++t) {
      bufs[t] = (float*)malloc(sizeof(float) * (size_t)sizes[t]);
      for (long j = 0; j < sizes[t]; ++j) bufs[t][j] = 0.2f * (unif() - 0.5f) + ((bn && sizes[t] < 64 && (t % 3) == 1) ? 1.f : 0.f);
      ptrs[t] = bufs[t];
    }
    void* h = ost_create(&c, ptrs);
    if (!h) { printf("ost_create failed (lik %d bn %d)\n", lik, bn); ++fails; continue; }
    if (ost_num_tensors(h) != nt) { printf("tensor count %d != %d\n", ost_num_tensors(h), nt); ++fails; }
    for (int t = 0; t < nt && t < ost_num_tensors(h); ++t) if (ost_tensor_size(h, t) != sizes[t]) { printf("tensor %d size %ld != %ld\n", t, ost_tensor_size(h, t), sizes[t]); ++fails; }
    float* X = (float*)malloc(sizeof(float) * N * G);
    for (int i = 0; i < N * G; ++i) { const float u = unif(); X[i] = u < 0.7f ? 0.f : floorf(1.f + 30.f * unif() * unif()); }
    X[5] = 300.f;
    float* xb = (float*)malloc(sizeof(float) * B * G);
    int64_t cells[24];
    for (int threads = 1; threads <= 3; threads += 2) {
      ost_set_threads(threads);
      for (int step = 0; step < 3; ++step) {
        for (int b = 0; b < B; ++b) { cells[b] = (step * B + b) % N; memcpy(xb + (size_t)b * G, X + (size_t)cells[b] * G, sizeof(float) * G); }
        const float loss = ost_train_step(h, xb, cells, B, step);
        if (!(loss == loss) || loss <= 0.f || loss > 1e6f) { printf("loss %g (lik %d bn %d step %d)\n", loss, lik, bn, step); ++fails; }
      }
    }
    { const float l1 = ost_train_step(h, xb, cells, 1, 7); if (!(l1 == l1)) { printf("batch of one: %g\n", l1); ++fails; } }   /* a one-cell batch */
    for (int t = 0; t < nt; ++t) {
      float* out = (float*)malloc(sizeof(float) * (size_t)sizes[t]);
      ost_get_param(h, t, out); ost_get_grad(h, t, out);
      free(out);
    }
    ost_destroy(h);
    for (int t = 0; t < nt; ++t) free(bufs[t]);
    free(X); free(xb);
  }
  if (fails) { printf("CSTEP DRIVER: %d failure(s)\n", fails); return 1; }
  printf("CSTEP DRIVER OK\n");
  return 0;
}
