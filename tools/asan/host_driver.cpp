// host_driver.cpp -- the HOST-ONLY entry points of libsisua_hip.so under AddressSanitizer + UBSan ("Race detection /
// sanitizers").  Built and run by tools/asan_build.sh on the CPU: the library's translation units are compiled with
// `hipcc --cuda-host-only -fsanitize=address,undefined` (no device code, no GPU needed) and this program walks what can run without a device:
// the ABI queries, smx_shuffle_order, every argument / configuration check of smx_model_create up to its first device call, and the error paths
// of the accessors on a null model.  Any sanitizer report aborts with a non-zero status; "HOST DRIVER OK" is printed last.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/sisua_hip.h"

static int failures = 0;
#define EXPECT(cond) do { if (!(cond)) { printf("FAILED line %d: %s (last error: %s)\n", __LINE__, #cond, smx_last_error()); ++failures; } } while (0)

static smx_config good_config() {
  smx_config c;
  memset(&c, 0, sizeof(c));
  c.abi_version = SMX_ABI_VERSION; c.model = SMX_MODEL_VAE; c.likelihood = SMX_LLK_ZINB; c.n_genes = 200; c.latent_dim = 10;
  c.n_enc = 2; c.enc_units[0] = 64; c.enc_units[1] = 32; c.n_dec = 1; c.dec_units[0] = 48;
  c.batchnorm = 1; c.log_norm = 1; c.dropout_enc = 0.1f; c.dropout_dec = 0.1f; c.beta = 1.f; c.alpha = 10.f; c.clip_library = 1e3f;
  c.bn_momentum = 0.99f; c.bn_eps = 1e-3f; c.lr = 1e-3f; c.adam_beta1 = 0.9f; c.adam_beta2 = 0.999f; c.adam_eps = 1e-7f; c.clipnorm = 100.f;
  c.max_batch = 64; c.seed = 8; c.n_components = 10; c.disc_units = 100; c.disc_layers = 2; c.gamma = 6.f; c.disc_leak = 0.2f;
  return c;
}

int main() {
  EXPECT(smx_abi_version() == SMX_ABI_VERSION);
  EXPECT(smx_last_error() != nullptr);
  const int n_dev = smx_device_count();
  printf("devices visible: %d\n", n_dev);
  EXPECT(smx_init(-1) != SMX_OK);
  EXPECT(smx_init(n_dev + 5) != SMX_OK && strlen(smx_last_error()) > 0);

  // ---- smx_shuffle_order: the streaming shuffle of create_dataset (buffer larger / smaller than the data, empty data, bad picks)
  for (int n : {0, 1, 7, 1000, 3381}) for (int buffer : {1, 5, 1000, 5000}) {
    std::vector<int64_t> picks((size_t)n);
    std::vector<int32_t> out((size_t)n, -1);
    uint64_t s = 12345u + (uint64_t)n * 31u + (uint64_t)buffer;
    for (auto& p : picks) { s = s * 6364136223846793005ull + 1442695040888963407ull; p = (int64_t)(s >> 33); }
    EXPECT(smx_shuffle_order(n, buffer, picks.data(), out.data()) == SMX_OK);
    std::vector<char> seen((size_t)n, 0);
    for (int v : out) { EXPECT(v >= 0 && v < n && !seen[(size_t)v]); if (v >= 0 && v < n) seen[(size_t)v] = 1; }   // a permutation
  }
  { int64_t bad[3] = {1, -4, 2}; int32_t out[3]; EXPECT(smx_shuffle_order(3, 2, bad, out) != SMX_OK); }
  EXPECT(smx_shuffle_order(3, 0, nullptr, nullptr) != SMX_OK);
  EXPECT(smx_shuffle_order(-1, 4, nullptr, nullptr) != SMX_OK);

  // ---- smx_model_create: every check in front of the first device call
  smx_model* m = nullptr;
  smx_config c = good_config();
  EXPECT(smx_model_create(nullptr, &m) != SMX_OK && smx_model_create(&c, nullptr) != SMX_OK);
  auto refused = [&](const char* what, smx_config cfg) {
    smx_model* mm = nullptr;
    const int rc = smx_model_create(&cfg, &mm);
    if (rc == SMX_OK) { printf("NOT refused: %s\n", what); ++failures; smx_model_destroy(mm); }
    else if (rc != SMX_ERR_INVALID) { printf("%s: expected SMX_ERR_INVALID, got %d (%s)\n", what, rc, smx_last_error()); ++failures; }
  };
  { smx_config b = c; b.abi_version = SMX_ABI_VERSION - 1; refused("abi", b); }
  { smx_config b = c; b.n_genes = 0; refused("n_genes", b); }
  { smx_config b = c; b.latent_dim = -3; refused("latent_dim", b); }
  { smx_config b = c; b.max_batch = 0; refused("max_batch", b); }
  { smx_config b = c; b.n_enc = 0; refused("n_enc", b); }
  { smx_config b = c; b.n_dec = SMX_MAX_LAYERS + 1; refused("n_dec", b); }
  { smx_config b = c; b.model = 99; refused("model kind", b); }
  { smx_config b = c; b.likelihood = 17; refused("likelihood", b); }
  { smx_config b = c; b.n_labels = SMX_MAX_LABELS + 1; refused("n_labels", b); }
  { smx_config b = c; b.n_labels = 1; b.label_dim[0] = 5; b.label_llk[0] = SMX_LABEL_NB; refused("label heads on a VAE", b); }
  { smx_config b = c; b.n_labels = 2; b.label_dim[0] = 5; b.label_dim[1] = 4; b.label_observed[1] = 1; refused("observed head behind a label head", b); }
  // (an extra output on FVAE is built since round 5, several one-hot label variables behind it since round 6; what stays refused: more than 32
  // classes in all, a label head on the mixture-density posterior)
  { smx_config b = c; b.model = SMX_MODEL_FVAE; b.disc_layers = 2; b.disc_units = 8; b.n_labels = 3; b.label_dim[0] = 5; b.label_llk[0] = SMX_LABEL_NBD; b.label_observed[0] = 1;
    b.label_dim[1] = 20; b.label_llk[1] = SMX_LABEL_ONEHOT; b.label_dim[2] = 13; b.label_llk[2] = SMX_LABEL_ONEHOT; refused("33 classes of label variables on FVAE", b); }
  { smx_config b = c; b.model = SMX_MODEL_FVAE; b.disc_layers = 2; b.disc_units = 8; b.n_labels = 2; b.label_dim[0] = 3; b.label_llk[0] = SMX_LABEL_ONEHOT;
    b.label_dim[1] = 1; b.label_llk[1] = SMX_LABEL_ONEHOT; refused("a one-class label variable on FVAE", b); }
  { smx_config b = c; b.model = SMX_MODEL_SCALE_POST; b.n_components = 3; b.n_labels = 1; b.label_dim[0] = 5; b.label_llk[0] = SMX_LABEL_NB; refused("label head on the mixture-density posterior", b); }
  { smx_config b = c; b.model = SMX_MODEL_FVAE; b.disc_layers = 0; refused("fvae discriminator depth", b); }
  { smx_config b = c; b.model = SMX_MODEL_FVAE; b.disc_leak = 1.5f; refused("fvae leak", b); }
  { smx_config b = c; b.model = SMX_MODEL_FVAE; b.n_labels = 1; b.label_llk[0] = SMX_LABEL_NB; b.label_dim[0] = 4; refused("fvae label kind", b); }
  { smx_config b = c; b.model = SMX_MODEL_SCALE; b.n_components = 1; refused("scale components", b); }
  { smx_config b = c; b.model = SMX_MODEL_SCALE_TRIL; b.latent_dim = 40; refused("scale tril latent width", b); }
  { smx_config b = c; b.model = SMX_MODEL_SCALE_POST; b.n_components = 12; refused("scale posterior components", b); }
  { smx_config b = c; b.model = SMX_MODEL_SCVI; b.likelihood = SMX_LLK_ZINB; b.n_encl = 1; b.encl_units[0] = 16; refused("scvi likelihood", b); }
  { smx_config b = c; b.model = SMX_MODEL_SCVI; b.likelihood = SMX_LLK_ZINBD; b.n_encl = 0; refused("scvi library encoder", b); }
  { smx_config b = c; b.scvi_dispersion = 1; refused("dispersion option on a VAE", b); }
  { smx_config b = c; b.model = SMX_MODEL_SCVI; b.likelihood = SMX_LLK_ZINBD; b.n_encl = 1; b.encl_units[0] = 16; b.scvi_inflation = 7; refused("scvi inflation value", b); }
  { smx_config b = c; b.dropout_enc = 1.0f; refused("dropout rate", b); }
  { smx_config b = c; b.input_dropout = -0.1f; refused("input dropout rate", b); }
  // a VALID configuration: with a GPU the model is built and destroyed (every buffer through the sanitizer's allocator checks on the host side);
  // without one the first device call fails cleanly (SMX_ERR_HIP), nothing leaks, nothing is half-built
  {
    const int rc = smx_model_create(&c, &m);
    if (n_dev > 0) { EXPECT(rc == SMX_OK && m != nullptr); if (rc == SMX_OK) { EXPECT(smx_num_tensors(m) > 0); EXPECT(smx_model_destroy(m) == SMX_OK); } }
    else EXPECT(rc == SMX_ERR_HIP && strlen(smx_last_error()) > 0);
  }
  // ---- accessors on a null model
  EXPECT(smx_model_destroy(nullptr) == SMX_OK);
  EXPECT(smx_num_tensors(nullptr) == 0);
  { char name[8]; int32_t r, cc; EXPECT(smx_tensor_info(nullptr, 0, name, 8, &r, &cc) != SMX_OK); }
  EXPECT(smx_dataset_size(nullptr) <= 0);
  { int32_t st = 0; EXPECT(smx_get_step(nullptr, &st) != SMX_OK); }
  if (failures) { printf("HOST DRIVER: %d failure(s)\n", failures); return 1; }
  printf("HOST DRIVER OK\n");
  return 0;
}
