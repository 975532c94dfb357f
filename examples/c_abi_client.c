/* A plain-C client of the victor_hip C ABI (include/victor_hip.h): no Python, no C++ types, no GPU headers.
 *
 *   cc -O2 -I include examples/c_abi_client.c -ldl -o c_abi_client
 *   ./c_abi_client libvictor_hip.so tables.bin params.bin out.bin rescale_from_ap assume_isotropic like_form nmocks nparams
 *
 * tables.bin is the record stream written by victor_amd.engine.dump_tables (name[24], kind, count, payload);
 * params.bin holds n rows of VK_NPAR doubles; out.bin receives lnL[n] then chi2[n].
 * This is what a binding in any other host language does: fill vk_tables with plain pointers and sizes, vk_create,
 * vk_eval_batch, vk_destroy.
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "victor_hip.h"

typedef struct { const char* name; int kind; void* dst; } field_t;   /* kind 0: int32, 1: double, 2: f64 array, 3: u16 array */

static void* xmalloc(size_t n) {
  void* p = malloc(n ? n : 1);
  if (!p) { fprintf(stderr, "out of memory\n"); exit(2); }
  return p;
}

int main(int argc, char** argv) {
  if (argc < 10) {
    fprintf(stderr, "usage: %s lib tables.bin params.bin out.bin rescale_from_ap assume_isotropic like_form nmocks nparams\n", argv[0]);
    return 2;
  }
  void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  int (*abi)(void) = (int (*)(void))dlsym(lib, "vk_abi_version");
  vk_ctx* (*create)(const vk_tables*, int, char*, size_t) = (vk_ctx * (*)(const vk_tables*, int, char*, size_t)) dlsym(lib, "vk_create");
  void (*destroy)(vk_ctx*) = (void (*)(vk_ctx*))dlsym(lib, "vk_destroy");
  void (*defaults)(vk_eval_opts*) = (void (*)(vk_eval_opts*))dlsym(lib, "vk_default_opts");
  int (*eval)(vk_ctx*, const vk_eval_opts*, const double*, int64_t, double*, double*, double*) =
      (int (*)(vk_ctx*, const vk_eval_opts*, const double*, int64_t, double*, double*, double*))dlsym(lib, "vk_eval_batch");
  const char* (*last_error)(const vk_ctx*) = (const char* (*)(const vk_ctx*))dlsym(lib, "vk_last_error");
  if (!abi || !create || !destroy || !defaults || !eval || !last_error) { fprintf(stderr, "missing symbols\n"); return 2; }
  if (abi() != VK_ABI_VERSION) { fprintf(stderr, "ABI %d, header %d\n", abi(), VK_ABI_VERSION); return 2; }

  vk_tables t;
  memset(&t, 0, sizeof t);
  const field_t fields[] = {
      {"n_s", 0, &t.n_s}, {"n_mu", 0, &t.n_mu}, {"n_x", 0, &t.n_x}, {"n_ell", 0, &t.n_ell},
      {"s", 2, &t.s}, {"mu", 2, &t.mu}, {"w_ell", 2, &t.w_ell}, {"x", 2, &t.x}, {"w_x", 2, &t.w_x},
      {"n_ell_r", 0, &t.n_ell_r}, {"n_beta_r", 0, &t.n_beta_r}, {"beta_r", 2, &t.beta_r},
      {"xi.n_int", 0, &t.xi.n_int}, {"xi.lead", 0, &t.xi.lead}, {"xi.inv_h", 1, &t.xi.inv_h},
      {"xi.knots", 2, &t.xi.knots}, {"xi.coef", 2, &t.xi.coef},
      {"matter_model", 0, &t.matter_model}, {"vr_beta_dep", 0, &t.vr_beta_dep},
      {"vr.n_int", 0, &t.vr.n_int}, {"vr.lead", 0, &t.vr.lead}, {"vr.inv_h", 1, &t.vr.inv_h},
      {"vr.knots", 2, &t.vr.knots}, {"vr.coef", 2, &t.vr.coef}, {"vr_emp", 2, &t.vr_emp}, {"vt_amp", 1, &t.vt_amp},
      {"sv.n_int", 0, &t.sv.n_int}, {"sv.lead", 0, &t.sv.lead}, {"sv.inv_h", 1, &t.sv.inv_h},
      {"sv.knots", 2, &t.sv.knots}, {"sv.coef", 2, &t.sv.coef},
      {"sv_n_mu", 0, &t.sv_n_mu}, {"sv_mu_inv_h", 1, &t.sv_mu_inv_h}, {"sv_mu", 2, &t.sv_mu}, {"sv2d", 2, &t.sv2d},
      {"uni_n", 0, &t.uni_n}, {"uni_u0", 1, &t.uni_u0}, {"uni_inv_h", 1, &t.uni_inv_h},
      {"uni_sv_v", 2, &t.uni_sv_v}, {"uni_xi", 2, &t.uni_xi}, {"uni_xic", 2, &t.uni_xic}, {"uni_vb", 2, &t.uni_vb},
      {"uni_v2", 2, &t.uni_v2}, {"uni_da", 2, &t.uni_da}, {"uni_ge", 2, &t.uni_ge},
      {"uni_dab", 2, &t.uni_dab}, {"uni_empb", 2, &t.uni_empb},
      {"uni_lut_n", 0, &t.uni_lut_n}, {"uni_lut_inv_g", 1, &t.uni_lut_inv_g}, {"uni_lut", 3, &t.uni_lut},
      {"uni_knots", 2, &t.uni_knots},
      {"iaH", 1, &t.iaH}, {"template_sigma8", 1, &t.template_sigma8},
      {"n_beta_d", 0, &t.n_beta_d}, {"beta_d", 2, &t.beta_d}, {"data", 2, &t.data},
      {"n_beta_c", 0, &t.n_beta_c}, {"beta_c", 2, &t.beta_c}, {"prec", 2, &t.prec}, {"logdet", 2, &t.logdet},
      {"eig", 2, &t.eig},
  };
  const int n_fields = (int)(sizeof fields / sizeof fields[0]);

  FILE* f = fopen(argv[2], "rb");
  if (!f) { perror(argv[2]); return 2; }
  char magic[8];
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "VKTB1\0\0\0", 8)) { fprintf(stderr, "bad tables file\n"); return 2; }
  for (;;) {
    char name[24];
    int32_t kind;
    int64_t count;
    if (fread(name, 1, 24, f) != 24 || fread(&kind, 4, 1, f) != 1 || fread(&count, 8, 1, f) != 1) break;
    if (!strcmp(name, "END")) break;
    const field_t* fd = NULL;
    for (int i = 0; i < n_fields; ++i)
      if (!strcmp(fields[i].name, name)) fd = &fields[i];
    if (!fd || fd->kind != kind) { fprintf(stderr, "unexpected record '%s'\n", name); return 2; }
    if (kind == 0) {
      int64_t v;
      if (fread(&v, 8, 1, f) != 1) return 2;
      *(int32_t*)fd->dst = (int32_t)v;
    } else if (kind == 1) {
      if (fread(fd->dst, 8, 1, f) != 1) return 2;
    } else {
      const size_t esz = kind == 2 ? 8 : 2;
      const size_t bytes = ((size_t)count * esz + 7) & ~(size_t)7;
      void* buf = xmalloc(bytes);
      if (bytes && fread(buf, 1, bytes, f) != bytes) return 2;
      *(void**)fd->dst = count ? buf : NULL;
    }
  }
  fclose(f);

  f = fopen(argv[3], "rb");
  if (!f) { perror(argv[3]); return 2; }
  fseek(f, 0, SEEK_END);
  const long nbytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  const int64_t n = nbytes / (long)(VK_NPAR * sizeof(double));
  double* params = (double*)xmalloc((size_t)nbytes);
  if (fread(params, 1, (size_t)nbytes, f) != (size_t)nbytes) return 2;
  fclose(f);

  char err[512] = "";
  vk_ctx* ctx = create(&t, 0, err, sizeof err);
  if (!ctx) { fprintf(stderr, "vk_create: %s\n", err); return 1; }
  vk_eval_opts o;
  defaults(&o);
  o.rescale_from_ap = atoi(argv[5]);
  o.assume_isotropic = atoi(argv[6]);
  o.like_form = atoi(argv[7]);
  o.nmocks = atof(argv[8]);
  o.nparams = atof(argv[9]);
  double* out = (double*)xmalloc((size_t)(2 * n) * sizeof(double));
  const int rc = eval(ctx, &o, params, n, out, out + n, NULL);
  if (rc) { fprintf(stderr, "vk_eval_batch: %d %s\n", rc, last_error(ctx)); destroy(ctx); return 1; }
  f = fopen(argv[4], "wb");
  if (!f) { perror(argv[4]); return 2; }
  fwrite(out, sizeof(double), (size_t)(2 * n), f);
  fclose(f);
  destroy(ctx);
  printf("%lld points evaluated through the C ABI (version %d)\n", (long long)n, abi());
  return 0;
}
