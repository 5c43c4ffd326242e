/* Plain-C client of include/mimo_hip.h (no Python, no C++): what a cgo / JNI / FFI binding would do.
 * Reads a tiny problem from stdin (N D K, then Z, c, b, W row-major), runs the fused E-step, the Gibbs label
 * step with Philox uniforms and the diagonal structure, and prints the results for the Python test to compare
 * with the oracle.   gcc -std=c99 -I include tests/abi_smoke.c -L mimo_amd -lmimo_hip -o abi_smoke */
#include <stdio.h>
#include <stdlib.h>
#include "mimo_hip.h"

#define CHECK(call)                                                                         \
  do {                                                                                      \
    int rc_ = (call);                                                                       \
    if (rc_ != MIMO_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mimo_last_error(ctx)); return 1; } \
  } while (0)

static double* read_doubles(size_t n) {
  double* p = (double*)malloc(sizeof(double) * (n ? n : 1));
  for (size_t i = 0; i < n; ++i)
    if (scanf("%lf", &p[i]) != 1) { fprintf(stderr, "short input\n"); exit(2); }
  return p;
}

int main(void) {
  long long N; int D, K;
  if (scanf("%lld %d %d", &N, &D, &K) != 3) return 2;
  double *Z = read_doubles((size_t)N * D), *c = read_doubles(K), *b = read_doubles((size_t)K * D),
         *W = read_doubles((size_t)K * D * D);
  const size_t slen = (size_t)K * (1 + D + (size_t)D * D);
  double* S = (double*)malloc(sizeof(double) * slen);
  double sc[3];
  int32_t* labels = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N ? N : 1));
  mimo_ctx* ctx = NULL;
  if (mimo_create(&ctx, 0) != MIMO_OK) { fprintf(stderr, "mimo_create: %s\n", mimo_last_error(NULL)); return 1; }
  printf("version %s\n", mimo_version());
  CHECK(mimo_upload(ctx, Z, N, D));
  CHECK(mimo_estep(ctx, c, b, W, K, 0, S, sc));
  printf("estep_scalar0 %.17g\n", sc[0]);
  for (size_t i = 0; i < slen; ++i) printf("S %.17g\n", S[i]);
  CHECK(mimo_gibbs_labels(ctx, c, b, W, K, 42u, 3u, NULL, 0, labels, S));
  for (long long n = 0; n < N; ++n) printf("L %d\n", (int)labels[n]);
  /* an error must come back as a code + message, never as a crash */
  if (mimo_estep(ctx, c, b, W, 0, 0, S, sc) == MIMO_OK) { fprintf(stderr, "K = 0 was accepted\n"); return 1; }
  printf("error_message %s\n", mimo_last_error(ctx));
  /* diagonal structure rejects a full W */
  CHECK(mimo_set_structure(ctx, MIMO_STRUCT_DIAG));
  if (D > 1 && mimo_estep(ctx, c, b, W, K, 0, S, sc) != MIMO_E_INVALID) { fprintf(stderr, "full W accepted as diagonal\n"); return 1; }
  CHECK(mimo_set_structure(ctx, MIMO_STRUCT_FULL));
  CHECK(mimo_destroy(ctx));
  printf("done\n");
  return 0;
}
