/*
 * src/hip_cdna4.c of the lsbench tree -- the MI355X (gfx950) backend's place in
 * the reference's one-file-per-backend layout.
 *
 * ENABLE_HIP=ON  (-DLSBENCH_HIP): this file is empty; hip_cdna4_init /
 *   _finalize / _bench come from liblsbench_hip.so (libs/hip.cmake links it).
 * ENABLE_HIP=OFF: the reference's convention for a backend that is compiled out
 *   (src/cholmod.c:74-81, src/cusparse.c:218-225): stubs returning 1, so that
 *   lsbench_init / lsbench_bench / lsbench_finalize link and `--solver hip` is
 *   the same silent no-op as any other disabled solver.
 */
#include "lsbench-impl.h"

#if !defined(LSBENCH_HIP)
int hip_cdna4_init() { return 1; }
int hip_cdna4_finalize() { return 1; }
int hip_cdna4_bench(double *x, struct csr *A, const double *r,
                    const struct lsbench *cb) {
  return 1;
}
int hip_cdna4_set_option(const char *name, const char *value) { return 1; }
struct csr *hip_cdna4_matrix_synth(const char *spec) { return 0; }
#endif
