/*
 * lsbench public API, as seen by a driver program.
 *
 * This header keeps the reference's public surface ABI-identical
 * (reference: src/lsbench.h:8-40 -- three enums, two opaque structs, seven
 * functions) and adds exactly one enumerator, LSBENCH_SOLVER_HIP = 6, the
 * next free value after LSBENCH_SOLVER_GINKGO = 5 (reference:
 * src/lsbench.h:15).  Everything else that is new lives in lsbench_hip.h and
 * is additive.
 */
#ifndef LSBENCH_PUBLIC_H
#define LSBENCH_PUBLIC_H

#ifdef __cplusplus
extern "C" {
#endif

/* Solver selector.  Values 0..5 are the reference's (src/lsbench.h:8-16). */
typedef enum {
  LSBENCH_SOLVER_NONE = -1,
  LSBENCH_SOLVER_CUSOLVER = 0,
  LSBENCH_SOLVER_HYPRE = 1,
  LSBENCH_SOLVER_AMGX = 2,
  LSBENCH_SOLVER_CHOLMOD = 3,
  LSBENCH_SOLVER_PARALMOND = 4,
  LSBENCH_SOLVER_GINKGO = 5,
  LSBENCH_SOLVER_HIP = 6 /* new: MI355X-native PCG/Jacobi backend */
} lsbench_solver_t;

/* reference: src/lsbench.h:18-22 */
typedef enum {
  LSBENCH_PRECISION_FP64 = 0,
  LSBENCH_PRECISION_FP32 = 1,
  LSBENCH_PRECISION_FP16 = 2
} lsbench_precision_t;

/* reference: src/lsbench.h:24-29 */
typedef enum {
  LSBENCH_ORDERING_NONE = -1,
  LSBENCH_ORDERING_RCM = 0,
  LSBENCH_ORDERING_AMD = 1,
  LSBENCH_ORDERING_METIS = 2
} lsbench_ordering_t;

/* Matrix container (opaque here; layout in lsbench_hip.h). */
struct csr;
struct csr *lsbench_matrix_read(const char *fname);  /* src/lsbench.h:32 */
void lsbench_matrix_print(const struct csr *A);      /* src/lsbench.h:33 */
void lsbench_matrix_free(struct csr *A);             /* src/lsbench.h:34 */

/* Run configuration (opaque here; layout in lsbench_hip.h). */
struct lsbench;
struct lsbench *lsbench_init(int argc, char *argv[]);     /* src/lsbench.h:37 */
const char *lsbench_get_matrix_name(struct lsbench *cb);  /* src/lsbench.h:38 */
void lsbench_bench(struct csr *A, const struct lsbench *cb); /* :39 */
void lsbench_finalize(struct lsbench *cb);                /* src/lsbench.h:40 */

#ifdef __cplusplus
}
#endif

#endif /* LSBENCH_PUBLIC_H */
