/* Command-line driver: the same five calls as the reference's bin/driver.c:5-15,
 * linked against liblsbench_hip.so.
 *   driver --solver hip --matrix tests/golden/matrices/I1_05x05.txt --trials=10 */
#include "lsbench.h"

int main(int argc, char **argv) {
  struct lsbench *cb = lsbench_init(argc, argv);
  struct csr *A = lsbench_matrix_read(lsbench_get_matrix_name(cb));
  lsbench_bench(A, cb);
  lsbench_matrix_free(A);
  lsbench_finalize(cb);
  return 0;
}
