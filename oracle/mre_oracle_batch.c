/* mre_oracle_batch.c -- TEST INFRASTRUCTURE (see mre_oracle.h).
 * OpenMP fan-out of the single-env oracle over a batch of independent envs;
 * used by bench.py's cpu_baseline leg and by the parity tests. */
#include "mre_oracle.h"
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* steps every env nstep times with its own ctrl row; returns threads used */
int mro_batch_step(const mro_model* m, mro_data** envs, int nenv, const double* ctrl /*[nenv][8]*/,
                   int nstep, int nthreads) {
  int used = 1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
  used = nthreads > 0 ? nthreads : omp_get_max_threads();
#else
  (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
  for (int e = 0; e < nenv; e++) {
    if (ctrl) {
      int n;
      double* c = mro_get(envs[e], "ctrl", &n);
      memcpy(c, ctrl + 8 * e, 8 * sizeof(double));
    }
    mro_step(m, envs[e], nstep);
  }
  return used;
}
