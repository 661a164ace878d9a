/* oracle/ora_cmaes.h -- TEST INFRASTRUCTURE ONLY (see ora.h). CMA-ES restatement (cmaes_ref.c). */
#ifndef ORA_CMAES_H
#define ORA_CMAES_H
#ifdef __cplusplus
extern "C" {
#endif
typedef struct ora_cmaes ora_cmaes;
ora_cmaes *ora_cmaes_create(int N, const double *xstart, const double *stds, long seed, int lambda);
void   ora_cmaes_free(ora_cmaes *t);
const double *ora_cmaes_sample(ora_cmaes *t);          /* lambda x N row-major */
void   ora_cmaes_resample(ora_cmaes *t, int k);
void   ora_cmaes_update(ora_cmaes *t, const double *fit);
void   ora_sorted_index(const double *f, int *iindex, int n);
void   ora_cmaes_eigen(int N, const double *C, double *diag, double *Q, double *tmp /* N+1 */);
double ora_cmaes_gauss(ora_cmaes *t);
int    ora_cmaes_get_int(const ora_cmaes *t, const char *key);
double ora_cmaes_get(const ora_cmaes *t, const char *key);
const double *ora_cmaes_ptr(const ora_cmaes *t, const char *key);
const int *ora_cmaes_index(const ora_cmaes *t);
/* test hook (teacher forcing: resync the oracle to the device's fp32-rounded state between steps) */
void   ora_cmaes_set_sigma(ora_cmaes *t, double sigma);
int    ora_cmaes_test_for_termination(const ora_cmaes *t, char *out /* >= 128 bytes */);
#ifdef __cplusplus
}
#endif
#endif
