/* oracle/collapse_oracle.h -- TEST INFRASTRUCTURE ONLY (see collapse_oracle.c). */
#ifndef SBO_COLLAPSE_ORACLE_H
#define SBO_COLLAPSE_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* One cluster's read pairs -> its unique hits.  Pair p: left mate blocks [lo[p], lo[p+1]) of (ll, lr) (closed
 * coordinates, ascending), right mate [ro[p], ro[p+1]) of (rl, rr); either may be empty (a singleton); NH tag nh[p].
 * Out, per unique hit in order: uniq_pair (the input pair it was made of), uniq_mass (its collapse mass);
 * *cluster_mass (HitCluster::_weighted_mass), *n_filtered (pairs the span filter skipped).  Returns the number of
 * unique hits, or -1 on a malformed pair (no mate at all).                                                      */
int sbo_collapse_cluster(int n_pairs, const int64_t *lo, const uint32_t *ll, const uint32_t *lr, const int64_t *ro,
                         const uint32_t *rl, const uint32_t *rr, const int32_t *nh, int32_t *uniq_pair,
                         double *uniq_mass, double *cluster_mass, int32_t *n_filtered);
double sbo_phi(double x);
#ifdef __cplusplus
}
#endif
#endif
