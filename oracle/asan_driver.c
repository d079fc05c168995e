/* asan_driver.c -- runs every entry point of the CPU oracle on small seeded inputs; built with
 * -fsanitize=address,undefined by `make -C oracle asan` (sanitizers run on the CPU build only: the GPU pool
 * refuses them).  TEST INFRASTRUCTURE ONLY.  Exit code 0 and no sanitizer report = pass. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sfm_oracle.h"

static unsigned long long rng_s = 88172645463325252ull;
static unsigned rnd(void) {
  rng_s ^= rng_s << 13;
  rng_s ^= rng_s >> 7;
  rng_s ^= rng_s << 17;
  return (unsigned)(rng_s >> 11);
}
static double urand(void) { return (rnd() % 1000001) / 1000000.0; }

static int match_case(int nq, int nt, int dim, int dtype, int norm, int threads) {
  const size_t esz = dtype == ORC_DTYPE_F32 ? 4 : 1;
  unsigned char* q = (unsigned char*)malloc((size_t)(nq ? nq : 1) * dim * esz);
  unsigned char* t = (unsigned char*)malloc((size_t)(nt ? nt : 1) * dim * esz);
  for (int i = 0; i < nq * dim; ++i) {
    const unsigned v = rnd() % 4 * 60;  /* few levels: plenty of ties */
    if (dtype == ORC_DTYPE_F32) ((float*)q)[i] = (float)v; else q[i] = (unsigned char)v;
  }
  for (int i = 0; i < nt * dim; ++i) {
    const unsigned v = rnd() % 4 * 60;
    if (dtype == ORC_DTYPE_F32) ((float*)t)[i] = (float)v; else t[i] = (unsigned char)v;
  }
  int32_t* oq = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nq ? nq : 1));
  int32_t* ot = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nq ? nq : 1));
  float* od = (float*)malloc(sizeof(float) * (size_t)(nq ? nq : 1));
  int32_t* ki = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(nq ? nq : 1));
  float* kd = (float*)malloc(sizeof(float) * 2 * (size_t)(nq ? nq : 1));
  int32_t n = -1;
  int rc = orc_match_knn2(q, nq, t, nt, dim, dtype, norm, 0.8f, oq, ot, od, &n, ki, kd, threads);
  if (rc == 0 && (n < 0 || n > nq)) rc = 100;
  const void* imgs[2] = {q, t};
  const int32_t rows[2] = {nq, nt}, pairs[4] = {0, 1, 1, 0};
  int32_t counts[2];
  uint64_t cs[4];
  if (rc == 0) rc = orc_match_many_checksum(imgs, rows, dim, dtype, norm, pairs, 2, 0.8f, threads, counts, cs);
  if (rc == 0 && counts[0] != n) rc = 101;
  free(q); free(t); free(oq); free(ot); free(od); free(ki); free(kd);
  return rc;
}

int main(void) {
  int rc = 0;
  const int shapes[][2] = {{0, 5}, {5, 0}, {3, 1}, {4, 2}, {33, 65}, {70, 31}};
  for (unsigned s = 0; s < sizeof shapes / sizeof shapes[0] && !rc; ++s) {
    rc = match_case(shapes[s][0], shapes[s][1], 128, ORC_DTYPE_F32, ORC_NORM_L2, 1);
    if (!rc) rc = match_case(shapes[s][0], shapes[s][1], 61, ORC_DTYPE_U8, ORC_NORM_L2, 3);
    if (!rc) rc = match_case(shapes[s][0], shapes[s][1], 32, ORC_DTYPE_U8, ORC_NORM_HAMMING, 2);
  }
  if (rc) { fprintf(stderr, "match rc=%d\n", rc); return 1; }

  /* two views of a small scene: triangulation + reprojection filter */
  {
    const double K[9] = {1520, 0, 302.2, 0, 1520, 246.87, 0, 0, 1}, dist[5] = {0, 0, 0, 0, 0};
    const double P1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}, P2[12] = {1, 0, 0, -0.5, 0, 1, 0, 0.02, 0, 0, 1, 0.01};
    enum { M = 40 };
    double xy1[2 * M], xy2[2 * M], X[3 * M];
    float err[2 * M];
    uint8_t keep[M];
    for (int i = 0; i < M; ++i) {
      const double W[3] = {urand() - 0.5, urand() - 0.5, 4 + urand()};
      const double a[3] = {W[0], W[1], W[2]}, b[3] = {W[0] - 0.5, W[1] + 0.02, W[2] + 0.01};
      xy1[2 * i] = 1520 * a[0] / a[2] + 302.2 + (i % 7 == 0 ? 9.0 : 0.0);  /* some beyond the 6 px filter */
      xy1[2 * i + 1] = 1520 * a[1] / a[2] + 246.87;
      xy2[2 * i] = 1520 * b[0] / b[2] + 302.2;
      xy2[2 * i + 1] = 1520 * b[1] / b[2] + 246.87;
    }
    if (orc_triangulate(P1, P2, K, dist, xy1, xy2, M, 6.0f, X, err, keep)) { fprintf(stderr, "triangulate\n"); return 1; }
    if (orc_triangulate(P1, P2, K, dist, xy1, xy2, 0, 6.0f, X, err, keep)) { fprintf(stderr, "triangulate m=0\n"); return 1; }
  }

  /* a small bundle: 5 cameras on an arc, 60 points, 4 observations each */
  {
    enum { NC = 5, NP = 60, K_ = 4, NO = NP * K_ };
    double cams[6 * NC], pts[3 * NP], focal = 1500 * 1.01, xy[2 * NO];
    int32_t oc[NO], op[NO];
    for (int c = 0; c < NC; ++c) {
      cams[6 * c] = c == 0 ? 0 : 0.02 * c; cams[6 * c + 1] = c == 0 ? 0 : -0.01 * c; cams[6 * c + 2] = 0;
      cams[6 * c + 3] = 0.1 * c; cams[6 * c + 4] = 0; cams[6 * c + 5] = 6;
    }
    for (int p = 0; p < NP; ++p) for (int j = 0; j < 3; ++j) pts[3 * p + j] = urand() - 0.5;
    int o = 0;
    for (int p = 0; p < NP; ++p)
      for (int k = 0; k < K_; ++k, ++o) {
        const int c = (p + k) % NC;
        double r[2];
        const double zero[2] = {0, 0};
        oc[o] = c; op[o] = p;
        orc_ba_residual(cams + 6 * c, pts + 3 * p, 1500.0, zero, r, NULL, NULL, NULL);
        xy[2 * o] = r[0] + (urand() - 0.5); xy[2 * o + 1] = r[1] + (urand() - 0.5);
      }
    for (int p = 0; p < NP; ++p) pts[3 * p] += 0.01 * (urand() - 0.5);
    orc_ba_opts opt;
    orc_ba_default_opts(&opt);
    opt.max_time_s = 0;
    orc_ba_summary sm;
    double c2[6 * NC], p2[3 * NP], f2 = focal;
    memcpy(c2, cams, sizeof c2); memcpy(p2, pts, sizeof p2);
    if (orc_ba_solve(NC, NP, NO, c2, p2, &f2, oc, op, xy, &opt, &sm) || !(sm.final_cost <= sm.initial_cost)) { fprintf(stderr, "ba_solve\n"); return 1; }
    const int dim = 6 * NC + 1;
    double* S = (double*)malloc(sizeof(double) * dim * dim);
    double g[6 * NC + 1], scale[6 * NC + 3 * NP + 1], cost;
    if (orc_ba_reduced_system(NC, NP, NO, cams, pts, focal, oc, op, xy, 1e4, NULL, scale, S, g, &cost)) { fprintf(stderr, "reduced\n"); return 1; }
    free(S);
    if (!(orc_ba_time_iterations(NC, NP, NO, cams, pts, focal, oc, op, xy, 2, NULL) >= 0)) { fprintf(stderr, "time_iterations\n"); return 1; }
    double R[9], aa[3] = {0.3, -0.2, 0.5}, aa2[3], out[3];
    orc_angleaxis_to_rotmat_colmajor(aa, R);
    orc_rotmat_colmajor_to_angleaxis(R, aa2);
    orc_angleaxis_rotate_point(aa, pts, out);
    if (fabs(aa[0] - aa2[0]) > 1e-12) { fprintf(stderr, "angle-axis round trip\n"); return 1; }
  }

  /* incremental glue */
  {
    const int32_t trk_ptr[4] = {0, 2, 4, 6}, trk_view[6] = {0, 1, 0, 2, 1, 2}, trk_feat[6] = {5, 7, 6, 1, 8, 2};
    const int32_t mq[3] = {5, 6, 9}, mt[3] = {11, 12, 13};
    int32_t oc[3], of[3], n = -1;
    if (orc_find_2d3d(trk_ptr, trk_view, trk_feat, 3, 0, 1, mq, mt, 3, oc, of, &n) || n < 0 || n > 3) { fprintf(stderr, "find_2d3d\n"); return 1; }
    const double cloud[6] = {0, 0, 0, 1, 1, 1}, fresh[9] = {0, 0, 0.004, 5, 5, 5, 5, 5, 5.001};
    uint8_t acc[3];
    int32_t na = -1;
    if (orc_merge_new_points(cloud, 2, fresh, 3, 0.01f, acc, &na) || na != 1) { fprintf(stderr, "merge_new_points %d\n", na); return 1; }
  }
  printf("oracle asan driver ok\n");
  return 0;
}
