/* Plain-C client of include/sfem.h: no Python, no torch -- only the HIP
 * runtime for device memory.  Built and run by tests/test_c_abi.py.
 *
 * Checks, on one affine hex element of P^3 GLL nodes scaled to [0,h]^3:
 *   gather / scatter-add round trip with a -1 sentinel,
 *   stiffness of a constant = 0, mass of a constant sums to the volume,
 *   sfem_dot, error reporting through sfem_last_error(); the fused apply also
 *   through the compact connectivity (sfem_facet_table_build,
 *   sfem_helmholtz_setup_affine, facet table + chains, box and affine);
 *   one CG update with the mean projection folded in against host arithmetic.
 * Exit code 0 = all good; prints the failing check otherwise. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sfem.h"

#define CHECK(cond, ...)                       \
  do {                                         \
    if (!(cond)) {                             \
      fprintf(stderr, "FAIL: " __VA_ARGS__);   \
      fprintf(stderr, "\n");                   \
      return 1;                                \
    }                                          \
  } while (0)
#define HIP(call) CHECK((call) == hipSuccess, #call)

static void* to_device(const void* src, size_t bytes) {
  void* d = NULL;
  if (hipMalloc(&d, bytes) != hipSuccess) return NULL;
  if (src && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess)
    return NULL;
  return d;
}

int main(void) {
  CHECK(sfem_abi_version() == SFEM_ABI_VERSION, "ABI version");
  {
    /* limits of the cluster-assembly kernels: compiled for P = 4..8 */
    int size = 0, kmax = 0;
    CHECK(sfem_helmholtz_cluster_limits(8, SFEM_F64, &size, &kmax) == SFEM_OK &&
              size == 8 && kmax >= 1647,
          "cluster limits P=8 fp64 (a 2x2x2 cluster has 1647 table nodes)");
    CHECK(sfem_helmholtz_cluster_limits(4, SFEM_F32, &size, &kmax) == SFEM_OK &&
              size == 8 && kmax >= 8 * 56,
          "cluster limits P=4 fp32");
    CHECK(sfem_helmholtz_cluster_limits(9, SFEM_F64, &size, &kmax) ==
              SFEM_EUNSUPPORTED && strlen(sfem_last_error()) > 0,
          "cluster limits outside the compiled range");
    CHECK(SFEM_CG_NSCALARS == SFEM_CG_NSCALARS_NAMED + SFEM_CG_RR_SLOTS,
          "CG scalar block layout");
  }

  /* --- gather / scatter-add with the -1 sentinel (gather_scatter.py:121-133) */
  {
    const double u[4] = {1.0, 2.0, 3.0, 4.0};
    const int32_t idx[6] = {0, 3, -1, 3, 1, 0};
    double got[6], back[4];
    double* du = (double*)to_device(u, sizeof u);
    int32_t* di = (int32_t*)to_device(idx, sizeof idx);
    double* dl = (double*)to_device(NULL, sizeof got);
    double* db = (double*)to_device(NULL, sizeof back);
    CHECK(du && di && dl && db, "hipMalloc");
    CHECK(sfem_gather(du, di, dl, 6, -7.0, SFEM_F64, NULL) == SFEM_OK, "%s",
          sfem_last_error());
    HIP(hipMemcpy(got, dl, sizeof got, hipMemcpyDeviceToHost));
    const double want[6] = {1.0, 4.0, -7.0, 4.0, 2.0, 1.0};
    for (int i = 0; i < 6; ++i) CHECK(got[i] == want[i], "gather[%d]", i);
    CHECK(sfem_gather(du, di, dl, 6, 0.0, SFEM_F64, NULL) == SFEM_OK, "gather");
    CHECK(sfem_scatter_add(dl, di, db, 6, 4, 1, SFEM_F64, NULL) == SFEM_OK, "%s",
          sfem_last_error());
    HIP(hipMemcpy(back, db, sizeof back, hipMemcpyDeviceToHost));
    const double sums[4] = {2.0, 2.0, 0.0, 8.0};
    for (int i = 0; i < 4; ++i) CHECK(back[i] == sums[i], "scatter[%d]", i);
    double* dres = (double*)to_device(NULL, sizeof(double));
    double res = 0.0;
    CHECK(sfem_dot(du, du, 4, dres, SFEM_F64, NULL) == SFEM_OK, "dot");
    HIP(hipMemcpy(&res, dres, sizeof res, hipMemcpyDeviceToHost));
    CHECK(res == 30.0, "dot = %g", res);
    /* periodic QQ^T from the classes themselves: {0, 3} and {1, 2} */
    {
      const double v[4] = {1.0, 2.0, 3.0, 4.0};
      const int32_t members[4] = {0, 3, 1, 2}, offsets[3] = {0, 2, 4};
      double* dv = (double*)to_device(v, sizeof v);
      int32_t* dm = (int32_t*)to_device(members, sizeof members);
      int32_t* dof = (int32_t*)to_device(offsets, sizeof offsets);
      CHECK(sfem_exchange_classes(dv, dm, dof, 2, 1, 1, 1, SFEM_F64, NULL) ==
                SFEM_OK, "%s", sfem_last_error());
      double q[4];
      HIP(hipMemcpy(q, dv, sizeof q, hipMemcpyDeviceToHost));
      CHECK(q[0] == 5.0 && q[3] == 5.0 && q[1] == 5.0 && q[2] == 5.0,
            "exchange_classes");
      /* out = w - (b . w / total) 1, with w . out on the side */
      const double b[4] = {1.0, 1.0, 1.0, 1.0};
      double* dbb = (double*)to_device(b, sizeof b);
      double* dpart = (double*)to_device(NULL, SFEM_DOT_SLOTS * sizeof(double));
      double* dwz = (double*)to_device(NULL, sizeof(double));
      HIP(hipMemset(dwz, 0, sizeof(double)));
      HIP(hipMemcpy(dv, v, sizeof v, hipMemcpyHostToDevice));
      CHECK(sfem_subtract_weighted_mean(dv, dbb, 4.0, dv, dpart, 4, dwz,
                                        SFEM_F64, NULL) == SFEM_OK,
            "%s", sfem_last_error());
      double wz = 0.0;
      HIP(hipMemcpy(q, dv, sizeof q, hipMemcpyDeviceToHost));
      HIP(hipMemcpy(&wz, dwz, sizeof wz, hipMemcpyDeviceToHost));
      CHECK(q[0] == -1.5 && q[1] == -0.5 && q[2] == 0.5 && q[3] == 1.5,
            "subtract_weighted_mean");
      CHECK(wz == 5.0, "w . out = %g", wz);   /* sum v_i (v_i - 2.5) */
      CHECK(sfem_zero_strips(dv + 1, 1, 2, 2, SFEM_F64, NULL) == SFEM_OK,
            "%s", sfem_last_error());
      HIP(hipMemcpy(q, dv, sizeof q, hipMemcpyDeviceToHost));
      CHECK(q[0] == -1.5 && q[1] == 0.0 && q[2] == 0.5 && q[3] == 0.0,
            "zero_strips");
    }
    /* invalid arguments come back as a status + message, never a crash */
    CHECK(sfem_gather(du, di, dl, -1, 0.0, SFEM_F64, NULL) != SFEM_OK, "bad count");
    CHECK(strlen(sfem_last_error()) > 0, "error message");
  }

  /* --- fused Helmholtz apply on one affine element, P = 4 GLL nodes --------- */
  {
    enum { P = 4, N = P * P * P };
    const double h = 0.5;
    const double s5 = sqrt(0.2);
    const double x1[P] = {-1.0, -s5, s5, 1.0};              /* GLL nodes    */
    const double w1[P] = {1.0 / 6, 5.0 / 6, 5.0 / 6, 1.0 / 6};
    /* barycentric differentiation matrix on x1 (interpolation.py:230-244) */
    double bw[P], D[P * P];
    for (int i = 0; i < P; ++i) {
      bw[i] = 1.0;
      for (int j = 0; j < P; ++j)
        if (j != i) bw[i] /= (x1[i] - x1[j]);
    }
    for (int i = 0; i < P; ++i) {
      double row = 0.0;
      for (int j = 0; j < P; ++j)
        if (j != i) {
          D[i * P + j] = (bw[j] / bw[i]) / (x1[i] - x1[j]);
          row += D[i * P + j];
        }
      D[i * P + i] = -row;
    }
    double coords[N * 3], ones[N], lin[N], out[N];
    int32_t elems[N];
    for (int a = 0; a < P; ++a)
      for (int b = 0; b < P; ++b)
        for (int c = 0; c < P; ++c) {
          const int k = (a * P + b) * P + c;
          coords[3 * k + 0] = h * (x1[a] + 1) / 2;
          coords[3 * k + 1] = h * (x1[b] + 1) / 2;
          coords[3 * k + 2] = h * (x1[c] + 1) / 2;
          ones[k] = 1.0;
          lin[k] = coords[3 * k + 0];                       /* u = x         */
          elems[k] = k;
        }
    double* dco = (double*)to_device(coords, sizeof coords);
    double* dge = (double*)to_device(NULL, 24 * sizeof(double));
    int32_t* del = (int32_t*)to_device(elems, sizeof elems);
    int32_t* den = (int32_t*)to_device(NULL, sizeof elems);
    int32_t mult[N];
    for (int k = 0; k < N; ++k) mult[k] = 1;
    int32_t* dmu = (int32_t*)to_device(mult, sizeof mult);
    double* du1 = (double*)to_device(ones, sizeof ones);
    double* dux = (double*)to_device(lin, sizeof lin);
    double* dout = (double*)to_device(NULL, sizeof out);
    CHECK(dco && dge && del && den && dmu && du1 && dux && dout, "hipMalloc");
    CHECK(sfem_helmholtz_setup_multilinear(dco, dge, 1, 3, P, SFEM_F64, NULL) ==
              SFEM_OK, "%s", sfem_last_error());
    CHECK(sfem_encode_elements(del, NULL, dmu, NULL, den, N, NULL) == SFEM_OK,
          "%s", sfem_last_error());
    sfem_helmholtz_args a;
    memset(&a, 0, sizeof a);
    a.out = dout; a.enc = den; a.geo_elem = dge;
    a.dmat = D; a.weights = w1; a.nodes = x1;
    a.num_elements = 1; a.num_nodes = N; a.ndim = 3; a.P = P; a.ncomp = 1;
    a.dtype = SFEM_F64; a.geo_mode = SFEM_GEO_AFFINE;
    /* stiffness of a constant vanishes */
    a.u = du1; a.lambda0 = 0.0; a.lambda1 = 1.0;
    CHECK(sfem_helmholtz_apply(&a, NULL) == SFEM_OK, "%s", sfem_last_error());
    HIP(hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost));
    for (int k = 0; k < N; ++k) CHECK(fabs(out[k]) < 1e-13, "A 1 [%d] = %g", k, out[k]);
    /* mass of a constant sums to the volume h^3 */
    a.lambda0 = 1.0; a.lambda1 = 0.0;
    CHECK(sfem_helmholtz_apply(&a, NULL) == SFEM_OK, "%s", sfem_last_error());
    HIP(hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost));
    double vol = 0.0;
    for (int k = 0; k < N; ++k) vol += out[k];
    CHECK(fabs(vol - h * h * h) < 1e-14, "sum B 1 = %.17g", vol);
    /* u = x:  x^T A x = int |grad x|^2 = h^3 */
    a.u = dux; a.lambda0 = 0.0; a.lambda1 = 1.0;
    CHECK(sfem_helmholtz_apply(&a, NULL) == SFEM_OK, "%s", sfem_last_error());
    HIP(hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost));
    double energy = 0.0;
    for (int k = 0; k < N; ++k) energy += lin[k] * out[k];
    CHECK(fabs(energy - h * h * h) < 1e-14, "x^T A x = %.17g", energy);
  }
  /* --- the same through compact connectivity: facet table, box constants,
   *     a chain of one element; P = 6 GLL nodes ---------------------------- */
  {
    enum { P = 6, N = P * P * P };
    const double h = 0.25;
    const double x1[P] = {-1.0, -0.76505532392946469, -0.28523151648064510,
                          0.28523151648064510, 0.76505532392946469, 1.0};
    const double w1[P] = {1.0 / 15, 0.37847495629784698, 0.55485837703548635,
                          0.55485837703548635, 0.37847495629784698, 1.0 / 15};
    double bw[P], D[P * P];
    for (int i = 0; i < P; ++i) {
      bw[i] = 1.0;
      for (int j = 0; j < P; ++j)
        if (j != i) bw[i] /= (x1[i] - x1[j]);
    }
    for (int i = 0; i < P; ++i) {
      double row = 0.0;
      for (int j = 0; j < P; ++j)
        if (j != i) {
          D[i * P + j] = (bw[j] / bw[i]) / (x1[i] - x1[j]);
          row += D[i * P + j];
        }
      D[i * P + i] = -row;
    }
    static double coords[N * 3], ones[N], lin[N], out[N];
    static int32_t elems[N], mult[N];
    for (int a = 0; a < P; ++a)
      for (int b = 0; b < P; ++b)
        for (int c = 0; c < P; ++c) {
          const int k = (a * P + b) * P + c;
          coords[3 * k + 0] = h * (x1[a] + 1) / 2;
          coords[3 * k + 1] = 2 * h * (x1[b] + 1) / 2;      /* a box, not a cube */
          coords[3 * k + 2] = 3 * h * (x1[c] + 1) / 2;
          ones[k] = 1.0;
          lin[k] = coords[3 * k + 1];                       /* u = y         */
          elems[k] = k;
          mult[k] = 1;
        }
    const double vol = 6 * h * h * h;
    double* dco = (double*)to_device(coords, sizeof coords);
    double* dge = (double*)to_device(NULL, 24 * sizeof(double));
    double* dcs = (double*)to_device(NULL, 8 * sizeof(double));
    int32_t* del = (int32_t*)to_device(elems, sizeof elems);
    int32_t* dmu = (int32_t*)to_device(mult, sizeof mult);
    int32_t* dtab = (int32_t*)to_device(NULL, 27 * 4 * sizeof(int32_t));
    uint8_t* dok = (uint8_t*)to_device(NULL, 1);
    const int32_t offsets[2] = {0, 1}, chain[1] = {0};
    int32_t* doff = (int32_t*)to_device(offsets, sizeof offsets);
    int32_t* dch = (int32_t*)to_device(chain, sizeof chain);
    double* du1 = (double*)to_device(ones, sizeof ones);
    double* dux = (double*)to_device(lin, sizeof lin);
    double* dout = (double*)to_device(NULL, sizeof out);
    CHECK(dco && dge && dcs && del && dmu && dtab && dok && doff && dch && du1 &&
              dux && dout, "hipMalloc");
    CHECK(sfem_facet_table_build(del, NULL, dmu, dtab, dok, 1, N, P, NULL) ==
              SFEM_OK, "%s", sfem_last_error());
    uint8_t ok = 0;
    int32_t tab[27 * 4];
    HIP(hipMemcpy(&ok, dok, 1, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(tab, dtab, sizeof tab, hipMemcpyDeviceToHost));
    CHECK(ok == 1, "lexicographic ids are 27 affine facet maps");
    /* facet 13 = the element interior: first node (1,1,1), strides P^2, P, 1 */
    CHECK(tab[13 * 4] == (P + 1) * P + 1 && tab[13 * 4 + 1] == P * P &&
              tab[13 * 4 + 2] == P && tab[13 * 4 + 3] == 1, "interior facet");
    CHECK(sfem_helmholtz_setup_multilinear(dco, dge, 1, 3, P, SFEM_F64, NULL) ==
              SFEM_OK, "%s", sfem_last_error());
    CHECK(sfem_helmholtz_setup_affine(dge, dcs, 1, 1e-13, SFEM_F64, NULL) ==
              SFEM_OK, "%s", sfem_last_error());
    double cst[8];
    HIP(hipMemcpy(cst, dcs, sizeof cst, hipMemcpyDeviceToHost));
    CHECK(cst[7] == 1.0 && fabs(cst[6] - vol / 8) < 1e-15, "box, detJ = %g",
          cst[6]);
    sfem_helmholtz_args a;
    memset(&a, 0, sizeof a);
    a.out = dout; a.geo_elem = dge; a.geo_const = dcs; a.facet_table = dtab;
    a.chain_offsets = doff; a.chain_elems = dch; a.num_chains = 1;
    a.dmat = D; a.weights = w1; a.nodes = x1;
    a.num_elements = 1; a.num_nodes = N; a.ndim = 3; a.P = P; a.ncomp = 1;
    a.dtype = SFEM_F64;
    for (int mode = 0; mode < 2; ++mode) {      /* box, then general affine */
      a.geo_mode = mode == 0 ? SFEM_GEO_BOX : SFEM_GEO_AFFINE;
      a.u = du1; a.lambda0 = 0.0; a.lambda1 = 1.0;
      CHECK(sfem_helmholtz_apply(&a, NULL) == SFEM_OK, "%s", sfem_last_error());
      HIP(hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost));
      for (int k = 0; k < N; ++k)
        CHECK(fabs(out[k]) < 1e-13, "facet A 1 [%d] = %g", k, out[k]);
      a.lambda0 = 1.0; a.lambda1 = 0.0;
      CHECK(sfem_helmholtz_apply(&a, NULL) == SFEM_OK, "%s", sfem_last_error());
      HIP(hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost));
      double sum = 0.0;
      for (int k = 0; k < N; ++k) sum += out[k];
      CHECK(fabs(sum - vol) < 1e-13, "facet sum B 1 = %.17g", sum);
      a.u = dux; a.lambda0 = 0.0; a.lambda1 = 1.0;   /* y^T A y = volume */
      CHECK(sfem_helmholtz_apply(&a, NULL) == SFEM_OK, "%s", sfem_last_error());
      HIP(hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost));
      double energy = 0.0;
      for (int k = 0; k < N; ++k) energy += lin[k] * out[k];
      CHECK(fabs(energy - vol) < 1e-13, "facet y^T A y = %.17g", energy);
    }
    /* SFEM_GEO_BOX without a facet table is refused */
    a.facet_table = NULL; a.geo_mode = SFEM_GEO_BOX;
    CHECK(sfem_helmholtz_apply(&a, NULL) != SFEM_OK, "box without table");
  }
  /* --- CG updates with the mean projection folded in (navier_stokes.py:73-78,
   * linalg/cg.py:75-86): r -= a Ap; z = r - (w.r / total) 1; p = z + b p      */
  {
    enum { M = 1000 };
    static double x[M], r[M], p[M], ap[M], w[M], xr[M], rr[M], pr[M];
    double total = 0.0;
    for (int k = 0; k < M; ++k) {
      x[k] = sin(0.3 * k); r[k] = cos(0.7 * k) + 0.25; p[k] = sin(1.1 * k + 1.0);
      ap[k] = cos(0.2 * k * k); w[k] = 1.0 + 0.5 * sin(0.05 * k);
      total += w[k];
    }
    double scal[SFEM_CG_NSCALARS] = {0}, sums[SFEM_CG_MEAN_SUMS] = {0};
    const double gamma = 2.5, pap = 1.75, alpha = gamma / pap;
    scal[0] = gamma; scal[1] = pap;
    double *dx = (double*)to_device(x, sizeof x), *dr = (double*)to_device(r, sizeof r);
    double *dp = (double*)to_device(p, sizeof p), *dap = (double*)to_device(ap, sizeof ap);
    double *dw = (double*)to_device(w, sizeof w);
    double *dsc = (double*)to_device(scal, sizeof scal);
    double *dsm = (double*)to_device(sums, sizeof sums);
    CHECK(dx && dr && dp && dap && dw && dsc && dsm, "device buffers");
    CHECK(sfem_cg_scalars(dsc, 0, 1e9, 0.0, 0.0, NULL, NULL) == SFEM_OK, "%s",
          sfem_last_error());
    CHECK(sfem_cg_update_r_mean(dr, dap, dw, M, dsc, dsm, SFEM_F64, NULL) ==
              SFEM_OK, "%s", sfem_last_error());
    CHECK(sfem_cg_update_xp_mean(dx, dp, dr, M, dsc, dsm, total, SFEM_F64,
                                 NULL) == SFEM_OK, "%s", sfem_last_error());
    CHECK(sfem_cg_scalars(dsc, 1, 1e9, 0.0, 0.0, NULL, NULL) == SFEM_OK, "%s",
          sfem_last_error());
    double wr = 0.0, rz = 0.0;
    for (int k = 0; k < M; ++k) { rr[k] = r[k] - alpha * ap[k]; wr += w[k] * rr[k]; }
    const double c = wr / total;
    for (int k = 0; k < M; ++k) rz += rr[k] * (rr[k] - c);
    for (int k = 0; k < M; ++k) {
      xr[k] = x[k] + alpha * p[k];
      pr[k] = (rr[k] - c) + (rz / gamma) * p[k];
    }
    HIP(hipMemcpy(x, dx, sizeof x, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(r, dr, sizeof r, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(p, dp, sizeof p, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(scal, dsc, sizeof scal, hipMemcpyDeviceToHost));
    for (int k = 0; k < M; ++k)
      CHECK(fabs(x[k] - xr[k]) < 1e-13 && fabs(r[k] - rr[k]) < 1e-13 &&
                fabs(p[k] - pr[k]) < 1e-12,
            "mean-projection CG update at %d", k);
    CHECK(fabs(scal[0] - rz) < 1e-10 * fabs(rz) && scal[8] == 1.0,
          "closed iteration: gamma = r.z (%.17g vs %.17g)", scal[0], rz);
    CHECK(sfem_cg_update_xp_mean(dx, dp, dr, M, dsc, dsm, 0.0, SFEM_F64, NULL) !=
              SFEM_OK, "total = 0 is refused");
  }
  HIP(hipDeviceSynchronize());
  printf("c-abi OK\n");
  return 0;
}
