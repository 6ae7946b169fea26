/* abi_caller.c -- a plain C99 (and C++) program that drives one task chain of the MeshBlock update through
 * the C ABI of include/akmi.h, the way a maintainer's binding inside the reference's task functions would
 * (INTEGRATION.md section 2): Hydro::Fluxes -> RKUpdate -> ApplyPhysicalBCs -> ConToPrim -> NewTimeStep
 * (src/hydro/hydro_tasks.cpp:159-201, hydro_update.cpp:23-83, eos/ideal_hyd.cpp:29-115,
 * hydro_newdt.cpp:30-139) on one 16^3 MeshBlock with outflow boundaries, then the same stage through the
 * fused entry akmi_hydro_stage_fused.  No torch, no Python: device memory comes from the HIP runtime's C API.
 *
 *   gcc -std=c99  -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/abi_caller.c -L athenak_amd/lib -lakmi
 *       -L /opt/rocm/lib -lamdhip64 -lm -o abi_caller        (g++ -std=c++17 -x c++ compiles the same file)
 *
 * Checks (exit status 0 = all passed):
 *   1. a uniform state is a fixed point of the task chain, bit for bit, and dt3 = dx/(|v| + cs);
 *   2. a state with a density step moves mass across the step and conserves total mass to round-off
 *      (periodic wrap is not used: outflow faces see zero gradient for the two cycles taken);
 *   3. the fused stage entry returns the same bits as the task chain for the same input.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "akmi.h"

#define NX 16
#define NG 2
#define N (NX + 2*NG)
#define NCELL (N*N*N)

static int failures = 0;
#define CHECK(c, msg) do { if (!(c)) { printf("FAIL: %s\n", msg); ++failures; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s\n", (int)e_, #x); exit(2); } } while (0)
#define AK(x) do { if ((x) < 0) { printf("akmi error in %s: %s\n", #x, akmi_last_error()); exit(3); } } while (0)

static double *dev_alloc(size_t n) {
  void *p = NULL;
  HIP(hipMalloc(&p, n*sizeof(double)));
  HIP(hipMemset(p, 0, n*sizeof(double)));
  return (double *)p;
}

static size_t idx(int n, int k, int j, int i) { return (((size_t)n*N + k)*N + j)*N + i; }

/* one RK1 "stage" through the task entries; flux arrays are cell-shaped (face_shaped = 0) as Hydro::uflx is */
static void task_chain(const akmi_pack *pk, const int *bcs_d, double dt, double *w0, double *u0, double *u1,
                       double *f1, double *f2, double *f3, int *counters, double *dt3) {
  AK(akmi_copy_cons(pk, u0, u1, NULL));
  AK(akmi_hydro_fluxes(pk, AKMI_RECON_PLM, AKMI_RS_HLLC, w0, f1, f2, f3, 0, NULL));
  AK(akmi_rk_update(pk, 0.0, 1.0, dt, u0, u1, f1, f2, f3, 0, NULL));
  AK(akmi_hydro_bcs(pk, 5, bcs_d, u0, NULL));
  AK(akmi_hydro_c2p(pk, u0, w0, 0, N - 1, 0, N - 1, 0, N - 1, counters, NULL));
  AK(akmi_hydro_newdt(pk, w0, dt3, NULL));
}

int main(void) {
  const double gamma = 1.4, dxv = 1.0/NX;
  akmi_pack pk;
  double hdx[3], *dx_d, *w0, *u0, *u1, *f1, *f2, *f3, *dt3, *w0b, *u0b, *u1b;
  double *hu = (double *)malloc(5*NCELL*sizeof(double)), *hw = (double *)malloc(5*NCELL*sizeof(double));
  double *hu2 = (double *)malloc(5*NCELL*sizeof(double));
  int hb[6], *bcs_d, *counters, n, k, j, i, q;
  double hdt3[3], mass0, mass1, cs;
  long long wsb;
  void *ws = NULL;

  printf("akmi_version = %d\n", akmi_version());
  memset(&pk, 0, sizeof pk);
  hdx[0] = hdx[1] = hdx[2] = dxv;
  dx_d = dev_alloc(3);
  HIP(hipMemcpy(dx_d, hdx, sizeof hdx, hipMemcpyHostToDevice));
  pk.nmb = 1; pk.nvar = 5; pk.nx1 = pk.nx2 = pk.nx3 = NX; pk.ng = NG; pk.dx = dx_d;
  pk.gamma = gamma; pk.dfloor = pk.pfloor = pk.tfloor = pk.sfloor = 1.17549435e-38; pk.sigma_max = 3.4e38;
  pk.iso_cs = 0.0; pk.is_ideal = 1;
  for (q = 0; q < 6; ++q) hb[q] = AKMI_BC_OUTFLOW;
  HIP(hipMalloc((void **)&bcs_d, sizeof hb));
  HIP(hipMemcpy(bcs_d, hb, sizeof hb, hipMemcpyHostToDevice));
  HIP(hipMalloc((void **)&counters, 3*sizeof(int)));
  HIP(hipMemset(counters, 0, 3*sizeof(int)));
  w0 = dev_alloc(5*NCELL); u0 = dev_alloc(5*NCELL); u1 = dev_alloc(5*NCELL);
  w0b = dev_alloc(5*NCELL); u0b = dev_alloc(5*NCELL); u1b = dev_alloc(5*NCELL);
  f1 = dev_alloc(5*NCELL); f2 = dev_alloc(5*NCELL); f3 = dev_alloc(5*NCELL);
  dt3 = dev_alloc(3);

  /* ---- 1. uniform state: d = 1, v = (0.3, -0.2, 0.1), p = 0.6 ---------------------------------------- */
  for (n = 0; n < 5; ++n)
    for (k = 0; k < N; ++k) for (j = 0; j < N; ++j) for (i = 0; i < N; ++i) {
      const double d = 1.0, vx = 0.3, vy = -0.2, vz = 0.1, p = 0.6;
      const double e = p/(gamma - 1.0) + 0.5*d*(vx*vx + vy*vy + vz*vz);
      const double uv[5] = {d, d*vx, d*vy, d*vz, e};
      hu[idx(n, k, j, i)] = uv[n];
    }
  HIP(hipMemcpy(u0, hu, 5*NCELL*sizeof(double), hipMemcpyHostToDevice));
  AK(akmi_hydro_c2p(&pk, u0, w0, 0, N - 1, 0, N - 1, 0, N - 1, counters, NULL));
  task_chain(&pk, bcs_d, 0.01, w0, u0, u1, f1, f2, f3, counters, dt3);
  HIP(hipDeviceSynchronize());
  HIP(hipMemcpy(hu2, u0, 5*NCELL*sizeof(double), hipMemcpyDeviceToHost));
  HIP(hipMemcpy(hdt3, dt3, sizeof hdt3, hipMemcpyDeviceToHost));
  CHECK(memcmp(hu, hu2, 5*NCELL*sizeof(double)) == 0, "uniform state is not a fixed point of the task chain");
  cs = sqrt(gamma*0.6/1.0);
  CHECK(fabs(hdt3[0] - dxv/(0.3 + cs)) < 1e-15 && fabs(hdt3[1] - dxv/(0.2 + cs)) < 1e-15 &&
        fabs(hdt3[2] - dxv/(0.1 + cs)) < 1e-15, "dt3 != dx/(|v| + cs)");

  /* ---- 2. density step at the mid plane in x1, at rest; two steps ------------------------------------ */
  for (n = 0; n < 5; ++n)
    for (k = 0; k < N; ++k) for (j = 0; j < N; ++j) for (i = 0; i < N; ++i) {
      const double d = (i < N/2) ? 1.0 : 0.125, p = (i < N/2) ? 1.0 : 0.1;
      const double uv[5] = {d, 0.0, 0.0, 0.0, p/(gamma - 1.0)};
      hu[idx(n, k, j, i)] = uv[n];
    }
  HIP(hipMemcpy(u0, hu, 5*NCELL*sizeof(double), hipMemcpyHostToDevice));
  HIP(hipMemcpy(u0b, hu, 5*NCELL*sizeof(double), hipMemcpyHostToDevice));
  AK(akmi_hydro_c2p(&pk, u0, w0, 0, N - 1, 0, N - 1, 0, N - 1, counters, NULL));
  AK(akmi_hydro_c2p(&pk, u0b, w0b, 0, N - 1, 0, N - 1, 0, N - 1, counters, NULL));
  mass0 = 0.0;
  for (k = NG; k < NG + NX; ++k) for (j = NG; j < NG + NX; ++j) for (i = NG; i < NG + NX; ++i) mass0 += hu[idx(0, k, j, i)];
  task_chain(&pk, bcs_d, 0.005, w0, u0, u1, f1, f2, f3, counters, dt3);
  HIP(hipDeviceSynchronize());
  HIP(hipMemcpy(hu2, u0, 5*NCELL*sizeof(double), hipMemcpyDeviceToHost));
  HIP(hipMemcpy(hw, w0, 5*NCELL*sizeof(double), hipMemcpyDeviceToHost));
  mass1 = 0.0;
  for (k = NG; k < NG + NX; ++k) for (j = NG; j < NG + NX; ++j) for (i = NG; i < NG + NX; ++i) mass1 += hu2[idx(0, k, j, i)];
  CHECK(fabs(mass1 - mass0) < 1e-12*mass0, "mass not conserved by flux-form update");
  CHECK(hu2[idx(1, N/2, N/2, N/2)] > 0.0 && hu2[idx(0, N/2, N/2, N/2)] > 0.125, "no mass flux across the step");
  CHECK(hw[idx(1, N/2, N/2, N/2 - 1)] > 0.0, "ConsToPrim did not return a positive velocity at the step");

  /* ---- 3. the same stage through the fused entry ------------------------------------------------------ */
  wsb = akmi_stage_workspace_bytes(&pk, 0);
  if (wsb > 0) HIP(hipMalloc(&ws, (size_t)wsb));
  AK(akmi_hydro_stage_fused(&pk, AKMI_RECON_PLM, AKMI_RS_HLLC, 0.0, 1.0, 0.005, 1, w0b, u0b, u1b, 1, counters, dt3,
                            ws, NULL));
  AK(akmi_hydro_bcs(&pk, 5, bcs_d, u0b, NULL));
  AK(akmi_hydro_c2p_shell(&pk, u0b, w0b, counters, NULL));
  HIP(hipDeviceSynchronize());
  HIP(hipMemcpy(hu, u0b, 5*NCELL*sizeof(double), hipMemcpyDeviceToHost));
  CHECK(memcmp(hu, hu2, 5*NCELL*sizeof(double)) == 0, "fused stage and task chain differ");
  HIP(hipMemcpy(hu, w0b, 5*NCELL*sizeof(double), hipMemcpyDeviceToHost));
  CHECK(memcmp(hu, hw, 5*NCELL*sizeof(double)) == 0, "primitives of fused stage and task chain differ");

  printf(failures ? "abi_caller: %d check(s) FAILED\n" : "abi_caller: all checks passed\n", failures);
  return failures ? 1 : 0;
}
