// akmi_stage.hip -- per-stage fast path of a MeshBlockPack (C ABI: akmi_*_stage_update,
// akmi_*_c2p_newdt).  Results are bit-identical to the task chain of akmi_tasks.hip.
#include "akmi_common.hpp"

using namespace akmi;

namespace akmi {

struct StageWs {
  double *flx1, *flx2, *flx3;
  double *efc[6];
  double *e1, *e2, *e3;
  size_t total;
};

static StageWs carve(const Geo &g, int is_mhd, void *ws) {
  StageWs w;
  double *p = (double *)ws;
  size_t nmb = g.nmb, nv = g.nvar;
  size_t n1 = nmb*nv*g.N3*g.N2*(g.N1 + 1), n2 = nmb*nv*g.N3*(g.N2 + 1)*g.N1,
         n3 = nmb*nv*(g.N3 + 1)*g.N2*g.N1;
  size_t off = 0;
  auto take = [&](size_t n) { double *r = p ? p + off : nullptr; off += (n + 31) & ~(size_t)31; return r; };
  w.flx1 = take(n1); w.flx2 = take(n2); w.flx3 = take(n3);
  for (int q = 0; q < 6; ++q) w.efc[q] = nullptr;
  w.e1 = w.e2 = w.e3 = nullptr;
  if (is_mhd) {
    size_t nc = nmb*g.N3*g.N2*g.N1;
    for (int q = 0; q < 6; ++q) w.efc[q] = take(nc);
    w.e1 = take(nmb*(g.N3 + 1)*(g.N2 + 1)*g.N1);
    w.e2 = take(nmb*(g.N3 + 1)*g.N2*(g.N1 + 1));
    w.e3 = take(nmb*g.N3*(g.N2 + 1)*(g.N1 + 1));
  }
  w.total = off*sizeof(double);
  return w;
}

}  // namespace akmi

extern "C" {

long long akmi_stage_workspace_bytes(const akmi_pack *p, int is_mhd) {
  Geo g = make_geo(p);
  return (long long)carve(g, is_mhd, nullptr).total;
}

int akmi_hydro_stage_update(const akmi_pack *p, int recon, int rsolver, double gam0, double gam1,
                            double beta_dt, int copy_u1, const double *w0, double *u0, double *u1,
                            void *ws, void *stream) {
  Geo g = make_geo(p);
  StageWs w = carve(g, 0, ws);
  int rc = AKMI_COMPLETE;
  if (copy_u1) rc = akmi_copy_cons(p, u0, u1, stream);
  if (rc == AKMI_COMPLETE) rc = akmi_hydro_fluxes(p, recon, rsolver, w0, w.flx1, w.flx2, w.flx3, 1, stream);
  if (rc == AKMI_COMPLETE) rc = akmi_rk_update(p, gam0, gam1, beta_dt, u0, u1, w.flx1, w.flx2, w.flx3, 1, stream);
  return rc;
}

int akmi_mhd_stage_update(const akmi_pack *p, int recon, int rsolver, double gam0, double gam1,
                          double beta_dt, int copy_u1, const double *w0, const double *bcc0,
                          double *u0, double *u1, double *b0x1f, double *b0x2f, double *b0x3f,
                          double *b1x1f, double *b1x2f, double *b1x3f, void *ws, void *stream) {
  Geo g = make_geo(p);
  StageWs w = carve(g, 1, ws);
  hipStream_t st = (hipStream_t)stream;
  int rc = AKMI_COMPLETE;
  if (copy_u1) {
    rc = akmi_copy_cons(p, u0, u1, stream);
    size_t nmb = g.nmb;
    hipMemcpyAsync(b1x1f, b0x1f, sizeof(double)*nmb*g.N3*g.N2*(g.N1 + 1), hipMemcpyDeviceToDevice, st);
    hipMemcpyAsync(b1x2f, b0x2f, sizeof(double)*nmb*g.N3*(g.N2 + 1)*g.N1, hipMemcpyDeviceToDevice, st);
    hipMemcpyAsync(b1x3f, b0x3f, sizeof(double)*nmb*(g.N3 + 1)*g.N2*g.N1, hipMemcpyDeviceToDevice, st);
  }
  if (rc == AKMI_COMPLETE)
    rc = akmi_mhd_fluxes(p, recon, rsolver, w0, bcc0, b0x1f, b0x2f, b0x3f, w.flx1, w.flx2, w.flx3,
                         w.efc[0], w.efc[1], w.efc[2], w.efc[3], w.efc[4], w.efc[5], stream);
  if (rc == AKMI_COMPLETE)
    rc = akmi_rk_update(p, gam0, gam1, beta_dt, u0, u1, w.flx1, w.flx2, w.flx3, 1, stream);
  if (rc == AKMI_COMPLETE)
    rc = akmi_mhd_corner_e(p, w0, bcc0, w.efc[0], w.efc[1], w.efc[2], w.efc[3], w.efc[4], w.efc[5],
                           w.flx1, w.flx2, w.flx3, w.e1, w.e2, w.e3, stream);
  if (rc == AKMI_COMPLETE)
    rc = akmi_mhd_ct(p, gam0, gam1, beta_dt, w.e1, w.e2, w.e3, b0x1f, b0x2f, b0x3f, b1x1f, b1x2f,
                     b1x3f, stream);
  return rc;
}

int akmi_hydro_c2p_newdt(const akmi_pack *p, double *u0, double *w0, int do_newdt, int *counters,
                         double *dt3, void *stream) {
  Geo g = make_geo(p);
  int rc = akmi_hydro_c2p(p, u0, w0, 0, g.N1 - 1, 0, g.N2 - 1, 0, g.N3 - 1, counters, stream);
  if (rc == AKMI_COMPLETE && do_newdt) rc = akmi_hydro_newdt(p, w0, dt3, stream);
  return rc;
}

int akmi_mhd_c2p_newdt(const akmi_pack *p, double *u0, const double *bx1f, const double *bx2f,
                       const double *bx3f, double *w0, double *bcc0, int do_newdt, int *counters,
                       double *dt3, void *stream) {
  Geo g = make_geo(p);
  int rc = akmi_mhd_c2p(p, u0, bx1f, bx2f, bx3f, w0, bcc0, 0, g.N1 - 1, 0, g.N2 - 1, 0, g.N3 - 1,
                        counters, stream);
  if (rc == AKMI_COMPLETE && do_newdt) rc = akmi_mhd_newdt(p, w0, bcc0, dt3, stream);
  return rc;
}

}  // extern "C"
