/* akmi.h -- C ABI of the MI355X-native MeshBlock finite-volume update (AthenaK hot path).
 *
 * AthenaK has no FFI layer; the seam this library replaces is the *task member function*
 *     TaskStatus (Hydro|MHD)::Name(Driver *pdriver, int stage)
 * (reference: src/tasklist/task_list.hpp:178-185, src/hydro/hydro.hpp:124-154,
 * src/mhd/mhd.hpp:155-199).  Each entry point below is the body of one such task, taking
 * what the task reads from pmy_pack as plain pointers and scalars.  See INTEGRATION.md for
 * the one-line binding a maintainer adds inside each reference task.
 *
 * Conventions
 *  - All field pointers are DEVICE pointers (HBM) to IEEE fp64 in the reference's
 *    LayoutRight order (src/athena.hpp:111,127-128):
 *        cell array   (nmb, nvar, N3, N2, N1)            i fastest, no padding
 *        x1-face      (nmb, [nvar,] N3,   N2,   N1+1)    (src/athena.hpp:178-196)
 *        x2-face      (nmb, [nvar,] N3,   N2+1, N1  )
 *        x3-face      (nmb, [nvar,] N3+1, N2,   N1  )
 *        x1-edge      (nmb, N3+1, N2+1, N1  )            (src/athena.hpp:223-231)
 *        x2-edge      (nmb, N3+1, N2,   N1+1)
 *        x3-edge      (nmb, N3,   N2+1, N1+1)
 *    with N1 = nx1+2*ng, N2 = nx2>1 ? nx2+2*ng : 1, N3 likewise
 *    (src/hydro/hydro.cpp:283-288, src/mhd/mhd.cpp:148-160).
 *  - The library never allocates, frees or reallocates a caller's array.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls enqueue
 *    work and return; only functions documented as synchronous block.
 *  - Return value mirrors TaskStatus (src/tasklist/task_list.hpp:30):
 *        AKMI_COMPLETE 0, AKMI_INCOMPLETE 1, AKMI_FAIL <0 (message via akmi_last_error()).
 */
#ifndef AKMI_H_
#define AKMI_H_

#ifdef __cplusplus
extern "C" {
#endif

#define AKMI_COMPLETE    0
#define AKMI_INCOMPLETE  1
#define AKMI_FAIL       -1

/* variable indices, src/athena.hpp:65-67 */
enum { AKMI_IDN = 0, AKMI_IM1 = 1, AKMI_IM2 = 2, AKMI_IM3 = 3, AKMI_IEN = 4 };
enum { AKMI_IBX = 0, AKMI_IBY = 1, AKMI_IBZ = 2 };

/* ReconstructionMethod, src/athena.hpp (enum class) ; only dc/plm/ppm4 are on the path */
enum { AKMI_RECON_DC = 0, AKMI_RECON_PLM = 1, AKMI_RECON_PPM4 = 2, AKMI_RECON_PPMX = 3,
       AKMI_RECON_WENOZ = 4, AKMI_RECON_TENO = 5 };
/* Hydro_RSolver / MHD_RSolver */
enum { AKMI_RS_LLF = 0, AKMI_RS_HLLE = 1, AKMI_RS_HLLC = 2, AKMI_RS_HLLD = 3, AKMI_RS_ROE = 4,
       AKMI_RS_ADVECT = 5 /* kinematic runs (<time>/evolution = kinematic), task-granular entries only */ };
/* BoundaryFlag, src/mesh/mesh.hpp */
enum { AKMI_BC_BLOCK = -1, AKMI_BC_PERIODIC = 0, AKMI_BC_OUTFLOW = 1, AKMI_BC_REFLECT = 2,
       AKMI_BC_USER = 3 /* left to the caller's user function */, AKMI_BC_INFLOW = 4, AKMI_BC_DIODE = 5,
       AKMI_BC_VACUUM = 6 };

/* MeshBlockPack descriptor: RegionIndcs (src/mesh/mesh.hpp:35-41), mb_size.dx1..3
 * (src/mesh/mesh.hpp:25-29) and EOS_Data (src/eos/eos.hpp:27-34) by value. */
typedef struct akmi_pack {
  int nmb;               /* nmb_thispack: MeshBlocks looped over                        */
  int nvar;              /* nhydro|nmhd (+ nscalars): 5 ideal gas, 4 isothermal         */
  int nx1, nx2, nx3;     /* active cells per MeshBlock                                  */
  int ng;                /* ghost cells                                                 */
  const double *dx;      /* [nmb][3] dx1,dx2,dx3 per block, same memory space as fields */
  double gamma;          /* EOS_Data::gamma (ideal gas)                                 */
  double dfloor, pfloor, tfloor, sfloor;   /* default FLT_MIN, src/eos/eos.cpp:22-25    */
  double sigma_max;      /* default FLT_MAX, src/eos/ideal_mhd.cpp:22                   */
  double iso_cs;         /* EOS_Data::iso_cs (isothermal sound speed)                   */
  int is_ideal;          /* EOS_Data::is_ideal: 1 ideal gas (nvar 5), 0 isothermal (nvar 4:
                          * no energy variable, src/eos/isothermal_hyd.cpp:20-23)        */
} akmi_pack;

/* MHD packs of up to this many cells count as "small": both hosts run the task-granular chain instead of the fused stage
 * for such 3-D packs (unless the deck sets mhd/fused_stage itself or small_pack_tasks = false) -- measured crossover 72^3
 * (64^3: chain 1 113 against 1 042 Mcell-updates/s, 80^3: 1 392 against 1 494; PPM4 the same).  Hydro packs always take the
 * fused stage: its one-kernel form wins at every size (64^3: 2 838 against 1 854, 32^3: 485 against 303).  INTEGRATION.md. */
#define AKMI_SMALL_PACK_CELLS 375000

const char *akmi_last_error(void);
int akmi_version(void);
/* "production", or "experiments: ..." for a library built with -DAKMI_EXPERIMENTS (timing experiments that change
 * results; __graft_entry__.build() never defines it and tests/test_capi_symbols.py refuses such a library) */
const char *akmi_build_flags(void);

/* ---- Hydro tasks ------------------------------------------------------------------ */
/* Hydro::CopyCons (src/hydro/hydro_tasks.cpp:130-152): u1 <- u0 */
int akmi_copy_cons(const akmi_pack *p, const double *u0, double *u1, void *stream);
/* Hydro::Fluxes with <hydro>/fofc = true (src/hydro/hydro_fluxes.cpp:92-101): the same fluxes over
 * ranges extended by one face / one transverse cell, as Hydro::FOFC needs them */
int akmi_hydro_fluxes_fofc(const akmi_pack *p, int recon, int rsolver, const double *w0,
                           double *flx1, double *flx2, double *flx3, int face_shaped, void *stream);
/* Hydro::FOFC (src/hydro/hydro_fofc.cpp:30-371, Newtonian): trial update of the cells
 * [is-1,ie+1] x [js-1,je+1] x [ks-1,ke+1], flag those whose conversion to primitives would need a
 * floor (ConsToPrim(..., only_testfloors=true), src/eos/ideal_hyd.cpp:67-72), replace the fluxes on
 * the faces of flagged cells by first-order LLF fluxes (SingleStateLLF_Hyd), reset the flags.
 * fofc = Hydro::fofc as unsigned char [nmb][N3][N2][N1], zero on entry and on exit (caller-owned);
 * nfofc = device int, incremented by the number of flagged cells (EventCounters::nfofc). */
int akmi_hydro_fofc(const akmi_pack *p, double gam0, double gam1, double beta_dt, const double *w0,
                    const double *u0, const double *u1, double *flx1, double *flx2, double *flx3,
                    int face_shaped, unsigned char *fofc, int *nfofc, void *stream);
/* MHD::Fluxes with <mhd>/fofc = true (src/mhd/mhd_fluxes.cpp:100-105): face-normal ranges extended
 * by one face on both sides (ideal gas, no scalars) */
int akmi_mhd_fluxes_fofc(const akmi_pack *p, int recon, int rsolver, const double *w0,
                         const double *bcc0, const double *bx1f, const double *bx2f,
                         const double *bx3f, double *flx1, double *flx2, double *flx3, double *e3x1,
                         double *e2x1, double *e1x2, double *e3x2, double *e2x3, double *e1x3,
                         void *stream);
/* MHD::FOFC (src/mhd/mhd_fofc.cpp:30-493, Newtonian ideal gas): as akmi_hydro_fofc, with the trial
 * cell-centred field bcctest = gam0*bcc0 + gam1*avg(b1) -/+ dt/dx * d(face EMFs) (:88-107) in the
 * floor test, and the face EMFs of flagged cells replaced together with the fluxes. */
int akmi_mhd_fofc(const akmi_pack *p, double gam0, double gam1, double beta_dt, const double *w0,
                  const double *bcc0, const double *b0x1f, const double *b0x2f, const double *b0x3f,
                  const double *b1x1f, const double *b1x2f, const double *b1x3f, const double *u0,
                  const double *u1, double *flx1, double *flx2, double *flx3, double *e3x1,
                  double *e2x1, double *e1x2, double *e3x2, double *e2x3, double *e1x3,
                  unsigned char *fofc, int *nfofc, void *stream);
/* Hydro::NewTimeStep / MHD::NewTimeStep with <time>/evolution = kinematic (src/hydro/hydro_newdt.cpp:55-72,
 * src/mhd/mhd_newdt.cpp:56-73): dt3[d] = min over the active cells of dx_d/|v_d| */
int akmi_kinematic_newdt(const akmi_pack *p, const double *w0, double *dt3, void *stream);

/* ---- diffusion hooks of the Fluxes / EField tasks (src/hydro/hydro_tasks.cpp:183-189,
 * src/mhd/mhd_tasks.cpp:198-206,381-383); constant coefficients ---------------------------- */
/* Viscosity::AddViscousFluxIso (src/diffusion/viscosity.cpp:64-229): momentum and energy fluxes of
 * the isotropic Navier-Stokes stress subtracted from flx on the faces of the active cells */
int akmi_viscous_fluxes(const akmi_pack *p, double nu_iso, const double *w0, double *flx1,
                        double *flx2, double *flx3, int face_shaped, void *stream);
/* Conduction::AddHeatFluxIso (src/diffusion/conduction.cpp:106-152): q = -alpha*d*grad((gamma-1)e/d) */
int akmi_heat_fluxes(const akmi_pack *p, double alpha_iso, const double *w0, double *flx1,
                     double *flx2, double *flx3, int face_shaped, void *stream);
/* Conduction::NewTimeStep (src/diffusion/conduction.cpp:314-377): *dtmin (device) = min over the
 * active cells of SQR(dx)/alpha*d/(gamma-1); the caller multiplies by fac = 1/2, 1/4, 1/6 */
int akmi_conduction_newdt(const akmi_pack *p, double alpha_iso, const double *w0, double *dtmin,
                          void *stream);
/* Resistivity::AddEMFConstantResist (src/diffusion/resistivity.cpp:78-177): efld += eta_ohm*J, J from
 * CurrentDensity (src/diffusion/current_density.hpp:30-57) */
int akmi_resistive_emfs(const akmi_pack *p, double eta_ohm, const double *bx1f, const double *bx2f,
                        const double *bx3f, double *e1, double *e2, double *e3, void *stream);
/* Resistivity::AddFluxConstantResist (src/diffusion/resistivity.cpp:185-272): Poynting flux of the
 * resistive field added to the energy component of the face-shaped MHD fluxes */
int akmi_resistive_fluxes(const akmi_pack *p, double eta_ohm, const double *bx1f, const double *bx2f,
                          const double *bx3f, double *flx1, double *flx2, double *flx3, void *stream);
/* Resistivity::AddEMFConstantAmbipolar (src/diffusion/ambipolar.cpp:66-246): efld += eta_ad*(B^2 J -
 * (J.B) B) with J (EdgeJ1/2/3) and B (faces, bcc0) averaged to each edge. */
int akmi_ambipolar_emfs(const akmi_pack *p, double eta_ad, const double *bcc0, const double *bx1f,
                        const double *bx2f, const double *bx3f, double *e1, double *e2, double *e3,
                        void *stream);
/* Resistivity::AddFluxConstantAmbipolar (src/diffusion/ambipolar.cpp:254-494): Poynting flux of the
 * ambipolar field added to the energy component of the face-shaped MHD fluxes (ideal gas) */
int akmi_ambipolar_fluxes(const akmi_pack *p, double eta_ad, const double *bcc0, const double *bx1f,
                          const double *bx2f, const double *bx3f, double *flx1, double *flx2,
                          double *flx3, void *stream);
/* Resistivity::NewTimeStep with eta_ad != 0 (src/diffusion/resistivity.cpp:313-345): *dtmin (device) =
 * min over the active cells of SQR(dx)/(eta_ohm + eta_ad*B^2); the caller multiplies by fac */
int akmi_resistive_newdt(const akmi_pack *p, double eta_ohm, double eta_ad, const double *bcc0,
                         double *dtmin, void *stream);

/* Hydro::CopyCons, stages 2..4 of integrator rk4 (src/hydro/hydro_tasks.cpp:134-148): the second
 * register of the 2S scheme, u1 += delta*u0 on the active cells */
int akmi_rk4_copy_cons(const akmi_pack *p, double delta, const double *u0, double *u1, void *stream);

/* Hydro::Fluxes -> CalculateFluxes<hllc> (src/hydro/hydro_fluxes.cpp:77-229): reconstruct
 * w0 (recon), solve Riemann problem at faces i in [is,ie+1] (x1), j in [js,je+1] (x2),
 * k in [ks,ke+1] (x3).  flx arrays are cell-shaped (nmb,nvar,N3,N2,N1) as Hydro::uflx
 * (src/hydro/hydro.cpp:290-292) when face_shaped==0, or face-shaped as MHD::uflx when 1. */
int akmi_hydro_fluxes(const akmi_pack *p, int recon, int rsolver, const double *w0,
                      double *flx1, double *flx2, double *flx3, int face_shaped,
                      void *stream);

/* Hydro::RKUpdate / MHD::RKUpdate (src/hydro/hydro_update.cpp:23-83,
 * src/mhd/mhd_update.cpp:24-84):  u0 = gam0*u0 + gam1*u1 - beta_dt*divF */
int akmi_rk_update(const akmi_pack *p, double gam0, double gam1, double beta_dt,
                   double *u0, const double *u1, const double *flx1, const double *flx2,
                   const double *flx3, int face_shaped, void *stream);

/* First stage OUT OF PLACE on the task-granular path: CopyCons (u1 := u0, src/hydro/hydro_tasks.cpp:130-152)
 * followed by RKUpdate in one pass.  u0 is only read; u1 receives, in EVERY cell, what CopyCons + akmi_rk_update would
 * have left in u0 (active cells: gam0*u0 + gam1*u0 - beta_dt*divF, same operands and order; ghost cells: u0).
 * Afterwards the caller swaps the two registers: the old u0 buffer IS the copy CopyCons would have made.
 * akmi_mhd_ct_oop: the same for CopyCons of the face fields + CT (src/mhd/mhd_ct.cpp:23-80). */
int akmi_rk_update_oop(const akmi_pack *p, double gam0, double gam1, double beta_dt, const double *u0, double *u1,
                       const double *flx1, const double *flx2, const double *flx3, int face_shaped, void *stream);
int akmi_mhd_ct_oop(const akmi_pack *p, double gam0, double gam1, double beta_dt, const double *e1, const double *e2,
                    const double *e3, const double *b0x1f, const double *b0x2f, const double *b0x3f, double *b1x1f,
                    double *b1x2f, double *b1x3f, void *stream);

/* IdealHydro::ConsToPrim (src/eos/ideal_hyd.cpp:29-115) over [il,iu]x[jl,ju]x[kl,ku];
 * counters = device int[3] (dfloor,efloor,tfloor), incremented (not reset). */
int akmi_hydro_c2p(const akmi_pack *p, double *u0, double *w0, int il, int iu, int jl,
                   int ju, int kl, int ku, int *counters, void *stream);

/* Hydro::NewTimeStep (src/hydro/hydro_newdt.cpp:30-139): dt3 = device double[3] receiving
 * min over active cells of dx1/(|vx|+cs), dx2/(|vy|+cs), dx3/(|vz|+cs); initial value
 * FLT_MAX as in the reference. */
int akmi_hydro_newdt(const akmi_pack *p, const double *w0, double *dt3, void *stream);

/* ---- MHD tasks -------------------------------------------------------------------- */
/* MHD::Fluxes -> CalculateFluxes<hlld> (src/mhd/mhd_fluxes.cpp:84-266).  flx face-shaped
 * (src/mhd/mhd.cpp:341-343); face EMFs e3x1,e2x1,e1x2,e3x2,e2x3,e1x3 cell-shaped
 * (src/mhd/mhd.cpp:349-354).  Ranges are the reference's CT-extended ranges. */
int akmi_mhd_fluxes(const akmi_pack *p, int recon, int rsolver, const double *w0,
                    const double *bcc0, const double *bx1f, const double *bx2f,
                    const double *bx3f, double *flx1, double *flx2, double *flx3,
                    double *e3x1, double *e2x1, double *e1x2, double *e3x2, double *e2x3,
                    double *e1x3, void *stream);

/* MHD::CornerE (src/mhd/mhd_corner_e.cpp:26-417), Newtonian 1D/2D/3D branches. */
int akmi_mhd_corner_e(const akmi_pack *p, const double *w0, const double *bcc0,
                      const double *e3x1, const double *e2x1, const double *e1x2,
                      const double *e3x2, const double *e2x3, const double *e1x3,
                      const double *flx1, const double *flx2, const double *flx3,
                      double *e1, double *e2, double *e3, void *stream);

/* MHD::CT (src/mhd/mhd_ct.cpp:23-80) */
int akmi_mhd_ct(const akmi_pack *p, double gam0, double gam1, double beta_dt,
                const double *e1, const double *e2, const double *e3, double *b0x1f,
                double *b0x2f, double *b0x3f, const double *b1x1f, const double *b1x2f,
                const double *b1x3f, void *stream);

/* IdealMHD::ConsToPrim (src/eos/ideal_mhd.cpp:30-134) */
int akmi_mhd_c2p(const akmi_pack *p, double *u0, const double *bx1f, const double *bx2f,
                 const double *bx3f, double *w0, double *bcc0, int il, int iu, int jl,
                 int ju, int kl, int ku, int *counters, void *stream);

/* MHD::NewTimeStep (src/mhd/mhd_newdt.cpp:31-174) */
int akmi_mhd_newdt(const akmi_pack *p, const double *w0, const double *bcc0, double *dt3,
                   void *stream);

/* ---- Boundary values (same-level) -------------------------------------------------- */
/* Neighbour table: int nghbr[nmb][27], direction d = (ox3+1)*9 + (ox2+1)*3 + (ox1+1)
 * (the reference's NeighborBlock table, src/mesh/mesh.hpp:47-52, reduced to the
 * same-level case).  Entry >= 0: local index of the neighbour MeshBlock in this pack
 * (same-rank path, src/bvals/bvals_cc.cpp:122-135).  Entry == -1: no neighbour (physical
 * boundary, filled later by akmi_*_bcs).  Entry <= -2: neighbour on another rank; the
 * ghost region is filled from receive-buffer segment  s = -(entry+2)  whose start offset
 * (in doubles) is seg_off[s]. */

/* MeshBoundaryValuesCC::PackAndSendCC + RecvAndUnpackCC, same-rank part
 * (src/bvals/bvals_cc.cpp:42-447; index ranges src/bvals/buffs_cc.cpp:28-70,150-190). */
int akmi_bvals_cc_local(const akmi_pack *p, int nvar, const int *nghbr, double *u,
                        void *stream);
/* off-rank part: send_tab[nsend][2] = {local block m, direction d towards the receiver};
 * send_off[nsend] = start offset (doubles) of that segment in sendbuf. A segment holds
 * [n][k][j][i] of the ng innermost active layers the receiver needs. */
int akmi_bvals_cc_pack(const akmi_pack *p, int nvar, int nsend, const int *send_tab,
                       const long long *send_off, const double *u, double *sendbuf,
                       void *stream);
int akmi_bvals_cc_unpack(const akmi_pack *p, int nvar, const int *nghbr,
                         const long long *seg_off, const double *recvbuf, double *u,
                         void *stream);
/* number of doubles in the CC segment for direction d (per variable) */
long long akmi_bvals_cc_segsize(const akmi_pack *p, int d);

/* MeshBoundaryValuesFC::PackAndSendFC + RecvAndUnpackFC (src/bvals/bvals_fc.cpp:63,289;
 * ranges src/bvals/buffs_fc.cpp:29-110,396-431): shared faces are never exchanged. */
int akmi_bvals_fc_local(const akmi_pack *p, const int *nghbr, double *bx1f, double *bx2f,
                        double *bx3f, void *stream);
int akmi_bvals_fc_pack(const akmi_pack *p, int nsend, const int *send_tab,
                       const long long *send_off, const double *bx1f, const double *bx2f,
                       const double *bx3f, double *sendbuf, void *stream);
int akmi_bvals_fc_unpack(const akmi_pack *p, const int *nghbr, const long long *seg_off,
                         const double *recvbuf, double *bx1f, double *bx2f, double *bx3f,
                         void *stream);
/* doubles in the FC segment for direction d (x1f+x2f+x3f parts, in that order) */
long long akmi_bvals_fc_segsize(const akmi_pack *p, int d);

/* MeshBoundaryValues::HydroBCs / BFieldBCs (src/bvals/physics/hydro_bcs.cpp:28-...,
 * src/bvals/physics/bfield_bcs.cpp:25-...).  bcs = device int[nmb][6] of AKMI_BC_* for
 * inner_x1,outer_x1,inner_x2,outer_x2,inner_x3,outer_x3 (mb_bcs). */
int akmi_hydro_bcs(const akmi_pack *p, int nvar, const int *bcs, double *u, void *stream);
int akmi_bfield_bcs(const akmi_pack *p, const int *bcs, double *bx1f, double *bx2f,
                    double *bx3f, void *stream);
/* the same with the inflow states of MeshBoundaryValues::u_in / b_in (src/bvals/bvals.cpp:323-326):
 * u_in = device double[nvar][6], b_in = device double[3][6], indexed [variable][BoundaryFace].
 * Flags handled: reflect, outflow, inflow, diode (normal velocity clipped to point outwards;
 * field as outflow), vacuum (zeros; field as outflow); AKMI_BC_USER faces are left alone. */
int akmi_hydro_bcs_inflow(const akmi_pack *p, int nvar, const int *bcs, const double *u_in, double *u,
                          void *stream);
int akmi_bfield_bcs_inflow(const akmi_pack *p, const int *bcs, const double *b_in, double *bx1f,
                           double *bx2f, double *bx3f, void *stream);
/* The same with the set of directions that have a physical boundary at all (bit d = direction d; the caller derives it from
 * the flags once): the other directions are not launched.  dirs = 7 equals the entries above; u_in / b_in may be NULL. */
int akmi_hydro_bcs_dirs(const akmi_pack *p, int nvar, const int *bcs, int dirs, const double *u_in, double *u,
                        void *stream);
int akmi_bfield_bcs_dirs(const akmi_pack *p, const int *bcs, int dirs, const double *b_in, double *bx1f, double *bx2f,
                         double *bx3f, void *stream);

/* Same-rank gather AND the physical boundary functions of the pack in ONE launch, for packs without off-rank neighbours
 * (every nghbr entry >= -1): the result equals akmi_bvals_cc_local followed by akmi_hydro_bcs_inflow (resp.
 * akmi_bvals_fc_local followed by akmi_bfield_bcs_inflow) bit for bit.  The reference applies its boundary functions one
 * direction after the other over all transverse indices (hydro_bcs.cpp:69-230, bfield_bcs.cpp:66-300, called from
 * hydro_tasks.cpp:357-375 / mhd_tasks.cpp after the receives); a ghost element therefore holds T3(T2(T1(x))) of ONE source
 * element x, and the kernel fetches that element directly.  u_in / b_in may be NULL when no face is AKMI_BC_INFLOW.
 * dt3_reset (may be NULL): the three CFL minima are set to FLT_MAX by the same launch, so the following
 * akmi_*_c2p_newdt may be called with do_newdt = 2 ("scan, dt3 already reset"). */
int akmi_bvals_cc_local_bcs(const akmi_pack *p, int nvar, const int *nghbr, const int *bcs, const double *u_in, double *u,
                            double *dt3_reset, void *stream);
int akmi_bvals_fc_local_bcs(const akmi_pack *p, const int *nghbr, const int *bcs, const double *b_in, double *bx1f,
                            double *bx2f, double *bx3f, void *stream);

/* ---- SMR/AMR operators between a MeshBlock and its coarse buffer ---------- *
 * Coarse arrays: cnx = nx/2 active cells, the same ng ghost cells, same layout:
 * (nmb,nvar,cN3,cN2,cN1), faces +1 in their own direction (src/mesh/mesh.cpp:286-330).  Index boxes
 * box = {il,iu,jl,ju,kl,ku} are COARSE indices (the iprol boxes of src/bvals/prolongation.cpp);
 * fine index = (coarse - cis)*2 + is.  The mesh tree and the level-aware exchange that drive these
 * operators in the reference are not part of this library yet. */
/* MeshRefinement::RestrictCC (src/mesh/mesh_refinement.cpp:1223-1277) over the active coarse cells */
int akmi_restrict_cc(const akmi_pack *p, int nvar, const double *u, double *cu, void *stream);
/* MeshRefinement::RestrictFC (src/mesh/mesh_refinement.cpp:1283-1382) */
int akmi_restrict_fc(const akmi_pack *p, const double *bx1f, const double *bx2f, const double *bx3f,
                     double *cbx1f, double *cbx2f, double *cbx3f, void *stream);
/* the same for the MeshBlocks m with mask[m] != 0 only (mask on the device, NULL = all).  On a statically refined
 * mesh only a block with a coarser neighbour reads its coarse buffer (sends to that neighbour, prolongation
 * stencils): akmi_smr::needs_coarse is that mask. */
int akmi_restrict_cc_masked(const akmi_pack *p, int nvar, const unsigned char *mask, const double *u, double *cu,
                            void *stream);
int akmi_restrict_fc_masked(const akmi_pack *p, const unsigned char *mask, const double *bx1f, const double *bx2f,
                            const double *bx3f, double *cbx1f, double *cbx2f, double *cbx3f, void *stream);
/* Conservation at fine/coarse faces: what a fine MeshBlock hands to a coarser neighbour.
 * akmi_restrict_flux_cc: the 2x2 (2-D: 2, 1-D: 1) fine face fluxes of direction `dir` behind each coarse
 * face of `box` (coarse indices il,iu,jl,ju,kl,ku; one face thick along dir), in the buffer order of
 * MeshBoundaryValuesCC::PackAndSendFluxCC (src/bvals/flux_correct_cc.cpp:78-148):
 * out[m][(t1-t1l) + n1*((t2-t2l) + n2*v)], (t1,t2) = (j,k) | (i,k) | (i,j).  flx is face-shaped
 * (nmb,nvar,N3(+1),N2(+1),N1(+1)).
 * akmi_restrict_emf: the two fine edges behind each coarse edge of component comp (x1e,x2e,x3e shapes of
 * akmi_mhd_corner_e), MeshBoundaryValuesFC::PackAndSendFluxFC (src/bvals/flux_correct_fc.cpp:84-360):
 * out[m][(i-il) + ni*((j-jl) + nj*(k-kl))]. */
int akmi_restrict_flux_cc(const akmi_pack *p, int nvar, int dir, const int *box, const double *flx,
                          double *out, void *stream);
int akmi_restrict_emf(const akmi_pack *p, int comp, const int *box, const double *e, double *out,
                      void *stream);
/* Primitive -> conserved over the fine cells of `box` (il,iu,jl,ju,kl,ku): SingleP2C_IdealHyd/_IdealMHD/
 * _Isothermal* (src/eos/ideal_c2p_hyd.hpp:76-83, ideal_c2p_mhd.hpp:75-84; scalars u = d*s), the step
 * MeshBoundaryValuesCC::PrimToConsFineBndry performs after primitives were prolongated
 * (src/bvals/prolong_prims.cpp:190-300,465-...).  bcc = NULL for hydro. */
int akmi_prim2cons(const akmi_pack *p, const int *box, const double *w, const double *bcc, double *u,
                   void *stream);
/* ProlongCC (src/mesh/prolongation.hpp:19-63): min-mod limited linear interpolation */
int akmi_prolong_cc(const akmi_pack *p, int nvar, const int *box, const double *cu, double *u,
                    void *stream);
/* ProlongFCSharedX1Face/X2Face/X3Face (src/mesh/prolongation.hpp:69-160): component comp = 0,1,2 */
int akmi_prolong_fc_shared(const akmi_pack *p, int comp, const int *box, const double *cb, double *b,
                           void *stream);
/* ProlongFCInternal (src/mesh/prolongation.hpp:166-230; Toth & Roe 2002), after the shared faces */
int akmi_prolong_fc_internal(const akmi_pack *p, const int *box, double *bx1f, double *bx2f,
                             double *bx3f, void *stream);

/* ---- boundary values of a statically refined pack --------------------------
 * The reference keeps, per MeshBoundaryValues object, 56 MeshBoundaryBuffers with the index ranges
 * of every slot (src/bvals/bvals.hpp:62-107, buffs_cc.cpp, buffs_fc.cpp) and the NeighborBlock table
 * of every MeshBlock (src/mesh/mesh.hpp:47-52).  A caller hands the same information over as flat
 * device tables; the entry points below are the bodies of PackAndSend* / RecvAndUnpack* /
 * FillCoarseInBndry* / Prolongate* / *FluxCC / *FluxFC for neighbours that live in the same pack.
 *   nghbr   [nmb][56][3]   {index of the neighbour block in this pack or -1, its level, dest slot}
 *   mblev   [nmb]          level of each block
 *   cc_tab, fc_tab [2][6][56][3][6]  (send|recv) x (isame, icoar, ifine, iprol, iflux_same,
 *                          iflux_coar) x slot x component x {bis,bie,bjs,bje,bks,bke}; cell-centred
 *                          tables use component 0 only
 *   ndat    [2][56][2][5]  (cc|fc) x slot x (send|recv) x {isame,icoar,ifine,iflxs,iflxc}_ndat
 *   layout  [4][56][2]     (cc vars, cc flux, fc vars, fc flux) x slot x {offset, per-block stride}
 *                          of the receive buffers inside the buffer the caller passes (doubles)
 *   slot_ox [56][3]        offsets (ox1,ox2,ox3) of each slot (inverse of NeighborIndex,
 *                          src/mesh/nghbr_index.hpp:28-54) */
typedef struct akmi_smr {
  int nnghbr;             /* 8 / 24 / 56: src/mesh/meshblock.cpp:145-147 */
  int multilevel;
  const int *nghbr, *mblev, *cc_tab, *fc_tab, *ndat, *slot_ox;
  const long long *layout;
  /* ranks: [4][nmb][56] offsets (doubles) inside the buffer of each class at which block m WRITES
   * the segment of its slot n (soff: the receive segment of a neighbour in this pack, or a place in
   * the message to another rank) and READS the segment it receives (roff).  NULL: everything is
   * in this pack and layout[] alone addresses the buffers.  With them an off-rank neighbour is
   * marked in nghbr by any index >= 0 (it is only tested for existence then). */
  const long long *soff, *roff;
  /* != 0: cell-centred ghost zones whose neighbour has the SAME level and lives in this pack are not packed /
   * unpacked by akmi_smr_*_cc: the caller fills them with akmi_bvals_cc_local (one gather straight from the
   * neighbour's active cells, no buffer round trip; the regions of different slots are disjoint for cell-centred
   * data, so the order does not matter).  0 (a zero-initialised descriptor): every slot goes through the buffers. */
  int direct_same;
  /* [nmb] or NULL: != 0 for blocks with at least one coarser neighbour.  FillCoarseInBndryCC/FC write ghost zones
   * of the COARSE arrays, which only ProlongateCC/FC of such a block ever read; with the table the fill skips the
   * other blocks (NULL: every block, as the reference does; the fine arrays come out the same either way). */
  const unsigned char *needs_coarse;
  /* Work lists (optional; NULL / zero-initialised: every entry point launches over all (MeshBlock, slot) pairs).
   * lists = device array of AKMI_SMR_NLISTS lists of (m, n) int pairs, list l at lists + l*2*nmb*56, list_cnt[l]
   * pairs each, filled by akmi_smr_build_lists once after the tables above are in place.  The entry points then
   * launch over the pairs of the list that covers their own test (existing neighbour; coarser; finer; same level of
   * a block that prolongates; faces and edges that average their EMFs) instead of nmb*56 pairs of which a few per
   * cent do anything.  Results cannot depend on the lists: every kernel still applies its own test. */
  const int *lists;
  int list_cnt[6];
} akmi_smr;
#define AKMI_SMR_NLISTS 6
/* lists: device, 2*nmb*56*AKMI_SMR_NLISTS ints; counts: HOST, AKMI_SMR_NLISTS ints (copy them into list_cnt).
 * Set-up call: synchronises `stream`.  direct_same and needs_coarse of *t must already have their final values. */
int akmi_smr_build_lists(const akmi_pack *p, const akmi_smr *t, int *lists, int *counts, void *stream);
/* RestrictU is akmi_restrict_cc / akmi_restrict_fc above.  SendU+RecvU (PackAndSendCC +
 * RecvAndUnpackCC, src/bvals/bvals_cc.cpp:42-447): u ghost cells from same-level and finer neighbours,
 * coarse buffer cu from coarser ones.  buf: receive buffers (layout[0]) */
int akmi_smr_exchange_cc(const akmi_pack *p, const akmi_smr *t, int nvar, double *u, double *cu,
                         double *buf, void *stream);
/* SendB+RecvB (src/bvals/bvals_fc.cpp:63-436): slots unpacked in order, active faces never
 * overwritten by same-level / finer data.  buf: layout[2] */
int akmi_smr_exchange_fc(const akmi_pack *p, const akmi_smr *t, double *b1, double *b2, double *b3,
                         double *cb1, double *cb2, double *cb3, double *buf, void *stream);
/* FillCoarseInBndryCC / FC (src/bvals/prolongation.cpp:366-462,556-644) */
int akmi_smr_fill_coarse_cc(const akmi_pack *p, const akmi_smr *t, int nvar, const double *u, double *cu,
                            void *stream);
int akmi_smr_fill_coarse_fc(const akmi_pack *p, const akmi_smr *t, const double *b1, const double *b2,
                            const double *b3, double *cb1, double *cb2, double *cb3, void *stream);
/* ProlongateCC / ProlongateFC (src/bvals/prolongation.cpp:470-546,650-785; the latter with the
 * owned-face rule of :90-147) */
int akmi_smr_prolong_cc(const akmi_pack *p, const akmi_smr *t, int nvar, const double *cu, double *u,
                        void *stream);
int akmi_smr_prolong_fc(const akmi_pack *p, const akmi_smr *t, const double *cb1, const double *cb2,
                        const double *cb3, double *b1, double *b2, double *b3, void *stream);
/* <mesh_refinement>/prolong_primitives = true: the two conversions MHD::Prolongate / Hydro::Prolongate put
 * round ProlongateCC (src/mhd/mhd_tasks.cpp:539-544, src/hydro/hydro_tasks.cpp:388-392).
 * ConsToPrimCoarseBndry (src/bvals/prolong_prims.cpp:35-186, 303-461): coarse conserved -> coarse primitive
 * variables on the cells the prolongation of every slot with a coarser neighbour reads (iprol widened by one);
 * cb1..3 = coarse face fields (NULL: hydro); cu is only modified where a passive scalar is negative.
 * PrimToConsFineBndry (:190-296, 465-575): the prolongated fine ghost cells of w back to conserved variables
 * in u; b1..3 = fine face fields after ProlongateFC (NULL: hydro).  Ideal gas only, as in the reference. */
int akmi_smr_c2p_coarse(const akmi_pack *p, const akmi_smr *t, int nvar, double *cu, const double *cb1,
                        const double *cb2, const double *cb3, double *cw, void *stream);
int akmi_smr_p2c_fine(const akmi_pack *p, const akmi_smr *t, int nvar, const double *w, const double *b1,
                      const double *b2, const double *b3, double *u, void *stream);
/* SendFlux+RecvFlux (PackAndSendFluxCC + RecvAndUnpackFluxCC, src/bvals/flux_correct_cc.cpp:29-304):
 * the fluxes on faces shared with finer neighbours are replaced by the restricted fine fluxes.
 * face_shaped: MHD flux arrays (N+1 along their direction), 0: hydro (cell-shaped).  buf: layout[1] */
int akmi_smr_flux_cc(const akmi_pack *p, const akmi_smr *t, int nvar, int face_shaped, double *flx1,
                     double *flx2, double *flx3, double *buf, void *stream);
/* SendE+RecvE (PackAndSendFluxFC + RecvAndUnpackFluxFC, src/bvals/flux_correct_fc.cpp:29-1034): edge
 * EMFs on block surfaces summed over same-level owners, replaced by the restricted EMFs of finer
 * neighbours, averaged.  nflx [nmb][48]: contributions per block edge (the counting of
 * SumBoundaryFluxes / ZeroFluxesAtBoundaryWithFiner, a function of the neighbour table).  buf: layout[3] */
int akmi_smr_emf_exchange(const akmi_pack *p, const akmi_smr *t, const int *nflx, double *e1, double *e2,
                          double *e3, double *buf, void *stream);
/* The four exchanges cut at the point where the reference posts its MPI messages: *_pack_* = the
 * PackAndSend* half (fills the segments soff names), *_unpack_* = the RecvAndUnpack* half (reads the
 * segments roff names; for the EMFs: sum, zero, sum, average).  A caller with off-rank neighbours
 * moves the per-rank slices of buf between the two calls. */
int akmi_smr_pack_cc(const akmi_pack *p, const akmi_smr *t, int nvar, const double *u, const double *cu,
                     double *buf, void *stream);
int akmi_smr_unpack_cc(const akmi_pack *p, const akmi_smr *t, int nvar, const double *buf, double *u,
                       double *cu, void *stream);
int akmi_smr_pack_fc(const akmi_pack *p, const akmi_smr *t, const double *b1, const double *b2,
                     const double *b3, const double *cb1, const double *cb2, const double *cb3, double *buf,
                     void *stream);
int akmi_smr_unpack_fc(const akmi_pack *p, const akmi_smr *t, const double *buf, double *b1, double *b2,
                       double *b3, double *cb1, double *cb2, double *cb3, void *stream);
int akmi_smr_pack_flux_cc(const akmi_pack *p, const akmi_smr *t, int nvar, int face_shaped, const double *flx1,
                          const double *flx2, const double *flx3, double *buf, void *stream);
int akmi_smr_unpack_flux_cc(const akmi_pack *p, const akmi_smr *t, int nvar, int face_shaped, const double *buf,
                            double *flx1, double *flx2, double *flx3, void *stream);
int akmi_smr_pack_emf(const akmi_pack *p, const akmi_smr *t, const double *e1, const double *e2,
                      const double *e3, double *buf, void *stream);
int akmi_smr_unpack_emf(const akmi_pack *p, const akmi_smr *t, const int *nflx, const double *buf, double *e1,
                        double *e2, double *e3, void *stream);

/* The face-field exchange of a refined mesh as one list of element copies.  PackAndSendFC + RecvAndUnpackFC
 * (src/bvals/bvals_fc.cpp:63-289, :300-436) only move values: which face of which array a ghost face receives -- with
 * the precedence of the reference's slot-by-slot unpack where slots overlap, and never a face the block owns -- depends
 * on the mesh alone.  akmi_smr_fc_map (set-up call, synchronises `stream`) finds the copies by running
 * akmi_smr_pack_fc [+ akmi_smr_unpack_fc] once on scratch arrays whose elements hold their own index, and returns their
 * number (-1: failure); map == NULL: count only; otherwise map receives (destination, source) pairs of 32-bit indices
 * into the concatenation [b1 | b2 | b3 | cb1 | cb2 | cb3 | buf], sorted by destination (cap = pairs it has room for).
 *   which = 0: what the unpack does (sources: arrays of this pack, or the part of buf other ranks fill);
 *   which = 1: the outgoing messages, buf[send_lo, send_hi) <- arrays (ranks only; 0 pairs on one rank).
 * The reference packs every message before it unpacks any, so a copy reads the value its source had BEFORE the exchange;
 * the few copies whose source is itself a destination (surface faces of a fine block's coarse array) are the last *ntail
 * pairs of the list -- akmi_smr_fc_copy performs them first, in a launch of their own -- and their place in the ordered
 * part holds a no-op (destination == source).
 * buf (the class-2 buffer of the exchange, buf_doubles long) is used as scratch and zeroed.  akmi_smr_fc_copy performs
 * a list: one rank -- the which = 0 list in place of akmi_smr_pack_fc + akmi_smr_unpack_fc; ranks -- the which = 1
 * list, the transfer, the which = 0 list. */
long long akmi_smr_fc_map(const akmi_pack *p, const akmi_smr *t, double *buf, long long buf_doubles, long long send_lo,
                          long long send_hi, int which, int *map, long long cap, long long *ntail, void *stream);
int akmi_smr_fc_copy(const akmi_pack *p, const int *map, long long npairs, long long ntail, long long buf_doubles,
                     double *b1, double *b2, double *b3, double *cb1, double *cb2, double *cb3, double *buf,
                     void *stream);

/* The same for the cell-centred variables (PackAndSendCC + RecvAndUnpackCC, src/bvals/bvals_cc.cpp:42-447, as this
 * library performs them: akmi_smr_pack_cc + akmi_smr_unpack_cc across levels, then the direct same-level gather
 * akmi_bvals_cc_local with the [nmb][27] table same27): the copies of the nvar variables of a cell are one copy nvar times,
 * so the list holds the pairs of variable 0 in the index space [u | cu] and akmi_smr_cc_copy applies it to every variable.
 * One rank only (returns -1 when akmi_smr::soff / roff are set: the caller keeps the three calls then). */
long long akmi_smr_cc_map(const akmi_pack *p, const akmi_smr *t, int nvar, const int *same27, double *buf,
                          long long buf_doubles, int *map, long long cap, long long *ntail, void *stream);
int akmi_smr_cc_copy(const akmi_pack *p, int nvar, const int *map, long long npairs, long long ntail, double *u,
                     double *cu, void *stream);

/* ---- Fused fast path ("one kernel sequence per MeshBlockPack stage") ----------------- *
 * Must produce results identical to the task chain above.  ws = device workspace of
 * akmi_stage_workspace_bytes() bytes owned by the caller. */
long long akmi_stage_workspace_bytes(const akmi_pack *p, int is_mhd);

/* Hydro: Fluxes + RKUpdate fused (stage 1 reads u0 as u1 when u1==NULL is not allowed;
 * CopyCons semantics are folded in when copy_u1 != 0: u1 <- u0 before the update).
 * copy_u1 == 2: the first stage OUT OF PLACE -- u0 (and b0) are only read, the new state is written
 * to u1 (and b1), and the caller swaps the two registers (pointers) after the call; the copy of
 * Hydro::CopyCons / MHD::CopyCons (5 + 3 arrays written) disappears.  Only the cells and faces the update
 * touches are written; the ghost zones of the new register are filled by the halo exchange and the
 * boundary conditions that follow every stage.  With akmi_*_stage_phase the swap of u follows the
 * SWEEPS part and the swap of b the EMF_CT part (the C2P part is passed the swapped pointers). */
int akmi_hydro_stage_update(const akmi_pack *p, int recon, int rsolver, double gam0,
                            double gam1, double beta_dt, int copy_u1, const double *w0,
                            double *u0, double *u1, void *ws, void *stream);
/* MHD: Fluxes + CornerE + RKUpdate + CT fused */
int akmi_mhd_stage_update(const akmi_pack *p, int recon, int rsolver, double gam0,
                          double gam1, double beta_dt, int copy_u1, const double *w0,
                          const double *bcc0, double *u0, double *u1, double *b0x1f,
                          double *b0x2f, double *b0x3f, double *b1x1f, double *b1x2f,
                          double *b1x3f, void *ws, void *stream);
/* ConsToPrim + (optionally, last stage) NewTimeStep fused.  do_newdt: 0 no scan, 1 reset dt3 and scan, 2 scan only (dt3 was
 * reset earlier in stream order, e.g. by akmi_bvals_cc_local_bcs) */
int akmi_hydro_c2p_newdt(const akmi_pack *p, double *u0, double *w0, int do_newdt,
                         int *counters, double *dt3, void *stream);
int akmi_mhd_c2p_newdt(const akmi_pack *p, double *u0, const double *bx1f,
                       const double *bx2f, const double *bx3f, double *w0, double *bcc0,
                       int do_newdt, int *counters, double *dt3, void *stream);

/* Whole stage in one call: pass A (as akmi_*_stage_update) + ConsToPrim of the ACTIVE cells
 * (+ the CFL scan when do_newdt) in the same call.  w0/bcc0 are read (old primitives) and
 * rewritten for the active cells.  After the halo exchange / physical BCs the caller converts
 * the ghost shell with akmi_*_c2p_shell; together they equal akmi_*_c2p_newdt over all cells
 * (ConToPrim covers ghosts: src/hydro/hydro_tasks.cpp:404-412). */
int akmi_hydro_stage_fused(const akmi_pack *p, int recon, int rsolver, double gam0,
                           double gam1, double beta_dt, int copy_u1, double *w0, double *u0,
                           double *u1, int do_newdt, int *counters, double *dt3, void *ws,
                           void *stream);
int akmi_mhd_stage_fused(const akmi_pack *p, int recon, int rsolver, double gam0, double gam1,
                         double beta_dt, int copy_u1, double *w0, double *bcc0, double *u0,
                         double *u1, double *b0x1f, double *b0x2f, double *b0x3f,
                         double *b1x1f, double *b1x2f, double *b1x3f, int do_newdt,
                         int *counters, double *dt3, void *ws, void *stream);
/* The same with the time step in DEVICE memory: the kernels form beta*(*dt_dev) themselves (the
 * product RKUpdate forms on the host, hydro_update.cpp:35 -- same operands, same rounding).  No
 * argument of the call changes from cycle to cycle then, so a whole cycle can be captured into a
 * hipGraph once and replayed (the C++ host does: akmi_host.cpp Driver::Execute). */
int akmi_hydro_stage_fused_dt(const akmi_pack *p, int recon, int rsolver, double gam0, double gam1,
                              double beta, const double *dt_dev, int copy_u1, double *w0, double *u0,
                              double *u1, int do_newdt, int *counters, double *dt3, void *ws,
                              void *stream);
int akmi_mhd_stage_fused_dt(const akmi_pack *p, int recon, int rsolver, double gam0, double gam1,
                            double beta, const double *dt_dev, int copy_u1, double *w0, double *bcc0,
                            double *u0, double *u1, double *b0x1f, double *b0x2f, double *b0x3f,
                            double *b1x1f, double *b1x2f, double *b1x3f, int do_newdt, int *counters,
                            double *dt3, void *ws, void *stream);
/* The same stage cut into its parts for callers that post halo messages in between (a rank
 * with off-rank neighbours sends u0 after SWEEPS and b0 after EMF_CT, so the transfers run
 * under CornerE/CT and under the c2p of the active cells, in the order of the reference's
 * task list: RKUpdate -> SendU -> EField -> CT -> SendB, src/mhd/mhd_tasks.cpp:48-75).
 * phases is a mask of AKMI_PHASE_*; the parts must be called in increasing order within a
 * stage with identical arguments; AKMI_PHASE_ALL equals akmi_*_stage_fused. */
enum { AKMI_PHASE_SWEEPS = 1,   /* reconstruct + Riemann x1..x3 + RK update of u0 (+ face EMFs) */
       AKMI_PHASE_EMF_CT = 2,   /* CornerE + CT of b0 (MHD only; ignored for hydro)            */
       AKMI_PHASE_C2P    = 4,   /* ConsToPrim of the active cells (+ CFL scan if do_newdt)     */
       AKMI_PHASE_ALL    = 7 };
int akmi_hydro_stage_phase(const akmi_pack *p, int recon, int rsolver, double gam0,
                           double gam1, double beta_dt, int copy_u1, double *w0, double *u0,
                           double *u1, int do_newdt, int *counters, double *dt3, int phases,
                           void *ws, void *stream);
int akmi_mhd_stage_phase(const akmi_pack *p, int recon, int rsolver, double gam0, double gam1,
                         double beta_dt, int copy_u1, double *w0, double *bcc0, double *u0,
                         double *u1, double *b0x1f, double *b0x2f, double *b0x3f,
                         double *b1x1f, double *b1x2f, double *b1x3f, int do_newdt,
                         int *counters, double *dt3, int phases, void *ws, void *stream);
/* the parts with dt read from device memory (beta = the RK weight alone, as in akmi_*_stage_fused_dt): for callers that
 * enqueue a cycle before the previous cycle's new time step has reached the host */
int akmi_hydro_stage_phase_dt(const akmi_pack *p, int recon, int rsolver, double gam0, double gam1, double beta,
                              const double *dt_dev, int copy_u1, double *w0, double *u0, double *u1, int do_newdt,
                              int *counters, double *dt3, int phases, void *ws, void *stream);
int akmi_mhd_stage_phase_dt(const akmi_pack *p, int recon, int rsolver, double gam0, double gam1, double beta,
                            const double *dt_dev, int copy_u1, double *w0, double *bcc0, double *u0, double *u1,
                            double *b0x1f, double *b0x2f, double *b0x3f, double *b1x1f, double *b1x2f, double *b1x3f,
                            int do_newdt, int *counters, double *dt3, int phases, void *ws, void *stream);
/* Hydro, the whole stage with ConsToPrim of the active cells INSIDE the update kernel (3-D, DC / PLM, ideal gas, no passive
 * scalars: akmi_hydro_stage_w_eligible returns 1): the kernel that finishes a cell holds its new conserved state in registers,
 * so the conversion (src/eos/ideal_c2p_hyd.hpp:22-66, floors and counters included) and, with do_newdt, the CFL scan
 * (src/hydro/hydro_newdt.cpp:97-118) cost five stores there instead of a pass that reads u0 back.  Other workgroups still
 * read w0 while one finishes, so the new primitives of the ACTIVE cells go to w0_new, an array of w0's shape, and
 * *wrote_new = 1: the caller uses w0_new as w0 from then on (swap the two) and fills the ghost cells of u0 and of it with
 * akmi_hydro_ghost_uw (NOT by converting the ghost copies of the floored u0 again: see there).  Where the kernel does not apply the call is akmi_hydro_stage_fused[_dt]
 * (active cells converted in place in w0, *wrote_new = 0).  beta: the RK weight when dt_dev != NULL (dt read from device
 * memory), beta*dt otherwise.  do_newdt: 0 no scan, 1 reset dt3 and scan, 2 scan (the caller has reset dt3). */
int akmi_hydro_stage_w_eligible(const akmi_pack *p, int recon, int rsolver);
int akmi_hydro_stage_w(const akmi_pack *p, int recon, int rsolver, double gam0, double gam1, double beta,
                       const double *dt_dev, int copy_u1, double *w0, double *w0_new, double *u0, double *u1,
                       int do_newdt, int *counters, double *dt3, void *ws, void *stream, int *wrote_new);
int akmi_hydro_c2p_shell(const akmi_pack *p, double *u0, double *w0, int *counters,
                         void *stream);
/* What is left of a hydro stage after akmi_hydro_stage_w on a pack without off-rank neighbours: the ghost zones of u0 AND of
 * the new primitive array, one launch, no conversion.  The reference converts every cell after the ghost fill
 * (hydro_tasks.cpp:357-375,404-412), so a ghost cell's (u, w) equals its source cell's under the boundary's value rule
 * (copy for neighbour / periodic / outflow, sign of the normal momentum and velocity for reflect) -- floors included, whereas
 * converting the copy of an already floored u a second time would not reproduce it.  bcs: device int[nmb][6]; bcs_host: the
 * same flags in host memory -- a pack with a diode, vacuum, inflow or user face is refused (AKMI_FAIL): keep
 * akmi_bvals_cc_local_bcs + akmi_hydro_c2p_newdt (no akmi_hydro_stage_w) there.  ws: the workspace akmi_hydro_stage_w was
 * given (it leaves one floor-flag byte per cell at its start); counters: the three floor counters -- every ghost image of a
 * cell a floor acted on is counted, as the reference's ConsToPrim over all cells counts it. */
int akmi_hydro_ghost_uw(const akmi_pack *p, const int *nghbr, const int *bcs, const int *bcs_host, double *u0, double *w0,
                        const void *ws, int *counters, void *stream);
int akmi_mhd_c2p_shell(const akmi_pack *p, double *u0, const double *bx1f, const double *bx2f,
                       const double *bx3f, double *w0, double *bcc0, int *counters,
                       void *stream);

/* History sums (HistoryOutput::LoadHydroHistoryData / LoadMHDHistoryData,
 * src/outputs/history.cpp:78-160,272-374): out[0..7] = volume sums over the active cells of
 * d, M1, M2, M3, E, and the kinetic energies 0.5*Mi^2/d; MHD adds out[8..10] = the magnetic
 * energies 0.25*(B_face^2 + B_face+1^2) per direction.  out is a device array of 8 (11)
 * doubles; it is zeroed by the call. */
int akmi_history_sums(const akmi_pack *p, int is_mhd, const double *u0, const double *bx1f,
                      const double *bx2f, const double *bx3f, double *out, void *stream);

/* ---- native host driver (C++ mirror of Mesh/MeshBlockPack/TaskList/Driver/Hydro/MHD) ----- *
 * athenak_amd/csrc/akmi_host.{hpp,cpp}: the reference's operator surface for this path in
 * C++, every task body one call of the entries above.  One process per GPU: with akmi_comm_init_*
 * called first the Z-ordered MeshBlock list is cut into one pack per rank (Mesh::LoadBalance,
 * src/mesh/load_balance.cpp:38-88) and the arrays of akmi_sim_array are those of THIS rank's pack
 * (blocks akmi_sim_gids .. +akmi_sim_nmb_thisrank-1); without it: one rank, all MeshBlocks in one
 * pack on the current GPU.  deck_text is an athinput deck (src/parameter_input.cpp grammar).  Initial conditions are written by the caller into the arrays returned by
 * akmi_sim_array (device pointers, layouts as above) before akmi_sim_initialize, which performs
 * Driver::Initialize (src/driver/driver.cpp:314-371); tlim_override > 0 replaces <time>/tlim
 * (the linear-wave generator rescales it).  akmi_sim_execute runs Driver::Execute for at most
 * max_cycles (<0: until tlim/nlim) and returns the number of cycles done; both return with the
 * device work finished.  stream = NULL: the simulation runs on a stream of its own (the legacy null
 * stream cannot be captured into a hipGraph) and waits for the device at entry.  On one rank with
 * the fused stage a cycle can be captured once and replayed as a hipGraph with dt read from device
 * memory: <time>/cycle_graph = auto (default: 1-D packs, where it pays) | true | false;
 * AKMI_CYCLE_GRAPH=0/1 overrides the deck.
 * Pointers returned by akmi_sim_array stay valid for the lifetime of the simulation: inside a cycle the
 * driver trades its two registers (u0/u1, b0/b1) after the out-of-place first stage, and
 * akmi_sim_execute copies the state back into the register of that name before it returns whenever
 * an odd number of trades has happened (one device-to-device copy per call, not per cycle).
 * akmi_sim_profile(sim, 1): from now on a HIP event pair is recorded on the launch stream around every
 * akmi_*_stage_fused / akmi_*_stage_phase call (the dominant launch group of a stage);
 * akmi_sim_profile_read waits for the recorded events, returns their summed duration in ms and their
 * number, and clears the record (bench.py: roofline.ms_per_launch measured inside the timed loop). */
void *akmi_sim_create(const char *deck_text, void *stream);
int akmi_sim_initialize(void *sim, double tlim_override);
int akmi_sim_execute(void *sim, int max_cycles);
int akmi_sim_profile(void *sim, int on);
int akmi_sim_profile_read(void *sim, double *ms_total, long long *calls);
void akmi_sim_destroy(void *sim);
double akmi_sim_time(void *sim);
double akmi_sim_dt(void *sim);
double akmi_sim_tlim(void *sim);
int akmi_sim_ncycle(void *sim);
int akmi_sim_nmb(void *sim);
void *akmi_sim_array(void *sim, const char *name, long long *count);
const int *akmi_sim_lloc(void *sim);
int akmi_sim_gids(void *sim);            /* first global block id of this rank's pack */
int akmi_sim_nmb_thisrank(void *sim);

/* ---- ranks (global_variable::my_rank/nranks + the MPI calls of the reference's hot path:
 * one Isend/Irecv per peer rank and variable class, src/bvals/bvals.cpp:134-310,
 * bvals_cc.cpp:247-303; MPI_Allreduce(MIN) of dt, src/mesh/mesh.cpp:634-637) ------------------- *
 * akmi_comm_init_rccl: RCCL over xGMI, one GPU per rank (hipSetDevice before the call).  id = the
 * 128 bytes rank 0 obtained from akmi_comm_unique_id and handed to every rank by whatever launched
 * the job (MPI_Bcast, a torch.distributed store, a file).  Messages are ncclSend/ncclRecv pairs
 * grouped per variable class on the communicator's own stream, ordered against the compute stream
 * with events; dt is reduced with ncclAllReduce(ncclMin).  akmi_comm_init_env does the same for a
 * job started with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment (the torchrun
 * convention): rank 0 serves the id on MASTER_PORT+1 over TCP.
 * akmi_comm_init_callbacks: any other transport.  exchange() receives HOST buffers (the library
 * stages device data through pinned memory) and returns when all npeer receives are complete;
 * allreduce_min() reduces n host doubles in place over all ranks.
 * All return AKMI_COMPLETE or AKMI_FAIL (akmi_last_error()).  akmi_comm_finalize releases the
 * communicator; akmi_sim_create reads the rank layout at the time it is called. */
typedef int (*akmi_comm_exchange_fn)(void *user, int npeer, const int *peers,
                                     const double *const *sendptr, const long long *sendcount,
                                     double *const *recvptr, const long long *recvcount);
typedef int (*akmi_comm_allreduce_min_fn)(void *user, double *vals, int n);
int akmi_comm_unique_id(char id[128]);
int akmi_comm_init_rccl(int rank, int nranks, const char id[128]);
int akmi_comm_init_env(void);
int akmi_comm_init_callbacks(int rank, int nranks, akmi_comm_exchange_fn exchange,
                             akmi_comm_allreduce_min_fn allreduce_min, void *user);
int akmi_comm_finalize(void);
/* min over all ranks of n <= 8 host doubles, in place (RCCL: ncclAllReduce on `stream`) */
int akmi_comm_allreduce_min(double *vals, int n, void *stream);
int akmi_comm_rank(void);
int akmi_comm_nranks(void);
/* Where a multi-rank stage spends its exchange (src/bvals/bvals_cc.cpp:108-135,247-258 pack / send / receive / unpack,
 * src/mesh/mesh.cpp:634-637 the dt reduction): akmi_comm_profile(1) starts recording HIP event pairs on the compute stream
 * round (0) the pack kernels of the off-rank segments, (1) the point where the compute stream waits for the receives of a
 * channel -- the time it really stalls there, zero when the transfer finished under the kernels enqueued before it --,
 * (2) the unpack kernels, (3) the dt all-reduce + read-back of a cycle.  akmi_comm_profile_read waits for the recorded
 * events, clears the record and fills out[0..11] = {pack_ms, pack_calls, exposed_wait_ms, wait_calls, unpack_ms,
 * unpack_calls, dt_reduce_ms, dt_calls, bytes_sent, posts, peers (most per post), ranks the communicator reports
 * (ncclCommCount; 0 without RCCL)}.  n = capacity of out (>= 12). */
int akmi_comm_profile(int on);
int akmi_comm_profile_read(double *out, int n);
/* The exchange plan rank `rank` of `nranks` derives from the deck alone -- host code only, no
 * device needed (tests/test_host_plan.py compares it with the Python host's plan).  fc = 0: the
 * cell-centred channel with nvar variables, 1: the face-centred channel.  out (capacity cap,
 * long long) receives: npeer, peers[npeer], {send_a, send_b, recv_a, recv_b}[npeer], nmb,
 * tab[nmb*27], nsend, send_tab[nsend*2], send_off[nsend], nseg, seg_off[nseg].  Returns the
 * number of entries needed (call again with a larger buffer if > cap), or -1 on error. */
long long akmi_host_exchange_plan(const char *deck_text, int rank, int nranks, int nvar, int fc,
                                  long long *out, long long cap);

/* ---- measurement utility ------------------------------------------------------------ *
 * dst[i] = src[i] for n doubles with the library's own access pattern (8 B per lane,
 * 512 B per wave, grid-stride): a kernel of KNOWN traffic (8n read + 8n written) used to
 * calibrate the rocprofv3 FETCH_SIZE/WRITE_SIZE counters on gfx950 (tools/pmc.sh). */
int akmi_calib_copy(double *dst, const double *src, long long n, void *stream);

/* ---- arithmetic self-test ------------------------------------------------------------- *
 * The stage kernels evaluate sqrt(x), 1/x and x/dx (dx a power of two) with shorter instruction
 * sequences than the compiler's expansions (csrc/akmi_numerics.hpp: sqrt_x, rcp_x; pow2_shift);
 * these must return the same bits, because the path is bit-comparable with the reference's CPU
 * build (src/eos/eos.hpp:49-57, src/mhd/rsolvers/hlld_mhd.hpp:120-160, src/mhd/mhd_update.cpp:57-80
 * are the expressions concerned).  mode 0: sqrt, 1: reciprocal, 2: division by a power of two.
 * n operands (random in- and out-of-window patterns + a fixed edge table) are evaluated both ways on
 * the device; *mismatch receives the number of operands whose results differ in any bit,
 * *shortform_waves (may be NULL) the number of wave-iterations that really ran the short form. */
int akmi_selftest_fp64(int mode, long long n, unsigned long long seed, long long *mismatch,
                       long long *shortform_waves, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AKMI_H_ */
