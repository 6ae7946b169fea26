// akmi_mhd_stage3d.hpp -- MHD, 3-D, PLM + HLLD: the three sweeps and the RK update of a stage in ONE kernel
// (included by akmi_stage.hip inside namespace akmi, after k_hydro_stage3d whose helpers it shares).
//
// Replaces, for one RK stage of a MeshBlockPack, the reference sequence MHD::CalculateFluxes (all three
// directions, src/mhd/mhd_fluxes.cpp:84-266) + MHD::RKUpdate (src/mhd/mhd_update.cpp:24-84) and the two-kernel
// form k_sweep12s + k_sweep_update<2> of rounds 2-5: no partial-divergence array `acc`, no x1/x2 flux arrays,
// w0 / bcc0 read once (+ tile halo, L2 hits: neighbouring tiles share an XCD).
//
// A workgroup owns a tile of (tw-1) x (th-1) cell columns of the CT-extended range [is-1,ie+1] x [js-1,je+1]
// (lanes flattened over the tw x th positions; the last column / row of positions only provides the face on its low
// side) and marches along k over the planes [ks-1,ke+1].  Per step k (two barriers):
//   (A) plane k-1 of the eight primitives (w0 five, bcc0 three) sits in LDS (2-low / 1-high halo in i and j).  Every
//       cell of it is reconstructed ONCE per in-plane direction (seven variables each: the normal field is the face
//       field): its position keeps the value at the cell's low face (right state of its own face) and hands the
//       value at the high face to the position above through LDS (that position's left state).
//   (B) every position solves its low x1 and x2 face (HLLD); the five fluid fluxes replace the left state in the same
//       LDS slot, the two face EMFs and the mass flux go to memory (CornerE), and so does the cell-centred
//       E = -(v x B) of the cell.  The x3 face below cell k comes from registers (own cells k-1 [LDS entry of the
//       position], k, k+1, pending left state).  Plane k replaces plane k-1.
//   (C) cell k-1 is finished: divf = dF1/dx1; divf += dF2/dx2; divf += dF3/dx3; u0 = gam0*u0 + gam1*u1 - beta_dt*divf
//       (mhd_update.cpp:50-81 order, the rounding sequence of the other paths).
// Face ranges (mhd_fluxes.cpp:117-248): x1 faces i in [is,ie+1] on (j,k) in [js-1,je+1] x [ks-1,ke+1]; x2 faces
// j in [js,je+1] on (i,k) in [is-1,ie+1] x [ks-1,ke+1]; x3 faces k in [ks,ke+1] on (i,j) in [is-1,ie+1] x [js-1,je+1].
// Compiled for TWO waves per SIMD (256 registers: HLLD alone keeps 116-122 alive, profiles/r04_hlld_liveness.txt).

#ifndef AKMI_MS_WAVES
#define AKMI_MS_WAVES 2
#endif
#ifndef AKMI_MS_EO
#define AKMI_MS_EO 1              // wave-uniform early-outs of HLLD
#endif
#ifndef AKMI_MS_FM
#define AKMI_MS_FM 1              // short square roots (sqrt_x)
#endif
#if defined(AKMI_MS_WHATIF) && AKMI_MS_WHATIF != 0 && !defined(AKMI_EXPERIMENTS)
#error "AKMI_MS_WHATIF builds give wrong results: define AKMI_EXPERIMENTS too (akmi_build_flags() then says so)"
#endif
#ifndef AKMI_MS_WHATIF
#define AKMI_MS_WHATIF 0          // timing experiments (wrong results): 1 no barriers, 2 no global stores, 4 no extra halo cells, 8 no x3 solve
#endif
#define MS_STU(b, o, v) do { if (!(AKMI_MS_WHATIF & 16) || u.copy_u1 == 77) stu(b, o, v); } while (0)
#define MS_SYNC() do { if (!(AKMI_MS_WHATIF & 1)) __syncthreads(); } while (0)
#ifndef AKMI_MS_THREADS
#define AKMI_MS_THREADS 256
#endif
constexpr int MS_THREADS = AKMI_MS_THREADS;
constexpr int MS_PE = 9;            // doubles per plane entry (eight used; odd stride: conflict-free 64-bit accesses)
constexpr int MS_FE = 7;            // doubles per face entry
static size_t mhd_lds_doubles(int tw, int th) { return 2*MS_PE*((size_t)(tw + 3)*(th + 3)) + 2*MS_FE*(size_t)tw*th; }

struct MhdTile { int tw, th, n1, n2, threads; };
static MhdTile mhd_tile(int c1, int c2) {
  // c1 x c2 cell columns to own ((nx1+2) x (nx2+2)); same cost model as hyd_tile: lanes launched per owned column
  static int f_tw = -1, f_th = 0;                        // AKMI_MS_TILE=tw,th pins the shape (experiments)
  if (f_tw < 0) {
    const char *e = getenv("AKMI_MS_TILE");
    f_tw = 0;
    if (e && sscanf(e, "%d%*[,x]%d", &f_tw, &f_th) != 2) f_tw = 0;
    if (f_tw < 4 || f_th < 3 || f_tw*f_th > MS_THREADS) f_tw = 0;
  }
  static const int maxlds = getenv("AKMI_MS_LDS") ? atoi(getenv("AKMI_MS_LDS")) : (MS_THREADS > 256 ? 160 : 80)*1024;
  if (f_tw > 0 && 3*(f_tw + 3) + 3*f_th <= f_tw*f_th && mhd_lds_doubles(f_tw, f_th)*sizeof(double) <= 150*1024)
    return MhdTile{f_tw, f_th, (c1 + f_tw - 2)/(f_tw - 1), (c2 + f_th - 2)/(f_th - 1), (f_tw*f_th + 63)/64*64};
  MhdTile best{0, 0, 0, 0, 0};
  double best_cost = -1.0;
  for (int n1 = 1; n1 <= c1; ++n1) {
    const int tw = (c1 + n1 - 1)/n1 + 1;
    if (tw > 130) continue;
    if (tw < 8 && n1 > 1) break;
    for (int th = 3; th <= 40; ++th) {
      if (tw*th > MS_THREADS) break;
      if (3*(tw + 3) + 3*th > tw*th) continue;         // one halo entry per thread at most
      if (tw + th > tw*th) continue;
      const size_t lds = mhd_lds_doubles(tw, th)*sizeof(double);
      if (lds > (size_t)maxlds) continue;
      const int n2 = (c2 + th - 2)/(th - 1);
      const int threads = (tw*th + 63)/64*64;
      const double halo = (double)(tw + 3)*(th + 3)/((double)(tw - 1)*(th - 1));
      const double cost = (double)n1*n2*threads*(1.0 + 0.1*halo)*(1.0 + 2.0/tw);
      if (best_cost < 0 || cost < best_cost) { best = MhdTile{tw, th, n1, n2, threads}; best_cost = cost; }
    }
  }
  return best;
}

struct MhdStageArgs {
  const double *w0, *bcc0;
  const double *bx1f, *bx2f, *bx3f;      // face fields of the stage's input state (normal field of each sweep)
  double *mf1, *mf2, *mf3;               // face-shaped flux arrays: variable 0 (the mass flux) is written
  double *e3x1, *e2x1, *e1x2, *e3x2, *e2x3, *e1x3;     // face EMFs (cell-shaped arrays)
  double *ecc1, *ecc2, *ecc3;            // cell-centred EMFs
};

// COPY: u.copy_u1 != 0 (first stage: the second register is not read)
template <int RS, bool COPY>
__global__ void __launch_bounds__(MS_THREADS, AKMI_MS_WAVES)
k_mhd_stage3d(Geo g, FaceEos eos, MhdStageArgs a, UpdArgs u, int nchunk, int ckl, int tw, int th) {
  constexpr bool EO = AKMI_MS_EO != 0, FM = AKMI_MS_FM != 0;
  extern __shared__ double ms_lds[];
  constexpr int PE = MS_PE, FE = MS_FE;
  const int pw = tw + 3, ph = th + 3;        // plane with halo: cols i0-2..i0+tw, rows j0-2..j0+th
  const int qn = ph*pw, fn = th*tw;
  const int PLSZ = PE*qn;                    // doubles per plane; two planes (k-1 and k), then the two face arrays
  const int tid = threadIdx.x;
  const int r = tid/tw, t = tid - r*tw;
  const bool in_tile = r < th;
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {                       // tiles of a k-chunk of a block side by side on one XCD (x fastest, then y, then chunk / block)
    const unsigned lin = xcd_order(bx + gridDim.x*(by + gridDim.y*bz), gridDim.x*gridDim.y*gridDim.z);
    const unsigned row = lin/gridDim.x;
    bx = lin - row*gridDim.x; bz = row/gridDim.y; by = row - bz*gridDim.y;
  }
  const int i0 = g.is - 1 + (int)bx*(tw - 1), j0 = g.js - 1 + (int)by*(th - 1);
  const int i = i0 + t, j = j0 + r;
  const int m = (int)bz/nchunk;
  const int ch = (int)bz - m*nchunk;
  const int kA = g.ks - 1, kB = g.ke + 1;                               // planes of the in-plane faces
  const int k0 = kA + ch*ckl;
  const int k1 = (k0 + ckl - 1 < kB) ? k0 + ckl - 1 : kB;
  const bool cell_ok = in_tile && i < g.N1 && j < g.N2;                // the column exists in memory
  const bool own = !(AKMI_MS_WHATIF & 2) && in_tile && t < tw - 1 && r < th - 1 && i <= g.ie + 1 && j <= g.je + 1;
  const bool st1 = own && i >= g.is, st2 = own && j >= g.js;            // stores the x1 / x2 face on its low side
  const bool act = own && i >= g.is && i <= g.ie && j >= g.js && j <= g.je;
  const double dx1 = g.dx[3*m], dx2 = g.dx[3*m + 1], dx3 = g.dx[3*m + 2];
  const bool p2 = AKMI_POW2DX && is_pow2(dx1) && is_pow2(dx2) && is_pow2(dx3);      // wave-uniform
  const int n1 = pow2_shift(dx1), n2 = pow2_shift(dx2), n3 = pow2_shift(dx3);
  const double bdt = to_sgpr(beta_dt_of(u.beta_dt, u.dtp));
  const size_t cs = (size_t)g.N3*g.N2*g.N1, ps = (size_t)g.N2*g.N1;
  const size_t ps1 = (size_t)g.N2*(g.N1 + 1), ps2 = (size_t)(g.N2 + 1)*g.N1;
  const double *wb = a.w0 + (size_t)m*g.nvar*cs;
  const double *bb = a.bcc0 + (size_t)m*3*cs;
  auto base = [&](int n) -> const double * { return n < 5 ? wb + n*cs : bb + (n - 5)*cs; };
  // halo entry of this thread: rows 0,1 and ph-1 in full, columns 0,1 and pw-1 of the tile rows
  int hy = -1, hx = 0;
  {
    const int nh = 3*pw + 3*th;
    if (tid < nh) {
      if (tid < 3*pw) { const int q = tid/pw; hy = q < 2 ? q : ph - 1; hx = tid - q*pw; }
      else { const int q = tid - 3*pw; const int rr = q/3, cc = q - rr*3; hy = 2 + rr; hx = cc < 2 ? cc : pw - 1; }
    }
  }
  const int hj = j0 - 2 + hy, hi = i0 - 2 + hx;
  const bool hload = hy >= 0 && hj >= 0 && hj < g.N2 && hi >= 0 && hi < g.N1;      // the halo cell exists in memory
  // the cells just outside the tile's low sides have no position of their own: column 1 of the plane (rows of the
  // tile) and row 1 (columns of the tile) are reconstructed by the first th + tw threads, one cell each, in the one
  // direction in which a face of the tile needs them
  int ha = -1, hs = 0, hd = 0, hdir = 0;         // entry below the cell, stride to the cell / the entry above, destination
  if (tid < th) { ha = (tid + 2)*pw*PE; hs = PE; hd = 2*PLSZ + tid*tw*FE; hdir = 1; }
  else if (tid < th + tw) { const int c = tid - th; ha = (c + 2)*PE; hs = PE*pw; hd = 2*PLSZ + FE*fn + c*FE; hdir = 2; }
  // own entries: cell (r+2, t+2) of a plane, position (r, t) of the two face arrays
  const int qo = in_tile ? ((r + 2)*pw + t + 2)*PE : 0, qy = PE*pw;
  const int xo = in_tile ? 2*PLSZ + (r*tw + t)*FE : 2*PLSZ, x2o = xo + FE*fn, xy = FE*tw;
  const int hq = hy >= 0 ? (hy*pw + hx)*PE : -1;
  // Addresses: scalar base of (block, array) + a 32-bit byte offset per lane and array shape, advanced by one plane per
  // step.  Every load is unconditional (lanes without a cell read element 0 of the plane, planes are clamped into the
  // array): a load in a branch makes the wait for it a wait for everything (s_waitcnt vmcnt(0) at the join).
  unsigned oc = (cell_ok ? ((unsigned)j*(unsigned)g.N1 + (unsigned)i)*8u : 0u) + (unsigned)(k0 - 1)*(unsigned)ps*8u;               // (.., N2, N1), plane k-1
  unsigned o1 = (cell_ok ? ((unsigned)j*(unsigned)(g.N1 + 1) + (unsigned)i)*8u : 0u) + (unsigned)(k0 - 1)*(unsigned)ps1*8u;        // (.., N2, N1+1)
  unsigned o2 = (cell_ok ? ((unsigned)j*(unsigned)g.N1 + (unsigned)i)*8u : 0u) + (unsigned)(k0 - 1)*(unsigned)ps2*8u;              // (.., N2+1, N1)
  unsigned oh = (hload ? ((unsigned)hj*(unsigned)g.N1 + (unsigned)hi)*8u : 0u) + (unsigned)(k0 - 1)*(unsigned)ps*8u;               // halo cell, plane k-1
  const size_t mb = (size_t)m*g.nvar*cs;
  const double *b1m = a.bx1f + (size_t)m*g.N3*ps1, *b2m = a.bx2f + (size_t)m*g.N3*ps2, *b3m = a.bx3f + (size_t)m*(g.N3 + 1)*ps;
  double *mf1 = a.mf1 + (size_t)m*g.nvar*g.N3*ps1, *mf2 = a.mf2 + (size_t)m*g.nvar*g.N3*ps2,
         *mf3 = a.mf3 + (size_t)m*g.nvar*(g.N3 + 1)*ps;
  const size_t mc = (size_t)m*cs;
  double *u0m = u.u0 + mb, *u1m = u.u1 + mb;
  // natural variable order of a plane entry: d, vx, vy, vz, e, bx, by, bz.  Plane kk lives in slot (kk - k0 + 1) & 1.
  double PL[8], F3p[5];
  {
    double qa[8], qb[8], qc[8], qh[8];
    const long sa = k0 >= 2 ? -(long)ps : 0;               // plane k0-2 (k0-1 where it does not exist: the value is not used)
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const double *q = base(n);
      qa[n] = ldu(q + sa, oc); qb[n] = ldu(q, oc); qc[n] = ldu(q + ps, oc); qh[n] = ldu(q + ps, oh);
    }
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      double dummy;
      plm(qa[n], qb[n], qc[n], PL[n], dummy);
      if (in_tile) { ms_lds[qo + n] = qb[n]; ms_lds[PLSZ + qo + n] = qc[n]; }      // own entries of planes k0-1 and k0
      if (hq >= 0) ms_lds[PLSZ + hq + n] = qh[n];
    }
  }
#pragma unroll
  for (int n = 0; n < 5; ++n) F3p[n] = 0.0;
  double bn1 = 0.0, bn2 = 0.0, bn3 = ldu(b3m + ps, oc);       // normal fields of the faces of the coming step
  // step k: x3 face k (below cell k); for k > k0 also the x1/x2 faces of plane k-1, which finishes cell k-1;
  // plane k+1 (loaded during the step) replaces plane k-1
  for (int k = k0; k <= k1 + 1; ++k) {
    const bool plane = k > k0;                        // workgroup-uniform
    const bool do3 = k >= g.ks && k <= g.ke + 1;      // the x3 face of this step exists
    const long s2 = (k + 1 < g.N3) ? 2*(long)ps : (long)ps;            // plane k+1 relative to plane k-1 (clamped into the array)
    const bool upd = plane && act && k - 1 >= g.ks && k - 1 <= g.ke;
    const int pP = ((k - k0) & 1) ? PLSZ : 0, pC = PLSZ - pP;          // slots of plane k-1 and of plane k
    const int qP = qo + pP, qC = qo + pC;
    double f1[5] = {0, 0, 0, 0, 0}, f2[5] = {0, 0, 0, 0, 0};      // this position's own in-plane fluxes (d, m1, m2, m3, E)
    double wp[8], hv[8];
    double e2by = 0.0, e2bz = 0.0;
    // the face fields of the NEXT step ride with the loads of plane k+1: the first wait of a step then comes a whole
    // solve after the last store (a wait with stores outstanding is a wait for them as well)
    double nb1, nb2, nb3;
    if (!plane) {
#pragma unroll
      for (int n = 0; n < 8; ++n) { wp[n] = ldu(base(n) + s2, oc); hv[n] = ldu(base(n) + s2, oh); }
      nb1 = ldu(b1m + ps1, o1); nb2 = ldu(b2m + ps2, o2); nb3 = ldu(b3m + 2*ps, oc);
    } else {
      // (A) every cell of plane k-1 once per direction
      double R1[8], R2[8];
#pragma unroll
      for (int n = 0; n < 8; ++n) { R1[n] = 0.0; R2[n] = 0.0; }
      if (in_tile) {
        // the cell and its four in-plane neighbours, all requested before the first limiter (one LDS round trip)
        double W0[8], Wl[8], Wr[8], Wd[8], Wu[8], U1[8], U2[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          W0[n] = ms_lds[qP + n];
          Wl[n] = ms_lds[qP - PE + n]; Wr[n] = ms_lds[qP + PE + n];
          Wd[n] = ms_lds[qP - qy + n]; Wu[n] = ms_lds[qP + qy + n];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          if (n != 5) plm(Wl[n], W0[n], Wr[n], U1[n], R1[n]);
          if (n != 6) plm(Wd[n], W0[n], Wu[n], U2[n], R2[n]);
        }
        if (t + 1 < tw) {            // x1 face entry: d, vx, vy, vz, e, by, bz
#pragma unroll
          for (int n = 0; n < 5; ++n) ms_lds[xo + FE + n] = U1[n];
          ms_lds[xo + FE + 5] = U1[6]; ms_lds[xo + FE + 6] = U1[7];
        }
        if (r + 1 < th) {            // x2 face entry: d, vx, vy, vz, e, bx, bz
#pragma unroll
          for (int n = 0; n < 5; ++n) ms_lds[x2o + xy + n] = U2[n];
          ms_lds[x2o + xy + 5] = U2[5]; ms_lds[x2o + xy + 6] = U2[7];
        }
        if (own) {                   // cell-centred E = -(v x B) of cell (k-1, j, i)  (mhd_corner_e.cpp:309-336)
          MS_STU(a.ecc1 + mc, oc, W0[3]*W0[6] - W0[2]*W0[7]);
          MS_STU(a.ecc2 + mc, oc, W0[1]*W0[7] - W0[3]*W0[5]);
          MS_STU(a.ecc3 + mc, oc, W0[2]*W0[5] - W0[1]*W0[6]);
        }
      }
      if (ha >= 0 && !(AKMI_MS_WHATIF & 4)) {
        const int hA = ha + pP;
        // d, vx, vy, vz, e, bz and the transverse field that is reconstructed (by for x1, bx for x2: slot 5 of the entry
        // either way); all three cells requested before the first limiter
        const int nt = hdir == 1 ? 6 : 5;
        double ea[7], eb[7], ec[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) {
          const int n = q < 5 ? q : (q == 5 ? nt : 7);
          ea[q] = ms_lds[hA + n]; eb[q] = ms_lds[hA + hs + n]; ec[q] = ms_lds[hA + 2*hs + n];
        }
#pragma unroll
        for (int q = 0; q < 7; ++q) {
          double up, dummy;
          plm(ea[q], eb[q], ec[q], up, dummy);
          ms_lds[hd + q] = up;
        }
      }
      MS_SYNC();
      // (B) the two in-plane faces of this position; the fluid flux takes the place of the left state
      if (in_tile) {   // x1: (d, vx, vy, vz, e, by, bz), normal field bx1f
        const Cons1D f = riemann_mhd_e<RS, EO, FM>(eos, ms_lds[xo], ms_lds[xo + 1], ms_lds[xo + 2], ms_lds[xo + 3],
            ms_lds[xo + 4], ms_lds[xo + 5], ms_lds[xo + 6], R1[0], R1[1], R1[2], R1[3], R1[4], R1[6], R1[7], bn1);
        f1[0] = f.d; f1[1] = f.mx; f1[2] = f.my; f1[3] = f.mz; f1[4] = f.e;
#pragma unroll
        for (int n = 0; n < 5; ++n) ms_lds[xo + n] = f1[n];
        if (st1) {
          MS_STU(mf1, o1, f.d);
          MS_STU(a.e3x1 + mc, oc, -f.by);
          MS_STU(a.e2x1 + mc, oc, f.bz);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // cell k+1 of the column and of the thread's halo entry: in flight during the x2 solve
#pragma unroll
      for (int n = 0; n < 8; ++n) { wp[n] = ldu(base(n) + s2, oc); hv[n] = ldu(base(n) + s2, oh); }
      nb1 = ldu(b1m + ps1, o1); nb2 = ldu(b2m + ps2, o2); nb3 = ldu(b3m + 2*ps, oc);
      __builtin_amdgcn_sched_barrier(0);
      if (in_tile) {   // x2: (d, vy, vz, vx, e, bz, bx), normal field bx2f
        const Cons1D f = riemann_mhd_e<RS, EO, FM>(eos, ms_lds[x2o], ms_lds[x2o + 2], ms_lds[x2o + 3], ms_lds[x2o + 1],
            ms_lds[x2o + 4], ms_lds[x2o + 6], ms_lds[x2o + 5], R2[0], R2[2], R2[3], R2[1], R2[4], R2[7], R2[5], bn2);
        f2[0] = f.d; f2[2] = f.mx; f2[3] = f.my; f2[1] = f.mz; f2[4] = f.e;
#pragma unroll
        for (int n = 0; n < 5; ++n) ms_lds[x2o + n] = f2[n];
        e2by = f.by; e2bz = f.bz;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // x3 face below cell k: (d, vz, vx, vy, e, bx, by), normal field bx3f.  Cells k-1 and k are the position's entries of
    // the two planes; plane k+1 then takes the place of plane k-1 (every reader of it is past (A))
    double L[8], R[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      if (n == 7) { L[n] = R[n] = 0.0; continue; }
      double qln;
      L[n] = PL[n];
      plm(ms_lds[qP + n], ms_lds[qC + n], wp[n], qln, R[n]);
      PL[n] = qln;
    }
    // (the position's own entry also on the last steps, whose x3 face reads it; the halo is not read again then)
    if (in_tile) {
#pragma unroll
      for (int n = 0; n < 8; ++n) ms_lds[qP + n] = wp[n];
    }
    if (hq >= 0) {
#pragma unroll
      for (int n = 0; n < 8; ++n) ms_lds[hq + pP + n] = hv[n];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (plane && st2) {              // the x2 face's stores, after the wait for plane k+1 (a store between a load and its
      MS_STU(mf2, o2, f2[0]);           // wait makes the wait one for the store as well)
      MS_STU(a.e1x2 + mc, oc, -e2by);
      MS_STU(a.e3x2 + mc, oc, e2bz);
    }
    __builtin_amdgcn_sched_barrier(0);
    double pu0[5], pu1[5];
    // operands of the update: in flight during the x3 solve
#pragma unroll
    for (int n = 0; n < 5; ++n) {
      pu0[n] = ldu(u0m + n*cs, oc);
      pu1[n] = COPY ? 0.0 : ldu(u1m + n*cs, oc);
    }
    __builtin_amdgcn_sched_barrier(0);
    double f3[5] = {0, 0, 0, 0, 0};
    double e3by = 0.0, e3bz = 0.0;
    if (do3 && !(AKMI_MS_WHATIF & 8)) {
      const Cons1D f = riemann_mhd_e<RS, EO, FM>(eos, L[0], L[3], L[1], L[2], L[4], L[5], L[6], R[0], R[3], R[1], R[2],
                                                 R[4], R[5], R[6], bn3);
      f3[0] = f.d; f3[3] = f.mx; f3[1] = f.my; f3[2] = f.mz; f3[4] = f.e;
      e3by = f.by; e3bz = f.bz;
    }
    MS_SYNC();
    if (upd) {                                         // (C) finish cell k-1
      double divf[5];
      if (p2) {                                        // one wave-uniform branch for the fifteen quotients
#pragma unroll
        for (int n = 0; n < 5; ++n) {
          divf[n] = ldexp(ms_lds[xo + FE + n] - f1[n], n1);
          divf[n] += ldexp(ms_lds[x2o + xy + n] - f2[n], n2);
          divf[n] += ldexp(f3[n] - F3p[n], n3);
        }
      } else {
#pragma unroll
        for (int n = 0; n < 5; ++n) {
          divf[n] = (ms_lds[xo + FE + n] - f1[n])/dx1;
          divf[n] += (ms_lds[x2o + xy + n] - f2[n])/dx2;
          divf[n] += (f3[n] - F3p[n])/dx3;
        }
      }
#pragma unroll
      for (int n = 0; n < 5; ++n) {
        const double u0v = pu0[n];
        const double u1v = COPY ? u0v : pu1[n];
        if (!(AKMI_MS_WHATIF & 16) || u.copy_u1 == 77) rk_store_u(u0m + n*cs, u1m + n*cs, u.copy_u1, oc, u0v, u.gam0*u0v + u.gam1*u1v - bdt*divf[n]);
      }
    }
    if (do3 && own && k <= k1) {
      MS_STU(mf3 + ps, oc, f3[0]);
      MS_STU(a.e2x3 + mc + ps, oc, -e3by);
      MS_STU(a.e1x3 + mc + ps, oc, e3bz);
    }
#pragma unroll
    for (int n = 0; n < 5; ++n) F3p[n] = f3[n];
    bn1 = nb1; bn2 = nb2; bn3 = nb3;
    oc += (unsigned)ps*8u; o1 += (unsigned)ps1*8u; o2 += (unsigned)ps2*8u; oh += (unsigned)ps*8u;
  }
}

static int launch_mhd_stage3d(const Geo &g, const Scheme &sc, const MhdStageArgs &a, const UpdArgs &u, hipStream_t st) {
  const MhdTile tl = mhd_tile(g.nx1 + 2, g.nx2 + 2);
  if (tl.tw == 0) { set_error("mhd_stage3d: no tile shape"); return AKMI_FAIL; }
  const int nplanes = g.nx3 + 2;
  int ckl = march_len((long)tl.n1*tl.n2, nplanes, g.nmb, ML);
  static const int ckl_env = getenv("AKMI_MS_CKL") ? atoi(getenv("AKMI_MS_CKL")) : 0;     // experiments: pin the chunk length
  if (ckl_env > 0) ckl = ckl_env;
  const int nchunk = cdiv(nplanes, ckl);
  const size_t lds = mhd_lds_doubles(tl.tw, tl.th)*sizeof(double);
  dim3 grid(tl.n1, tl.n2, nchunk*g.nmb), block(tl.threads);
  static size_t granted[2] = {64*1024, 64*1024};
  const int cv = u.copy_u1 ? 1 : 0;
  auto kern = cv ? k_mhd_stage3d<3, true> : k_mhd_stage3d<3, false>;
  if (lds > granted[cv]) {
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      set_error("mhd_stage3d: %zu bytes of LDS refused", lds);
      return AKMI_FAIL;
    }
    granted[cv] = lds;
  }
  kern<<<grid, block, lds, st>>>(g, sc.eos, a, u, nchunk, ckl, tl.tw, tl.th);
  AKMI_CHECK_LAUNCH("mhd_stage3d");
  return AKMI_COMPLETE;
}
