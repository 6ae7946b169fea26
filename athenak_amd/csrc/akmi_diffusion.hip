// akmi_diffusion.hip -- diffusion hooks of the task chain: constant isotropic
// viscosity, constant thermal diffusivity and Ohmic resistivity as adders to the face fluxes and
// the corner EMFs, exactly where the reference calls them:
//   Hydro/MHD::Fluxes   -> Conduction::AddHeatFluxes, Viscosity::AddViscousFluxes,
//                          Resistivity::AddResistiveFluxes (hydro_tasks.cpp:183-189, mhd_tasks.cpp:198-206)
//   MHD::EField         -> Resistivity::AddResistiveEMFs   (mhd_tasks.cpp:381-383)
// One thread per face (edge), the arithmetic of each expression kept in the reference's order.
// All of them are streaming kernels: a face reads its two (to eighteen) neighbouring cells.
#include "akmi_common.hpp"
#include <cfloat>

namespace akmi {

constexpr int DX = 64, DY = 4;

struct Wv {                      // primitive array access  w(n,k,j,i) of MeshBlock m
  const double *w;
  int N3, N2, N1;
  size_t cs;
  __device__ double operator()(int n, int k, int j, int i) const {
    return w[n*cs + ((size_t)k*N2 + j)*N1 + i];
  }
};

__device__ __forceinline__ Wv wview(const Geo &g, const double *w0, int m) {
  const size_t cs = (size_t)g.N3*g.N2*g.N1;
  return Wv{w0 + (size_t)m*g.nvar*cs, g.N3, g.N2, g.N1, cs};
}

// Viscosity::AddViscousFluxIso, src/diffusion/viscosity.cpp:64-229
template <int DIR>
__global__ void __launch_bounds__(DX*DY)
k_visc_flux(Geo g, double nu_iso, int ideal, const double *__restrict__ w0, double *__restrict__ flx,
            int f3, int f2, int f1, int nk) {
  const int i = g.is + blockIdx.x*DX + threadIdx.x;
  const int j = g.js + blockIdx.y*DY + threadIdx.y;
  const int m = blockIdx.z/nk;
  const int k = g.ks + (blockIdx.z - m*nk);
  if (i > g.ie + (DIR == 0) || j > g.je + (DIR == 1)) return;
  const double dx1 = g.dx[3*m], dx2 = g.dx[3*m + 1], dx3 = g.dx[3*m + 2];
  const Wv W = wview(g, w0, m);
  constexpr int IDN = 0, IVX = 1, IVY = 2, IVZ = 3, IEN = 4;
  double fvx, fvy, fvz, nud, ax, ay, az;
  if constexpr (DIR == 0) {
    fvx = 4.0*(W(IVX,k,j,i) - W(IVX,k,j,i-1))/(3.0*dx1);
    fvy =     (W(IVY,k,j,i) - W(IVY,k,j,i-1))/dx1;
    fvz =     (W(IVZ,k,j,i) - W(IVZ,k,j,i-1))/dx1;
    if (g.multi_d) {
      fvx -= ((W(IVY,k,j+1,i) + W(IVY,k,j+1,i-1)) - (W(IVY,k,j-1,i) + W(IVY,k,j-1,i-1)))/(6.0*dx2);
      fvy += ((W(IVX,k,j+1,i) + W(IVX,k,j+1,i-1)) - (W(IVX,k,j-1,i) + W(IVX,k,j-1,i-1)))/(4.0*dx2);
    }
    if (g.three_d) {
      fvx -= ((W(IVZ,k+1,j,i) + W(IVZ,k+1,j,i-1)) - (W(IVZ,k-1,j,i) + W(IVZ,k-1,j,i-1)))/(6.0*dx3);
      fvz += ((W(IVX,k+1,j,i) + W(IVX,k+1,j,i-1)) - (W(IVX,k-1,j,i) + W(IVX,k-1,j,i-1)))/(4.0*dx3);
    }
    nud = 0.5*nu_iso*(W(IDN,k,j,i) + W(IDN,k,j,i-1));
    ax = W(IVX,k,j,i-1) + W(IVX,k,j,i); ay = W(IVY,k,j,i-1) + W(IVY,k,j,i);
    az = W(IVZ,k,j,i-1) + W(IVZ,k,j,i);
  } else if constexpr (DIR == 1) {
    fvx = (W(IVX,k,j,i) - W(IVX,k,j-1,i))/dx2 +
          ((W(IVY,k,j,i+1) + W(IVY,k,j-1,i+1)) - (W(IVY,k,j,i-1) + W(IVY,k,j-1,i-1)))/(4.0*dx1);
    fvy = (W(IVY,k,j,i) - W(IVY,k,j-1,i))*4.0/(3.0*dx2) -
          ((W(IVX,k,j,i+1) + W(IVX,k,j-1,i+1)) - (W(IVX,k,j,i-1) + W(IVX,k,j-1,i-1)))/(6.0*dx1);
    fvz = (W(IVZ,k,j,i) - W(IVZ,k,j-1,i))/dx2;
    if (g.three_d) {
      fvy -= ((W(IVZ,k+1,j,i) + W(IVZ,k+1,j-1,i)) - (W(IVZ,k-1,j,i) + W(IVZ,k-1,j-1,i)))/(6.0*dx3);
      fvz += ((W(IVY,k+1,j,i) + W(IVY,k+1,j-1,i)) - (W(IVY,k-1,j,i) + W(IVY,k-1,j-1,i)))/(4.0*dx3);
    }
    nud = 0.5*nu_iso*(W(IDN,k,j,i) + W(IDN,k,j-1,i));
    ax = W(IVX,k,j-1,i) + W(IVX,k,j,i); ay = W(IVY,k,j-1,i) + W(IVY,k,j,i);
    az = W(IVZ,k,j-1,i) + W(IVZ,k,j,i);
  } else {
    fvx = (W(IVX,k,j,i) - W(IVX,k-1,j,i))/dx3 +
          ((W(IVZ,k,j,i+1) + W(IVZ,k-1,j,i+1)) - (W(IVZ,k,j,i-1) + W(IVZ,k-1,j,i-1)))/(4.0*dx1);
    fvy = (W(IVY,k,j,i) - W(IVY,k-1,j,i))/dx3 +
          ((W(IVZ,k,j+1,i) + W(IVZ,k-1,j+1,i)) - (W(IVZ,k,j-1,i) + W(IVZ,k-1,j-1,i)))/(4.0*dx2);
    fvz = (W(IVZ,k,j,i) - W(IVZ,k-1,j,i))*4.0/(3.0*dx3) -
          ((W(IVX,k,j,i+1) + W(IVX,k-1,j,i+1)) - (W(IVX,k,j,i-1) + W(IVX,k-1,j,i-1)))/(6.0*dx1) -
          ((W(IVY,k,j+1,i) + W(IVY,k-1,j+1,i)) - (W(IVY,k,j-1,i) + W(IVY,k-1,j-1,i)))/(6.0*dx2);
    nud = 0.5*nu_iso*(W(IDN,k,j,i) + W(IDN,k-1,j,i));
    ax = W(IVX,k-1,j,i) + W(IVX,k,j,i); ay = W(IVY,k-1,j,i) + W(IVY,k,j,i);
    az = W(IVZ,k-1,j,i) + W(IVZ,k,j,i);
  }
  const size_t fs = (size_t)f3*f2*f1;
  double *f = flx + ix5(g.nvar, f3, f2, f1, m, 0, k, j, i);
  f[IVX*fs] -= nud*fvx;
  f[IVY*fs] -= nud*fvy;
  f[IVZ*fs] -= nud*fvz;
  if (ideal) f[IEN*fs] -= 0.5*nud*(ax*fvx + ay*fvy + az*fvz);
}

// Conduction::AddHeatFluxIso, src/diffusion/conduction.cpp:106-152
template <int DIR>
__global__ void __launch_bounds__(DX*DY)
k_heat_flux(Geo g, double alpha_iso, double gm1, const double *__restrict__ w0,
            double *__restrict__ flx, int f3, int f2, int f1, int nk) {
  const int i = g.is + blockIdx.x*DX + threadIdx.x;
  const int j = g.js + blockIdx.y*DY + threadIdx.y;
  const int m = blockIdx.z/nk;
  const int k = g.ks + (blockIdx.z - m*nk);
  if (i > g.ie + (DIR == 0) || j > g.je + (DIR == 1)) return;
  const Wv W = wview(g, w0, m);
  const int kl = k - (DIR == 2), jl = j - (DIR == 1), il = i - (DIR == 0);
  const double dx = g.dx[3*m + DIR];
  const double tempr = W(4,k,j,i)/W(0,k,j,i);
  const double templ = W(4,kl,jl,il)/W(0,kl,jl,il);
  const double dtempdx = (tempr - templ) * gm1 / dx;
  const double densf = 0.5*(W(0,k,j,i) + W(0,kl,jl,il));
  flx[ix5(g.nvar, f3, f2, f1, m, 4, k, j, i)] -= alpha_iso * densf * dtempdx;
}

// Conduction::NewTimeStep, conduction.cpp:314-377: min over the active cells of
// SQR(dx)/alpha*d/gm1 per direction.  Every factor is positive, so the expression is a monotone
// function of d for a given block and the minimum over a block is the value at its minimum
// density: reduce min(d) per wave, evaluate once per workgroup, one atomicMin if it can win.
__global__ void __launch_bounds__(DX*DY)
k_cond_newdt(Geo g, double alpha_iso, double gm1, const double *__restrict__ w0,
             double *__restrict__ dtmin, int nk) {
  const int i = g.is + blockIdx.x*DX + threadIdx.x;
  const int j = g.js + blockIdx.y*DY + threadIdx.y;
  const int m = blockIdx.z/nk;
  const int k = g.ks + (blockIdx.z - m*nk);
  double d = DBL_MAX;
  if (i <= g.ie && j <= g.je) d = w0[ix5(g.nvar, g.N3, g.N2, g.N1, m, 0, k, j, i)];
  for (int off = 32; off > 0; off >>= 1) d = fmin(d, __shfl_xor(d, off, 64));
  __shared__ double sm[DY];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.y] = d;
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) {
    for (int q = 1; q < DY; ++q) d = fmin(d, sm[q]);
    if (d == DBL_MAX) return;
    double v = sqr(g.dx[3*m])/alpha_iso*d/gm1;
    if (g.multi_d) v = fmin(v, sqr(g.dx[3*m + 1])/alpha_iso*d/gm1);
    if (g.three_d) v = fmin(v, sqr(g.dx[3*m + 2])/alpha_iso*d/gm1);
    if (v < __hip_atomic_load(dtmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMin(reinterpret_cast<unsigned long long *>(dtmin),
                (unsigned long long)__double_as_longlong(v));
  }
}

__global__ void k_set_fltmax(double *x) { if (threadIdx.x == 0) *x = (double)FLT_MAX; }

struct Bf {                      // face-field access of MeshBlock m
  const double *b1, *b2, *b3;
  int N3, N2, N1;
  __device__ double x1f(int k, int j, int i) const { return b1[((size_t)k*N2 + j)*(N1 + 1) + i]; }
  __device__ double x2f(int k, int j, int i) const { return b2[((size_t)k*(N2 + 1) + j)*N1 + i]; }
  __device__ double x3f(int k, int j, int i) const { return b3[((size_t)k*N2 + j)*N1 + i]; }
};
__device__ __forceinline__ Bf bview(const Geo &g, const double *b1, const double *b2,
                                    const double *b3, int m) {
  return Bf{b1 + (size_t)m*g.N3*g.N2*(g.N1 + 1), b2 + (size_t)m*g.N3*(g.N2 + 1)*g.N1,
            b3 + (size_t)m*(g.N3 + 1)*g.N2*g.N1, g.N3, g.N2, g.N1};
}

// Resistivity::AddEMFConstantResist (resistivity.cpp:78-177) with CurrentDensity
// (current_density.hpp:30-57): E += eta_ohm*J at the edges [is,ie+1] x [js,je+1] x [ks,ke+1]
// (1-D / 2-D: the collapsed directions receive the same value on both of their planes)
__global__ void __launch_bounds__(DX*DY)
k_resist_emf(Geo g, double eta, const double *__restrict__ bx1f, const double *__restrict__ bx2f,
             const double *__restrict__ bx3f, double *__restrict__ e1, double *__restrict__ e2,
             double *__restrict__ e3, int nk) {
  const int i = g.is + blockIdx.x*DX + threadIdx.x;
  const int j = g.js + blockIdx.y*DY + threadIdx.y;
  const int m = blockIdx.z/nk;
  const int k = g.ks + (blockIdx.z - m*nk);
  if (i > g.ie + 1 || j > (g.multi_d ? g.je + 1 : g.js)) return;
  const Bf b = bview(g, bx1f, bx2f, bx3f, m);
  const double dx1 = g.dx[3*m], dx2 = g.dx[3*m + 1], dx3 = g.dx[3*m + 2];
  double j1 = 0.0;
  double j2 = -(b.x3f(k,j,i) - b.x3f(k,j,i-1))/dx1;
  double j3 =  (b.x2f(k,j,i) - b.x2f(k,j,i-1))/dx1;
  if (g.multi_d) {
    j1 += (b.x3f(k,j,i) - b.x3f(k,j-1,i))/dx2;
    j3 -= (b.x1f(k,j,i) - b.x1f(k,j-1,i))/dx2;
  }
  if (g.three_d) {
    j1 -= (b.x2f(k,j,i) - b.x2f(k-1,j,i))/dx3;
    j2 += (b.x1f(k,j,i) - b.x1f(k-1,j,i))/dx3;
  }
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3;
  if (g.three_d) {
    e1[ix4(N3 + 1, N2 + 1, N1, m, k, j, i)] += eta*j1;
    e2[ix4(N3 + 1, N2, N1 + 1, m, k, j, i)] += eta*j2;
    e3[ix4(N3, N2 + 1, N1 + 1, m, k, j, i)] += eta*j3;
  } else if (g.multi_d) {
    e1[ix4(N3 + 1, N2 + 1, N1, m, g.ks, j, i)] += eta*j1;
    e1[ix4(N3 + 1, N2 + 1, N1, m, g.ke + 1, j, i)] += eta*j1;
    e2[ix4(N3 + 1, N2, N1 + 1, m, g.ks, j, i)] += eta*j2;
    e2[ix4(N3 + 1, N2, N1 + 1, m, g.ke + 1, j, i)] += eta*j2;
    e3[ix4(N3, N2 + 1, N1 + 1, m, g.ks, j, i)] += eta*j3;
  } else {
    e2[ix4(N3 + 1, N2, N1 + 1, m, g.ks, g.js, i)] += eta*j2;
    e2[ix4(N3 + 1, N2, N1 + 1, m, g.ke + 1, g.js, i)] += eta*j2;
    e3[ix4(N3, N2 + 1, N1 + 1, m, g.ks, g.js, i)] += eta*j3;
    e3[ix4(N3, N2 + 1, N1 + 1, m, g.ks, g.je + 1, i)] += eta*j3;
  }
}

// Resistivity::AddFluxConstantResist, resistivity.cpp:185-272 (energy flux, face-shaped arrays)
template <int DIR>
__global__ void __launch_bounds__(DX*DY)
k_resist_flux(Geo g, double eta, const double *__restrict__ bx1f, const double *__restrict__ bx2f,
              const double *__restrict__ bx3f, double *__restrict__ flx, int nk) {
  const int i = g.is + blockIdx.x*DX + threadIdx.x;
  const int j = g.js + blockIdx.y*DY + threadIdx.y;
  const int m = blockIdx.z/nk;
  const int k = g.ks + (blockIdx.z - m*nk);
  if (i > g.ie + (DIR == 0) || j > g.je + (DIR == 1)) return;
  const Bf b = bview(g, bx1f, bx2f, bx3f, m);
  const double dx1 = g.dx[3*m], dx2 = g.dx[3*m + 1], dx3 = g.dx[3*m + 2];
  const double qa = 0.25*eta;
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3;
  if constexpr (DIR == 0) {
    double j2k   = -(b.x3f(k  ,j,i) - b.x3f(k  ,j,i-1))/dx1;
    double j2kp1 = -(b.x3f(k+1,j,i) - b.x3f(k+1,j,i-1))/dx1;
    double j3j   = (b.x2f(k,j  ,i) - b.x2f(k,j  ,i-1))/dx1;
    double j3jp1 = (b.x2f(k,j+1,i) - b.x2f(k,j+1,i-1))/dx1;
    if (g.multi_d) {
      j3j   -= (b.x1f(k,j  ,i) - b.x1f(k,j-1,i))/dx2;
      j3jp1 -= (b.x1f(k,j+1,i) - b.x1f(k,j  ,i))/dx2;
    }
    if (g.three_d) {
      j2k   += (b.x1f(k  ,j,i) - b.x1f(k-1,j,i))/dx3;
      j2kp1 += (b.x1f(k+1,j,i) - b.x1f(k  ,j,i))/dx3;
    }
    flx[ix5(g.nvar, N3, N2, N1 + 1, m, 4, k, j, i)] +=
        qa*(j2k  *(b.x3f(k  ,j  ,i) + b.x3f(k  ,j  ,i-1)) +
            j2kp1*(b.x3f(k+1,j  ,i) + b.x3f(k+1,j  ,i-1)) -
            j3j  *(b.x2f(k  ,j  ,i) + b.x2f(k  ,j  ,i-1)) -
            j3jp1*(b.x2f(k  ,j+1,i) + b.x2f(k  ,j+1,i-1)));
  } else if constexpr (DIR == 1) {
    double j1k   = (b.x3f(k  ,j,i) - b.x3f(k  ,j-1,i))/dx2;
    double j1kp1 = (b.x3f(k+1,j,i) - b.x3f(k+1,j-1,i))/dx2;
    double j3i   = (b.x2f(k,j,i  ) - b.x2f(k,j  ,i-1))/dx1 - (b.x1f(k,j,i  ) - b.x1f(k,j-1,i  ))/dx2;
    double j3ip1 = (b.x2f(k,j,i+1) - b.x2f(k,j  ,i  ))/dx1 - (b.x1f(k,j,i+1) - b.x1f(k,j-1,i+1))/dx2;
    if (g.three_d) {
      j1k   -= (b.x2f(k  ,j,i) - b.x2f(k-1,j,i))/dx3;
      j1kp1 -= (b.x2f(k+1,j,i) - b.x2f(k  ,j,i))/dx3;
    }
    flx[ix5(g.nvar, N3, N2 + 1, N1, m, 4, k, j, i)] +=
        qa*(j3i  *(b.x1f(k  ,j,i  ) + b.x1f(k  ,j-1,i  )) +
            j3ip1*(b.x1f(k  ,j,i+1) + b.x1f(k  ,j-1,i+1)) -
            j1k  *(b.x3f(k  ,j,i  ) + b.x3f(k  ,j-1,i  )) -
            j1kp1*(b.x3f(k+1,j,i  ) + b.x3f(k+1,j-1,i  )));
  } else {
    double j1j   = (b.x3f(k,j  ,i) - b.x3f(k  ,j-1,i))/dx2 - (b.x2f(k,j  ,i) - b.x2f(k-1,j  ,i))/dx3;
    double j1jp1 = (b.x3f(k,j+1,i) - b.x3f(k  ,j  ,i))/dx2 - (b.x2f(k,j+1,i) - b.x2f(k-1,j+1,i))/dx3;
    double j2i   = -(b.x3f(k,j,i  ) - b.x3f(k  ,j,i-1))/dx1 + (b.x1f(k,j,i  ) - b.x1f(k-1,j,i  ))/dx3;
    double j2ip1 = -(b.x3f(k,j,i+1) - b.x3f(k  ,j,i  ))/dx1 + (b.x1f(k,j,i+1) - b.x1f(k-1,j,i+1))/dx3;
    flx[ix5(g.nvar, N3 + 1, N2, N1, m, 4, k, j, i)] +=
        qa*(j1j  *(b.x2f(k,j  ,i  ) + b.x2f(k-1,j  ,i  )) +
            j1jp1*(b.x2f(k,j+1,i  ) + b.x2f(k-1,j+1,i  )) -
            j2i  *(b.x1f(k,j  ,i  ) + b.x1f(k-1,j  ,i  )) -
            j2ip1*(b.x1f(k,j  ,i+1) + b.x1f(k-1,j  ,i+1)));
  }
}

// ---- ambipolar diffusion, constant eta_ad (src/diffusion/ambipolar.cpp) ------------------------------
struct Amb {                     // field access of MeshBlock m + edge currents (EdgeJ1/2/3, :30-58)
  Bf b;
  const double *bcc;             // bcc0 of block m: [3][N3][N2][N1]
  size_t cs;
  double dx1, dx2, dx3;
  bool multi_d, three_d;
  __device__ double cc(int n, int k, int j, int i) const { return bcc[n*cs + ((size_t)k*b.N2 + j)*b.N1 + i]; }
  __device__ double j1(int k, int j, int i) const {
    double v = 0.0;
    if (multi_d) v += (b.x3f(k,j,i) - b.x3f(k,j-1,i))/dx2;
    if (three_d) v -= (b.x2f(k,j,i) - b.x2f(k-1,j,i))/dx3;
    return v;
  }
  __device__ double j2(int k, int j, int i) const {
    double v = -(b.x3f(k,j,i) - b.x3f(k,j,i-1))/dx1;
    if (three_d) v += (b.x1f(k,j,i) - b.x1f(k-1,j,i))/dx3;
    return v;
  }
  __device__ double j3(int k, int j, int i) const {
    double v = (b.x2f(k,j,i) - b.x2f(k,j,i-1))/dx1;
    if (multi_d) v -= (b.x1f(k,j,i) - b.x1f(k,j-1,i))/dx2;
    return v;
  }
};

// Resistivity::AddEMFConstantAmbipolar, ambipolar.cpp:66-246: E += eta*(B^2 J - (J.B) B) on the edges
__global__ void __launch_bounds__(DX*DY)
k_amb_emf(Geo g, double eta, const double *__restrict__ bcc0, const double *__restrict__ bx1f,
          const double *__restrict__ bx2f, const double *__restrict__ bx3f, double *__restrict__ e1,
          double *__restrict__ e2, double *__restrict__ e3, int nk) {
  const int i = g.is + blockIdx.x*DX + threadIdx.x;
  const int j = g.js + blockIdx.y*DY + threadIdx.y;
  const int m = blockIdx.z/nk;
  const int k = g.ks + (blockIdx.z - m*nk);
  if (i > g.ie + 1 || j > (g.multi_d ? g.je + 1 : g.js)) return;
  const size_t cs = (size_t)g.N3*g.N2*g.N1;
  const Amb a{bview(g, bx1f, bx2f, bx3f, m), bcc0 + (size_t)m*3*cs, cs, g.dx[3*m], g.dx[3*m + 1],
              g.dx[3*m + 2], (bool)g.multi_d, (bool)g.three_d};
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3, ks = g.ks, ke = g.ke, js = g.js, je = g.je;
  if (!g.multi_d) {
    const double intBx = a.b.x1f(ks,js,i);
    const double intBy = 0.5*(a.cc(1,ks,js,i) + a.cc(1,ks,js,i-1));
    const double intBz = 0.5*(a.cc(2,ks,js,i) + a.cc(2,ks,js,i-1));
    const double intJ2 = a.j2(ks,js,i), intJ3 = a.j3(ks,js,i);
    const double Bsq = sqr(intBx) + sqr(intBy) + sqr(intBz);
    const double JdotB = intJ2*intBy + intJ3*intBz;
    const double e2_amb = eta * (Bsq*intJ2 - JdotB*intBy);
    const double e3_amb = eta * (Bsq*intJ3 - JdotB*intBz);
    e2[ix4(N3 + 1, N2, N1 + 1, m, ks, js, i)] += e2_amb;
    e2[ix4(N3 + 1, N2, N1 + 1, m, ke + 1, js, i)] += e2_amb;
    e3[ix4(N3, N2 + 1, N1 + 1, m, ks, js, i)] += e3_amb;
    e3[ix4(N3, N2 + 1, N1 + 1, m, ks, je + 1, i)] += e3_amb;
  } else if (!g.three_d) {
    const double intJ1_e1 = a.j1(ks,j,i);
    const double intJ2_e1 = 0.25*(a.j2(ks,j-1,i) + a.j2(ks,j-1,i+1) + a.j2(ks,j,i) + a.j2(ks,j,i+1));
    const double intJ3_e1 = 0.5*(a.j3(ks,j,i) + a.j3(ks,j,i+1));
    const double intBx_e1 = 0.5*(a.cc(0,ks,j,i) + a.cc(0,ks,j-1,i));
    const double intBy_e1 = a.b.x2f(ks,j,i);
    const double intBz_e1 = 0.5*(a.cc(2,ks,j,i) + a.cc(2,ks,j-1,i));
    const double Bsq_e1 = sqr(intBx_e1) + sqr(intBy_e1) + sqr(intBz_e1);
    const double JdotB_e1 = intJ1_e1*intBx_e1 + intJ2_e1*intBy_e1 + intJ3_e1*intBz_e1;
    const double e1_amb = eta * (Bsq_e1*intJ1_e1 - JdotB_e1*intBx_e1);
    e1[ix4(N3 + 1, N2 + 1, N1, m, ks, j, i)] += e1_amb;
    e1[ix4(N3 + 1, N2 + 1, N1, m, ke + 1, j, i)] += e1_amb;
    const double intJ1_e2 = 0.25*(a.j1(ks,j,i-1) + a.j1(ks,j,i) + a.j1(ks,j+1,i-1) + a.j1(ks,j+1,i));
    const double intJ2_e2 = a.j2(ks,j,i);
    const double intJ3_e2 = 0.5*(a.j3(ks,j,i) + a.j3(ks,j+1,i));
    const double intBx_e2 = a.b.x1f(ks,j,i);
    const double intBy_e2 = 0.5*(a.cc(1,ks,j,i) + a.cc(1,ks,j,i-1));
    const double intBz_e2 = 0.5*(a.cc(2,ks,j,i) + a.cc(2,ks,j,i-1));
    const double Bsq_e2 = sqr(intBx_e2) + sqr(intBy_e2) + sqr(intBz_e2);
    const double JdotB_e2 = intJ1_e2*intBx_e2 + intJ2_e2*intBy_e2 + intJ3_e2*intBz_e2;
    const double e2_amb = eta * (Bsq_e2*intJ2_e2 - JdotB_e2*intBy_e2);
    e2[ix4(N3 + 1, N2, N1 + 1, m, ks, j, i)] += e2_amb;
    e2[ix4(N3 + 1, N2, N1 + 1, m, ke + 1, j, i)] += e2_amb;
    const double intJ1_e3 = 0.5*(a.j1(ks,j,i-1) + a.j1(ks,j,i));
    const double intJ2_e3 = 0.5*(a.j2(ks,j-1,i) + a.j2(ks,j,i));
    const double intJ3_e3 = a.j3(ks,j,i);
    const double intBx_e3 = 0.5*(a.b.x1f(ks,j,i) + a.b.x1f(ks,j-1,i));
    const double intBy_e3 = 0.5*(a.b.x2f(ks,j,i) + a.b.x2f(ks,j,i-1));
    const double intBz_e3 = 0.25*(a.cc(2,ks,j,i) + a.cc(2,ks,j-1,i) + a.cc(2,ks,j,i-1) + a.cc(2,ks,j-1,i-1));
    const double Bsq_e3 = sqr(intBx_e3) + sqr(intBy_e3) + sqr(intBz_e3);
    const double JdotB_e3 = intJ1_e3*intBx_e3 + intJ2_e3*intBy_e3 + intJ3_e3*intBz_e3;
    e3[ix4(N3, N2 + 1, N1 + 1, m, ks, j, i)] += eta * (Bsq_e3*intJ3_e3 - JdotB_e3*intBz_e3);
  } else {
    const double intJ1_e1 = a.j1(k,j,i);
    const double intJ2_e1 = 0.25*(a.j2(k,j-1,i) + a.j2(k,j-1,i+1) + a.j2(k,j,i) + a.j2(k,j,i+1));
    const double intJ3_e1 = 0.25*(a.j3(k-1,j,i) + a.j3(k-1,j,i+1) + a.j3(k,j,i) + a.j3(k,j,i+1));
    const double intBx_e1 = 0.25*(a.cc(0,k,j,i) + a.cc(0,k-1,j,i) + a.cc(0,k,j-1,i) + a.cc(0,k-1,j-1,i));
    const double intBy_e1 = 0.5*(a.b.x2f(k,j,i) + a.b.x2f(k-1,j,i));
    const double intBz_e1 = 0.5*(a.b.x3f(k,j,i) + a.b.x3f(k,j-1,i));
    const double Bsq_e1 = sqr(intBx_e1) + sqr(intBy_e1) + sqr(intBz_e1);
    const double JdotB_e1 = intJ1_e1*intBx_e1 + intJ2_e1*intBy_e1 + intJ3_e1*intBz_e1;
    e1[ix4(N3 + 1, N2 + 1, N1, m, k, j, i)] += eta * (Bsq_e1*intJ1_e1 - JdotB_e1*intBx_e1);
    const double intJ1_e2 = 0.25*(a.j1(k,j,i-1) + a.j1(k,j,i) + a.j1(k,j+1,i-1) + a.j1(k,j+1,i));
    const double intJ2_e2 = a.j2(k,j,i);
    const double intJ3_e2 = 0.25*(a.j3(k-1,j,i) + a.j3(k-1,j+1,i) + a.j3(k,j,i) + a.j3(k,j+1,i));
    const double intBx_e2 = 0.5*(a.b.x1f(k,j,i) + a.b.x1f(k-1,j,i));
    const double intBy_e2 = 0.25*(a.cc(1,k,j,i) + a.cc(1,k-1,j,i) + a.cc(1,k,j,i-1) + a.cc(1,k-1,j,i-1));
    const double intBz_e2 = 0.5*(a.b.x3f(k,j,i) + a.b.x3f(k,j,i-1));
    const double Bsq_e2 = sqr(intBx_e2) + sqr(intBy_e2) + sqr(intBz_e2);
    const double JdotB_e2 = intJ1_e2*intBx_e2 + intJ2_e2*intBy_e2 + intJ3_e2*intBz_e2;
    e2[ix4(N3 + 1, N2, N1 + 1, m, k, j, i)] += eta * (Bsq_e2*intJ2_e2 - JdotB_e2*intBy_e2);
    const double intJ1_e3 = 0.25*(a.j1(k,j,i-1) + a.j1(k,j,i) + a.j1(k+1,j,i-1) + a.j1(k+1,j,i));
    const double intJ2_e3 = 0.25*(a.j2(k,j-1,i) + a.j2(k,j,i) + a.j2(k+1,j-1,i) + a.j2(k+1,j,i));
    const double intJ3_e3 = a.j3(k,j,i);
    const double intBx_e3 = 0.5*(a.b.x1f(k,j,i) + a.b.x1f(k,j-1,i));
    const double intBy_e3 = 0.5*(a.b.x2f(k,j,i) + a.b.x2f(k,j,i-1));
    const double intBz_e3 = 0.25*(a.cc(2,k,j,i) + a.cc(2,k,j-1,i) + a.cc(2,k,j,i-1) + a.cc(2,k,j-1,i-1));
    const double Bsq_e3 = sqr(intBx_e3) + sqr(intBy_e3) + sqr(intBz_e3);
    const double JdotB_e3 = intJ1_e3*intBx_e3 + intJ2_e3*intBy_e3 + intJ3_e3*intBz_e3;
    e3[ix4(N3, N2 + 1, N1 + 1, m, k, j, i)] += eta * (Bsq_e3*intJ3_e3 - JdotB_e3*intBz_e3);
  }
}

// eta*B^2*J on an edge with B averaged to it as in the EMF routine (3-D forms, ambipolar.cpp:362-480)
__device__ __forceinline__ double amb_e1(const Amb &a, double eta, int k, int j, int i) {
  const double Bx = 0.25*(a.cc(0,k,j,i) + a.cc(0,k-1,j,i) + a.cc(0,k,j-1,i) + a.cc(0,k-1,j-1,i));
  const double By = 0.5*(a.b.x2f(k,j,i) + a.b.x2f(k-1,j,i));
  const double Bz = 0.5*(a.b.x3f(k,j,i) + a.b.x3f(k,j-1,i));
  return eta * (sqr(Bx) + sqr(By) + sqr(Bz)) * a.j1(k,j,i);
}
__device__ __forceinline__ double amb_e2(const Amb &a, double eta, int k, int j, int i) {
  const double Bx = 0.5*(a.b.x1f(k,j,i) + a.b.x1f(k-1,j,i));
  const double By = 0.25*(a.cc(1,k,j,i) + a.cc(1,k-1,j,i) + a.cc(1,k,j,i-1) + a.cc(1,k-1,j,i-1));
  const double Bz = 0.5*(a.b.x3f(k,j,i) + a.b.x3f(k,j,i-1));
  return eta * (sqr(Bx) + sqr(By) + sqr(Bz)) * a.j2(k,j,i);
}
__device__ __forceinline__ double amb_e3(const Amb &a, double eta, int k, int j, int i) {
  const double Bx = 0.5*(a.b.x1f(k,j,i) + a.b.x1f(k,j-1,i));
  const double By = 0.5*(a.b.x2f(k,j,i) + a.b.x2f(k,j,i-1));
  const double Bz = 0.25*(a.cc(2,k,j,i) + a.cc(2,k,j-1,i) + a.cc(2,k,j,i-1) + a.cc(2,k,j-1,i-1));
  return eta * (sqr(Bx) + sqr(By) + sqr(Bz)) * a.j3(k,j,i);
}

// Resistivity::AddFluxConstantAmbipolar, ambipolar.cpp:254-494 (energy flux, face-shaped arrays)
template <int DIR>
__global__ void __launch_bounds__(DX*DY)
k_amb_flux(Geo g, double eta, const double *__restrict__ bcc0, const double *__restrict__ bx1f,
           const double *__restrict__ bx2f, const double *__restrict__ bx3f, double *__restrict__ flx,
           int nk) {
  const int i = g.is + blockIdx.x*DX + threadIdx.x;
  const int j = g.js + blockIdx.y*DY + threadIdx.y;
  const int m = blockIdx.z/nk;
  const int k = g.ks + (blockIdx.z - m*nk);
  if (i > g.ie + (DIR == 0) || j > g.je + (DIR == 1)) return;
  const size_t cs = (size_t)g.N3*g.N2*g.N1;
  const Amb a{bview(g, bx1f, bx2f, bx3f, m), bcc0 + (size_t)m*3*cs, cs, g.dx[3*m], g.dx[3*m + 1],
              g.dx[3*m + 2], (bool)g.multi_d, (bool)g.three_d};
  const int N1 = g.N1, N2 = g.N2, N3 = g.N3;
  if constexpr (DIR == 0) {
    double *f = flx + ix5(g.nvar, N3, N2, N1 + 1, m, 4, k, j, i);
    if (!g.multi_d) {
      const double Bx = a.b.x1f(k,j,i);
      const double By = 0.5*(a.cc(1,k,j,i-1) + a.cc(1,k,j,i));
      const double Bz = 0.5*(a.cc(2,k,j,i-1) + a.cc(2,k,j,i));
      const double Bsq = sqr(Bx) + sqr(By) + sqr(Bz);
      const double e2_fc = eta * Bsq * a.j2(k,j,i);
      const double e3_fc = eta * Bsq * a.j3(k,j,i);
      *f += e2_fc*Bz - e3_fc*By;
    } else if (!g.three_d) {
      double Bx = a.b.x1f(k,j,i);
      double By = 0.5*(a.cc(1,k,j,i-1) + a.cc(1,k,j,i));
      double Bz = 0.5*(a.cc(2,k,j,i-1) + a.cc(2,k,j,i));
      const double e2_fc = eta * (sqr(Bx) + sqr(By) + sqr(Bz)) * a.j2(k,j,i);
      Bx = 0.5*(a.b.x1f(k,j,i) + a.b.x1f(k,j-1,i));
      By = 0.5*(a.b.x2f(k,j,i) + a.b.x2f(k,j,i-1));
      Bz = 0.25*(a.cc(2,k,j,i) + a.cc(2,k,j-1,i) + a.cc(2,k,j,i-1) + a.cc(2,k,j-1,i-1));
      const double e3_j = eta * (sqr(Bx) + sqr(By) + sqr(Bz)) * a.j3(k,j,i);
      Bx = 0.5*(a.b.x1f(k,j+1,i) + a.b.x1f(k,j,i));
      By = 0.5*(a.b.x2f(k,j+1,i) + a.b.x2f(k,j+1,i-1));
      Bz = 0.25*(a.cc(2,k,j+1,i) + a.cc(2,k,j,i) + a.cc(2,k,j+1,i-1) + a.cc(2,k,j,i-1));
      const double e3_jp1 = eta * (sqr(Bx) + sqr(By) + sqr(Bz)) * a.j3(k,j+1,i);
      const double e3_fc = 0.5*(e3_j + e3_jp1);
      const double b2_fc = 0.5*(a.cc(1,k,j,i-1) + a.cc(1,k,j,i));
      const double b3_fc = 0.5*(a.cc(2,k,j,i-1) + a.cc(2,k,j,i));
      *f += e2_fc*b3_fc - e3_fc*b2_fc;
    } else {
      const double e2_fc = 0.5*(amb_e2(a, eta, k, j, i) + amb_e2(a, eta, k + 1, j, i));
      const double e3_fc = 0.5*(amb_e3(a, eta, k, j, i) + amb_e3(a, eta, k, j + 1, i));
      const double b2_fc = 0.5*(a.cc(1,k,j,i-1) + a.cc(1,k,j,i));
      const double b3_fc = 0.5*(a.cc(2,k,j,i-1) + a.cc(2,k,j,i));
      *f += e2_fc*b3_fc - e3_fc*b2_fc;
    }
  } else if constexpr (DIR == 1) {
    double *f = flx + ix5(g.nvar, N3, N2 + 1, N1, m, 4, k, j, i);
    if (!g.three_d) {
      double Bx = 0.5*(a.b.x1f(k,j,i) + a.b.x1f(k,j-1,i));
      double By = 0.5*(a.b.x2f(k,j,i) + a.b.x2f(k,j,i-1));
      double Bz = 0.25*(a.cc(2,k,j,i) + a.cc(2,k,j-1,i) + a.cc(2,k,j,i-1) + a.cc(2,k,j-1,i-1));
      const double e3_i = eta * (sqr(Bx) + sqr(By) + sqr(Bz)) * a.j3(k,j,i);
      Bx = 0.5*(a.b.x1f(k,j,i+1) + a.b.x1f(k,j-1,i+1));
      By = 0.5*(a.b.x2f(k,j,i+1) + a.b.x2f(k,j,i));
      Bz = 0.25*(a.cc(2,k,j,i+1) + a.cc(2,k,j-1,i+1) + a.cc(2,k,j,i) + a.cc(2,k,j-1,i));
      const double e3_ip1 = eta * (sqr(Bx) + sqr(By) + sqr(Bz)) * a.j3(k,j,i+1);
      const double e3_fc = 0.5*(e3_i + e3_ip1);
      Bx = 0.5*(a.cc(0,k,j,i) + a.cc(0,k,j-1,i));
      By = a.b.x2f(k,j,i);
      Bz = 0.5*(a.cc(2,k,j,i) + a.cc(2,k,j-1,i));
      const double e1_fc = eta * (sqr(Bx) + sqr(By) + sqr(Bz)) * a.j1(k,j,i);
      const double b1_fc = 0.5*(a.cc(0,k,j-1,i) + a.cc(0,k,j,i));
      const double b3_fc = 0.5*(a.cc(2,k,j-1,i) + a.cc(2,k,j,i));
      *f += e3_fc*b1_fc - e1_fc*b3_fc;
    } else {
      const double e3_fc = 0.5*(amb_e3(a, eta, k, j, i) + amb_e3(a, eta, k, j, i + 1));
      const double e1_fc = 0.5*(amb_e1(a, eta, k, j, i) + amb_e1(a, eta, k + 1, j, i));
      const double b1_fc = 0.5*(a.cc(0,k,j-1,i) + a.cc(0,k,j,i));
      const double b3_fc = 0.5*(a.cc(2,k,j-1,i) + a.cc(2,k,j,i));
      *f += e3_fc*b1_fc - e1_fc*b3_fc;
    }
  } else {
    double *f = flx + ix5(g.nvar, N3 + 1, N2, N1, m, 4, k, j, i);
    const double e1_fc = 0.5*(amb_e1(a, eta, k, j, i) + amb_e1(a, eta, k, j + 1, i));
    const double e2_fc = 0.5*(amb_e2(a, eta, k, j, i) + amb_e2(a, eta, k, j, i + 1));
    const double b1_fc = 0.5*(a.cc(0,k-1,j,i) + a.cc(0,k,j,i));
    const double b2_fc = 0.5*(a.cc(1,k-1,j,i) + a.cc(1,k,j,i));
    *f += e1_fc*b2_fc - e2_fc*b1_fc;
  }
}

// Resistivity::NewTimeStep with eta_ad != 0 (resistivity.cpp:313-345): min over the active cells of
// SQR(dx)/(eta_ohm + eta_ad*B^2).  eta is a monotone function of the rounded B^2 sum, SQR(dx)/eta a
// monotone function of eta: reduce max(B^2) per workgroup, evaluate once.
__global__ void __launch_bounds__(DX*DY)
k_resist_newdt(Geo g, double eta_o, double eta_a, const double *__restrict__ bcc0,
               double *__restrict__ dtmin, int nk) {
  const int i = g.is + blockIdx.x*DX + threadIdx.x;
  const int j = g.js + blockIdx.y*DY + threadIdx.y;
  const int m = blockIdx.z/nk;
  const int k = g.ks + (blockIdx.z - m*nk);
  double b2 = -1.0;
  if (i <= g.ie && j <= g.je) {
    const size_t cs = (size_t)g.N3*g.N2*g.N1;
    const size_t c = ix5(3, g.N3, g.N2, g.N1, m, 0, k, j, i);
    b2 = sqr(bcc0[c]) + sqr(bcc0[c + cs]) + sqr(bcc0[c + 2*cs]);
  }
  for (int off = 32; off > 0; off >>= 1) b2 = fmax(b2, __shfl_xor(b2, off, 64));
  __shared__ double sm[DY];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.y] = b2;
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) {
    for (int q = 1; q < DY; ++q) b2 = fmax(b2, sm[q]);
    if (b2 < 0.0) return;
    const double eta = eta_o + eta_a*b2;
    if (!(eta > 0.0)) return;
    double v = sqr(g.dx[3*m])/eta;
    if (g.multi_d) v = fmin(v, sqr(g.dx[3*m + 1])/eta);
    if (g.three_d) v = fmin(v, sqr(g.dx[3*m + 2])/eta);
    if (v < __hip_atomic_load(dtmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMin(reinterpret_cast<unsigned long long *>(dtmin),
                (unsigned long long)__double_as_longlong(v));
  }
}

static dim3 face_grid(const Geo &g, int dir, int &nk) {
  nk = g.ke - g.ks + 1 + (dir == 2);
  return dim3(cdiv(g.nx1 + (dir == 0), DX), cdiv(g.je - g.js + 1 + (dir == 1), DY), nk*g.nmb);
}

}  // namespace akmi

using namespace akmi;

extern "C" {

int akmi_viscous_fluxes(const akmi_pack *p, double nu_iso, const double *w0, double *flx1,
                        double *flx2, double *flx3, int face_shaped, void *stream) {
  Geo g = make_geo(p);
  hipStream_t st = (hipStream_t)stream;
  const int fs = face_shaped ? 1 : 0;
  const dim3 block(DX, DY);
  int nk;
  dim3 g0 = face_grid(g, 0, nk);
  k_visc_flux<0><<<g0, block, 0, st>>>(g, nu_iso, p->is_ideal, w0, flx1, g.N3, g.N2, g.N1 + fs, nk);
  if (g.multi_d) {
    dim3 g1 = face_grid(g, 1, nk);
    k_visc_flux<1><<<g1, block, 0, st>>>(g, nu_iso, p->is_ideal, w0, flx2, g.N3, g.N2 + fs, g.N1, nk);
  }
  if (g.three_d) {
    dim3 g2 = face_grid(g, 2, nk);
    k_visc_flux<2><<<g2, block, 0, st>>>(g, nu_iso, p->is_ideal, w0, flx3, g.N3 + fs, g.N2, g.N1, nk);
  }
  AKMI_CHECK_LAUNCH("viscous_fluxes");
  return AKMI_COMPLETE;
}

int akmi_heat_fluxes(const akmi_pack *p, double alpha_iso, const double *w0, double *flx1,
                     double *flx2, double *flx3, int face_shaped, void *stream) {
  if (!p->is_ideal) {        // src/hydro/hydro.cpp:89-95
    set_error("heat_fluxes: thermal conduction requires the ideal gas EOS"); return AKMI_FAIL;
  }
  Geo g = make_geo(p);
  hipStream_t st = (hipStream_t)stream;
  const int fs = face_shaped ? 1 : 0;
  const double gm1 = p->gamma - 1.0;
  const dim3 block(DX, DY);
  int nk;
  dim3 g0 = face_grid(g, 0, nk);
  k_heat_flux<0><<<g0, block, 0, st>>>(g, alpha_iso, gm1, w0, flx1, g.N3, g.N2, g.N1 + fs, nk);
  if (g.multi_d) {
    dim3 g1 = face_grid(g, 1, nk);
    k_heat_flux<1><<<g1, block, 0, st>>>(g, alpha_iso, gm1, w0, flx2, g.N3, g.N2 + fs, g.N1, nk);
  }
  if (g.three_d) {
    dim3 g2 = face_grid(g, 2, nk);
    k_heat_flux<2><<<g2, block, 0, st>>>(g, alpha_iso, gm1, w0, flx3, g.N3 + fs, g.N2, g.N1, nk);
  }
  AKMI_CHECK_LAUNCH("heat_fluxes");
  return AKMI_COMPLETE;
}

int akmi_conduction_newdt(const akmi_pack *p, double alpha_iso, const double *w0, double *dtmin,
                          void *stream) {
  Geo g = make_geo(p);
  hipStream_t st = (hipStream_t)stream;
  k_set_fltmax<<<1, 64, 0, st>>>(dtmin);
  const int nk = g.ke - g.ks + 1;
  dim3 grid(cdiv(g.nx1, DX), cdiv(g.je - g.js + 1, DY), nk*g.nmb), block(DX, DY);
  k_cond_newdt<<<grid, block, 0, st>>>(g, alpha_iso, p->gamma - 1.0, w0, dtmin, nk);
  AKMI_CHECK_LAUNCH("conduction_newdt");
  return AKMI_COMPLETE;
}

int akmi_resistive_emfs(const akmi_pack *p, double eta_ohm, const double *bx1f, const double *bx2f,
                        const double *bx3f, double *e1, double *e2, double *e3, void *stream) {
  Geo g = make_geo(p);
  const int nk = g.three_d ? g.ke - g.ks + 2 : 1;
  dim3 grid(cdiv(g.nx1 + 1, DX), cdiv(g.multi_d ? g.nx2 + 1 : 1, DY), nk*g.nmb), block(DX, DY);
  k_resist_emf<<<grid, block, 0, (hipStream_t)stream>>>(g, eta_ohm, bx1f, bx2f, bx3f, e1, e2, e3, nk);
  AKMI_CHECK_LAUNCH("resistive_emfs");
  return AKMI_COMPLETE;
}

int akmi_resistive_fluxes(const akmi_pack *p, double eta_ohm, const double *bx1f, const double *bx2f,
                          const double *bx3f, double *flx1, double *flx2, double *flx3, void *stream) {
  if (!p->is_ideal) {        // mhd_tasks.cpp:204: only with an energy equation
    set_error("resistive_fluxes: needs the ideal gas EOS"); return AKMI_FAIL;
  }
  Geo g = make_geo(p);
  hipStream_t st = (hipStream_t)stream;
  const dim3 block(DX, DY);
  int nk;
  dim3 g0 = face_grid(g, 0, nk);
  k_resist_flux<0><<<g0, block, 0, st>>>(g, eta_ohm, bx1f, bx2f, bx3f, flx1, nk);
  if (g.multi_d) {
    dim3 g1 = face_grid(g, 1, nk);
    k_resist_flux<1><<<g1, block, 0, st>>>(g, eta_ohm, bx1f, bx2f, bx3f, flx2, nk);
  }
  if (g.three_d) {
    dim3 g2 = face_grid(g, 2, nk);
    k_resist_flux<2><<<g2, block, 0, st>>>(g, eta_ohm, bx1f, bx2f, bx3f, flx3, nk);
  }
  AKMI_CHECK_LAUNCH("resistive_fluxes");
  return AKMI_COMPLETE;
}

int akmi_ambipolar_emfs(const akmi_pack *p, double eta_ad, const double *bcc0, const double *bx1f,
                        const double *bx2f, const double *bx3f, double *e1, double *e2, double *e3,
                        void *stream) {
  Geo g = make_geo(p);
  const int nk = g.three_d ? g.ke - g.ks + 2 : 1;
  dim3 grid(cdiv(g.nx1 + 1, DX), cdiv(g.multi_d ? g.nx2 + 1 : 1, DY), nk*g.nmb), block(DX, DY);
  k_amb_emf<<<grid, block, 0, (hipStream_t)stream>>>(g, eta_ad, bcc0, bx1f, bx2f, bx3f, e1, e2, e3, nk);
  AKMI_CHECK_LAUNCH("ambipolar_emfs");
  return AKMI_COMPLETE;
}

int akmi_ambipolar_fluxes(const akmi_pack *p, double eta_ad, const double *bcc0, const double *bx1f,
                          const double *bx2f, const double *bx3f, double *flx1, double *flx2,
                          double *flx3, void *stream) {
  if (!p->is_ideal) {
    set_error("ambipolar_fluxes: needs the ideal gas EOS"); return AKMI_FAIL;
  }
  Geo g = make_geo(p);
  hipStream_t st = (hipStream_t)stream;
  const dim3 block(DX, DY);
  int nk;
  dim3 g0 = face_grid(g, 0, nk);
  k_amb_flux<0><<<g0, block, 0, st>>>(g, eta_ad, bcc0, bx1f, bx2f, bx3f, flx1, nk);
  if (g.multi_d) {
    dim3 g1 = face_grid(g, 1, nk);
    k_amb_flux<1><<<g1, block, 0, st>>>(g, eta_ad, bcc0, bx1f, bx2f, bx3f, flx2, nk);
  }
  if (g.three_d) {
    dim3 g2 = face_grid(g, 2, nk);
    k_amb_flux<2><<<g2, block, 0, st>>>(g, eta_ad, bcc0, bx1f, bx2f, bx3f, flx3, nk);
  }
  AKMI_CHECK_LAUNCH("ambipolar_fluxes");
  return AKMI_COMPLETE;
}

int akmi_resistive_newdt(const akmi_pack *p, double eta_ohm, double eta_ad, const double *bcc0,
                         double *dtmin, void *stream) {
  Geo g = make_geo(p);
  hipStream_t st = (hipStream_t)stream;
  k_set_fltmax<<<1, 64, 0, st>>>(dtmin);
  const int nk = g.ke - g.ks + 1;
  dim3 grid(cdiv(g.nx1, DX), cdiv(g.je - g.js + 1, DY), nk*g.nmb), block(DX, DY);
  k_resist_newdt<<<grid, block, 0, st>>>(g, eta_ohm, eta_ad, bcc0, dtmin, nk);
  AKMI_CHECK_LAUNCH("resistive_newdt");
  return AKMI_COMPLETE;
}

}  // extern "C"
