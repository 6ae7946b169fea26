// akmi_host.cpp -- C++ host mirror (see akmi_host.hpp) + its C entry points akmi_sim_*.
#include "akmi_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <sstream>

namespace akmi {
void set_error(const char *fmt, ...);
namespace host {

[[noreturn]] void Fatal(const char *file, int line, const std::string &msg) {
  std::fprintf(stderr, "### FATAL ERROR in %s at line %d\n%s\n", file, line, msg.c_str());
  std::exit(EXIT_FAILURE);     // the reference's error convention (e.g. src/mesh/mesh.cpp:234)
}

[[noreturn]] void Throw(const char *file, int line, const std::string &msg) {
  const char *b = std::strrchr(file, '/');
  throw HostError(std::string(b ? b + 1 : file) + ":" + std::to_string(line) + ": " + msg);
}
void NoteException(const char *entry) noexcept {
  try { throw; }
  catch (const std::exception &e) { akmi::set_error("%s: %s", entry, e.what()); }
  catch (...) { akmi::set_error("%s: unknown C++ exception", entry); }
}

#define HIPCHK(x)                                                                   \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) { (void)hipGetLastError(); AKMI_THROW(std::string(#x) + ": " + hipGetErrorString(e_)); } \
  } while (0)
#define AKCHK(x)                                                              \
  do {                                                                        \
    if ((x) < 0) AKMI_THROW(std::string(#x) + ": " + akmi_last_error());      \
  } while (0)

// ---- ParameterInput (src/parameter_input.cpp:155-209,369-409,508-552) -------------------
static std::string trim(const std::string &s) {
  size_t a = s.find_first_not_of(" \t\r"), b = s.find_last_not_of(" \t\r");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

void ParameterInput::LoadFromString(const std::string &text) {
  std::istringstream in(text);
  std::string raw, block;
  bool have = false;
  while (std::getline(in, raw)) {
    std::string line = trim(raw);
    if (line.empty() || line[0] == '#') continue;
    if (line[0] == '<') {
      std::string name = trim(line.substr(1, line.find('>') - 1));
      if (name == "par_end") break;
      block = name; have = true;
      blocks_[block];
      continue;
    }
    if (!have) AKMI_FATAL("parameter outside of a <block>: " + raw);
    size_t eq = line.find('=');
    if (eq == std::string::npos) AKMI_FATAL("no '=' in line: " + raw);
    std::string k = trim(line.substr(0, eq)), v = line.substr(eq + 1);
    size_t hash = v.find('#');
    if (hash != std::string::npos) v = v.substr(0, hash);
    blocks_[block][k] = trim(v);
  }
}

void ParameterInput::ModifyFromCmdline(const std::vector<std::string> &args) {
  for (const auto &a : args) {
    size_t sl = a.find('/'), eq = a.find('=');
    if (sl == std::string::npos || eq == std::string::npos) AKMI_FATAL("cannot parse override " + a);
    std::string b = a.substr(0, sl), n = a.substr(sl + 1, eq - sl - 1), v = a.substr(eq + 1);
    if (!DoesBlockExist(b)) AKMI_FATAL("block <" + b + "> not found");
    if (!DoesParameterExist(b, n)) AKMI_FATAL("parameter " + b + "/" + n + " not found");
    blocks_[b][n] = v;
  }
}

bool ParameterInput::DoesParameterExist(const std::string &b, const std::string &n) const {
  auto it = blocks_.find(b);
  return it != blocks_.end() && it->second.count(n) != 0;
}
std::string ParameterInput::GetString(const std::string &b, const std::string &n) const {
  if (!DoesParameterExist(b, n)) AKMI_FATAL("parameter " + b + "/" + n + " does not exist");
  return blocks_.at(b).at(n);
}
int ParameterInput::GetInteger(const std::string &b, const std::string &n) const {
  return std::atoi(GetString(b, n).c_str());
}
Real ParameterInput::GetReal(const std::string &b, const std::string &n) const {
  return std::strtod(GetString(b, n).c_str(), nullptr);
}
bool ParameterInput::GetBoolean(const std::string &b, const std::string &n) const {
  std::string v = GetString(b, n);
  std::transform(v.begin(), v.end(), v.begin(), ::tolower);
  if (v == "1" || v == "true") return true;
  if (v == "0" || v == "false") return false;
  AKMI_FATAL("bad boolean " + b + "/" + n + "=" + v);
}
std::string ParameterInput::GetOrAddString(const std::string &b, const std::string &n,
                                           const std::string &d) {
  if (DoesParameterExist(b, n)) return GetString(b, n);
  blocks_[b][n] = d;
  return d;
}
int ParameterInput::GetOrAddInteger(const std::string &b, const std::string &n, int d) {
  if (DoesParameterExist(b, n)) return GetInteger(b, n);
  blocks_[b][n] = std::to_string(d);
  return d;
}
Real ParameterInput::GetOrAddReal(const std::string &b, const std::string &n, Real d) {
  if (DoesParameterExist(b, n)) return GetReal(b, n);
  char buf[64]; std::snprintf(buf, sizeof(buf), "%.17g", d);
  blocks_[b][n] = buf;
  return d;
}
bool ParameterInput::GetOrAddBoolean(const std::string &b, const std::string &n, bool d) {
  if (DoesParameterExist(b, n)) return GetBoolean(b, n);
  blocks_[b][n] = d ? "true" : "false";
  return d;
}
void ParameterInput::SetReal(const std::string &b, const std::string &n, Real v) {
  // src/parameter_input.cpp:722-731 stores `stringstream << Real`: 6 significant digits
  char buf[64]; std::snprintf(buf, sizeof(buf), "%g", v);
  blocks_[b][n] = buf;
}

// ---- TaskList ---------------------------------------------------------------------------
bool TaskList::IsComplete() {
  for (auto &it : task_list_)
    if (!tasks_completed_.CheckDependencies(it.GetID())) return false;
  return true;
}
void TaskList::Reset() {
  tasks_completed_.Clear();
  for (auto &it : task_list_) it.SetIncomplete();
}
TaskListStatus TaskList::DoAvailable(Driver *d, int s) {
  for (auto &task : task_list_) {
    auto dep = task.GetDependency();
    if (tasks_completed_.CheckDependencies(dep) && !task.IsComplete()) {
      TaskStatus status = task(d, s);
      if (status == TaskStatus::fail) AKMI_FATAL("task failed");
      if (status == TaskStatus::complete) {
        task.SetComplete();
        tasks_completed_.SetComplete(task.GetID());
      }
    }
  }
  return IsComplete() ? TaskListStatus::complete : TaskListStatus::running;
}

// ---- device arrays ------------------------------------------------------------------------
template <typename T> void DvceArray<T>::Realloc(size_t count) {
  Free();
  // AKMI_FAIL_ALLOC_AFTER=n (tests): the n-th device allocation of the process fails as an exhausted device would
  static const long fail_at = std::getenv("AKMI_FAIL_ALLOC_AFTER") ? std::atol(std::getenv("AKMI_FAIL_ALLOC_AFTER")) : -1;
  static long nalloc = 0;
  if (fail_at >= 0 && ++nalloc > fail_at) AKMI_THROW("device allocation of " + std::to_string(count*sizeof(T)) + " bytes failed (injected)");
  n = count;
  HIPCHK(hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(count, 1)*sizeof(T)));
  HIPCHK(hipMemset(p, 0, std::max<size_t>(count, 1)*sizeof(T)));
}
template <typename T> void DvceArray<T>::Free() {
  if (p) (void)hipFree(p);
  p = nullptr; n = 0;
}
template struct DvceArray<Real>;
template struct DvceArray<int>;
template struct DvceArray<char>;
template struct DvceArray<unsigned char>;
template struct DvceArray<long long>;

// ---- Mesh -----------------------------------------------------------------------------------
static Real LeftEdgeX(int ith, int n, Real xmin, Real xmax) {   // cell_locations.hpp:23-28
  Real x = static_cast<Real>(ith)/static_cast<Real>(n);
  return (x*xmax - x*xmin) - (0.5*xmax - 0.5*xmin) + (0.5*xmin + 0.5*xmax);
}
static std::uint64_t Morton(int x, int y, int z) {
  std::uint64_t r = 0;
  for (int b = 0; b < 20; ++b) {
    r |= (static_cast<std::uint64_t>((x >> b) & 1)) << (3*b);
    r |= (static_cast<std::uint64_t>((y >> b) & 1)) << (3*b + 1);
    r |= (static_cast<std::uint64_t>((z >> b) & 1)) << (3*b + 2);
  }
  return r;
}
static RegionIndcs MakeIndcs(int ng, int nx1, int nx2, int nx3) {   // mesh.cpp:285-330
  RegionIndcs r;
  r.ng = ng; r.nx1 = nx1; r.nx2 = nx2; r.nx3 = nx3;
  r.is = ng; r.ie = ng + nx1 - 1;
  r.js = nx2 > 1 ? ng : 0; r.je = nx2 > 1 ? ng + nx2 - 1 : 0;
  r.ks = nx3 > 1 ? ng : 0; r.ke = nx3 > 1 ? ng + nx3 - 1 : 0;
  return r;
}
static int BCFlag(const std::string &v) {
  if (v == "periodic") return AKMI_BC_PERIODIC;
  if (v == "outflow") return AKMI_BC_OUTFLOW;
  if (v == "reflect") return AKMI_BC_REFLECT;
  if (v == "diode") return AKMI_BC_DIODE;       // hydro_bcs.cpp:105-113
  if (v == "vacuum") return AKMI_BC_VACUUM;     // hydro_bcs.cpp:114-118
  AKMI_FATAL("boundary flag '" + v + "' not supported by the C++ host (periodic/outflow/reflect/diode/vacuum; inflow and user: Python host)");
}

Mesh::Mesh(ParameterInput *pin, int my_rank_, int nranks_, bool host_only_)
    : my_rank(my_rank_), nranks(nranks_), host_only(host_only_) {
  mesh_size.x1min = pin->GetReal("mesh", "x1min"); mesh_size.x1max = pin->GetReal("mesh", "x1max");
  mesh_size.x2min = pin->GetReal("mesh", "x2min"); mesh_size.x2max = pin->GetReal("mesh", "x2max");
  mesh_size.x3min = pin->GetReal("mesh", "x3min"); mesh_size.x3max = pin->GetReal("mesh", "x3max");
  int ng = pin->GetOrAddInteger("mesh", "nghost", 2);
  int nx1 = pin->GetInteger("mesh", "nx1"), nx2 = pin->GetInteger("mesh", "nx2"),
      nx3 = pin->GetInteger("mesh", "nx3");
  mesh_indcs = MakeIndcs(ng, nx1, nx2, nx3);
  one_d = (nx2 == 1 && nx3 == 1); two_d = (nx2 > 1 && nx3 == 1); three_d = nx3 > 1; multi_d = nx2 > 1;
  if (nx2 == 1 && nx3 > 1) AKMI_FATAL("In mesh block in input file nx3>1 requires nx2>1");
  if (ng < 2) AKMI_FATAL("More than 1 ghost zone required");
  const char *names[6] = {"ix1_bc", "ox1_bc", "ix2_bc", "ox2_bc", "ix3_bc", "ox3_bc"};
  for (int q = 0; q < 6; ++q) mesh_bcs[q] = BCFlag(pin->GetOrAddString("mesh", names[q], "periodic"));
  auto per = [&](int q) { return mesh_bcs[q] == AKMI_BC_PERIODIC; };
  strictly_periodic = per(0) && per(1) && (!multi_d || (per(2) && per(3))) &&
                      (!three_d || (per(4) && per(5)));
  int mb1 = pin->GetOrAddInteger("meshblock", "nx1", nx1);
  int mb2 = pin->GetOrAddInteger("meshblock", "nx2", nx2);
  int mb3 = pin->GetOrAddInteger("meshblock", "nx3", nx3);
  if (nx1 % mb1 || nx2 % mb2 || nx3 % mb3) AKMI_FATAL("Mesh must be evenly divisible by MeshBlocks");
  mb_indcs = MakeIndcs(ng, mb1, mb2, mb3);
  nmb_rootx1 = nx1/mb1; nmb_rootx2 = nx2/mb2; nmb_rootx3 = nx3/mb3;
  nmb_total = nmb_rootx1*nmb_rootx2*nmb_rootx3;
  std::string ref = "none";
  if (pin->DoesBlockExist("mesh_refinement")) ref = pin->GetOrAddString("mesh_refinement", "refinement", "none");
  if (ref == "static") {
    // mesh_refinement.cpp:52: prolongate primitive instead of conserved variables into fine ghost zones
    prolong_prims = pin->GetOrAddBoolean("mesh_refinement", "prolong_primitives", false);
    multilevel = true;
    BuildTreeFromScratch(pin);                 // akmi_host_smr.cpp
  } else if (ref != "none") {
    AKMI_FATAL("<mesh_refinement>/refinement = '" + ref + "': only static refinement is on this build's path");
  } else {
    // Z-ordered logical locations (build_tree.cpp:243-258)
    struct Z { std::uint64_t key; int l[3]; };
    std::vector<Z> z;
    for (int l3 = 0; l3 < nmb_rootx3; ++l3)
      for (int l2 = 0; l2 < nmb_rootx2; ++l2)
        for (int l1 = 0; l1 < nmb_rootx1; ++l1) z.push_back({Morton(l1, l2, l3), {l1, l2, l3}});
    std::sort(z.begin(), z.end(), [](const Z &a, const Z &b) { return a.key < b.key; });
    lloc_eachmb.resize(3*nmb_total);
    for (int m = 0; m < nmb_total; ++m)
      for (int q = 0; q < 3; ++q) lloc_eachmb[3*m + q] = z[m].l[q];
  }
  time = pin->GetOrAddReal("time", "start_time", 0.0);
  dt = static_cast<Real>(FLT_MAX);            // build_tree.cpp:301
  dtold = 0.0;
  cfl_no = pin->GetReal("time", "cfl_number");
  ncycle = 0;
  // every MeshBlock costs the same (build_tree.cpp:262-272); one pack per rank (mesh.cpp:205-215)
  LoadBalance(std::vector<float>(nmb_total, 1.0f));
  const int gs = gids_eachrank[my_rank], nb = nmb_eachrank[my_rank];
  pmb_pack = new MeshBlockPack(this, gs, gs + nb - 1);
  pmb_pack->pmb = new MeshBlock(pmb_pack, gs, nb);
}
Mesh::~Mesh() { delete pmb_pack; }

MeshBlock::MeshBlock(MeshBlockPack *ppack, int igids, int nmb_) : nmb(nmb_) {
  Mesh *pm = ppack->pmesh;
  const RegionSize &ms = pm->mesh_size;
  const int nbr[3] = {pm->nmb_rootx1, pm->nmb_rootx2, pm->nmb_rootx3};
  const bool active[3] = {true, pm->multi_d, pm->three_d};
  const Real mmin[3] = {ms.x1min, ms.x2min, ms.x3min}, mmax[3] = {ms.x1max, ms.x2max, ms.x3max};
  const int nxb[3] = {pm->mb_indcs.nx1, pm->mb_indcs.nx2, pm->mb_indcs.nx3};
  mb_gid.resize(nmb); mb_size.resize(nmb); mb_bcs.resize(6*nmb);
  nghbr_gid.assign(27*nmb, -1); nghbr_rank.assign(27*nmb, -1);
  mb_lev.assign(nmb, pm->root_level);
  std::vector<int> gid_of(pm->multilevel ? 0 : pm->nmb_total);      // logical location -> global id
  for (int g = 0; g < pm->nmb_total && !pm->multilevel; ++g) {
    const int *l = &pm->lloc_eachmb[3*g];
    gid_of[(l[2]*nbr[1] + l[1])*nbr[0] + l[0]] = g;
  }
  std::vector<Real> dx(3*nmb);
  for (int m = 0; m < nmb; ++m) {
    mb_gid[m] = igids + m;
    if (pm->multilevel) mb_lev[m] = pm->lloc_tree[igids + m].level;
    // blocks per direction at the level of this block: nmb_rootx << (lev - root_level), meshblock.cpp:42
    int nb[3];
    for (int q = 0; q < 3; ++q) nb[q] = nbr[q] << (mb_lev[m] - pm->root_level);
    const int *l = &pm->lloc_eachmb[3*(igids + m)];
    Real lim[6];
    for (int q = 0; q < 3; ++q) {
      if (!active[q] || l[q] == 0) { lim[2*q] = mmin[q]; mb_bcs[6*m + 2*q] = pm->mesh_bcs[2*q]; }
      else { lim[2*q] = LeftEdgeX(l[q], nb[q], mmin[q], mmax[q]); mb_bcs[6*m + 2*q] = AKMI_BC_BLOCK; }
      if (!active[q] || l[q] == nb[q] - 1) { lim[2*q + 1] = mmax[q]; mb_bcs[6*m + 2*q + 1] = pm->mesh_bcs[2*q + 1]; }
      else { lim[2*q + 1] = LeftEdgeX(l[q] + 1, nb[q], mmin[q], mmax[q]); mb_bcs[6*m + 2*q + 1] = AKMI_BC_BLOCK; }
      dx[3*m + q] = (lim[2*q + 1] - lim[2*q])/static_cast<Real>(nxb[q]);
    }
    RegionSize &s = mb_size[m];
    s.x1min = lim[0]; s.x1max = lim[1]; s.x2min = lim[2]; s.x2max = lim[3]; s.x3min = lim[4]; s.x3max = lim[5];
    s.dx1 = dx[3*m]; s.dx2 = dx[3*m + 1]; s.dx3 = dx[3*m + 2];
    for (int d = 0; d < 27 && !pm->multilevel; ++d) {
      int o[3] = {d%3 - 1, (d/3)%3 - 1, d/9 - 1};
      if (d == 13 || (!pm->multi_d && o[1]) || (!pm->three_d && o[2])) continue;
      int ll[3]; bool ok = true;
      for (int q = 0; q < 3 && ok; ++q) {
        ll[q] = l[q] + o[q];
        if (ll[q] < 0) { if (pm->mesh_bcs[2*q] == AKMI_BC_PERIODIC) ll[q] += nb[q]; else ok = false; }
        else if (ll[q] >= nb[q]) { if (pm->mesh_bcs[2*q + 1] == AKMI_BC_PERIODIC) ll[q] -= nb[q]; else ok = false; }
      }
      if (ok) {
        const int g = gid_of[(ll[2]*nb[1] + ll[1])*nb[0] + ll[0]];
        nghbr_gid[27*m + d] = g;
        nghbr_rank[27*m + d] = pm->rank_eachmb[g];
      }
    }
  }
  BuildMeshBlockPlan(this, pm->my_rank, igids);       // akmi_host_comm.cpp: plan.tab = the device table
  if (pm->host_only) return;
  d_dx.Realloc(3*nmb); d_bcs.Realloc(6*nmb); d_nghbr.Realloc(27*nmb);
  HIPCHK(hipMemcpy(d_dx.p, dx.data(), sizeof(Real)*3*nmb, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_bcs.p, mb_bcs.data(), sizeof(int)*6*nmb, hipMemcpyHostToDevice));
  bc_dirs = 0;
  for (int m = 0; m < nmb; ++m)
    for (int f = 0; f < 6; ++f)
      if (mb_bcs[6*m + f] != AKMI_BC_BLOCK && mb_bcs[6*m + f] != AKMI_BC_PERIODIC) bc_dirs |= 1 << (f/2);
  HIPCHK(hipMemcpy(d_nghbr.p, plan.tab.data(), sizeof(int)*27*nmb, hipMemcpyHostToDevice));
  if (pm->multilevel) SetNeighborsSMR(pm);
}
MeshBlock::~MeshBlock() { d_dx.Free(); d_bcs.Free(); d_nghbr.Free(); }

MeshBlockPack::MeshBlockPack(Mesh *pm, int igids, int igide)
    : pmesh(pm), gids(igids), gide(igide), nmb_thispack(igide - igids + 1) {
  for (const char *n : {"before_timeintegrator", "after_timeintegrator", "before_stagen", "stagen",
                        "after_stagen"})
    tl_map[n] = std::make_shared<TaskList>();          // meshblock_pack.cpp:40-50
}
MeshBlockPack::~MeshBlockPack() { delete phydro; delete pmhd; delete pmb; }

void MeshBlockPack::AddPhysics(ParameterInput *pin) {   // meshblock_pack.cpp:102-262
  int nphys = 0;
  if (pin->DoesBlockExist("hydro")) { phydro = new hydro::Hydro(this, pin); ++nphys; }
  if (pin->DoesBlockExist("mhd")) { pmhd = new mhd::MHD(this, pin); ++nphys; }
  if (nphys == 0) AKMI_FATAL("At least one physics module must be specified in input file");
  if (phydro) phydro->AssembleHydroTasks(tl_map);
  if (pmhd) pmhd->AssembleMHDTasks(tl_map);
}

void Mesh::NewTimeStep(const Real tlim) {               // mesh.cpp:573-643
  dtold = dt;
  if (dt == static_cast<Real>(FLT_MAX)) dtold = 0.;
  dt = 2.0*dt;
  FluidBase *phys[2] = {pmb_pack->phydro, pmb_pack->pmhd};
  for (FluidBase *f : phys) {
    if (!f) continue;
    dt = std::min(dt, cfl_no*f->dtnew);
    if (f->has_visc) dt = std::min(dt, cfl_no*f->dt_visc);          // mesh.cpp:589-612
    if (f->has_resist) dt = std::min(dt, cfl_no*f->dt_resist);
    if (f->has_cond) dt = std::min(dt, cfl_no*f->dt_cond);
  }
  // minimum over all ranks (mesh.cpp:634-637): ncclAllReduce(ncclMin) on the compute stream
  // (already done on the device inside FinishNewDt when every physics module reports dt_reduced)
  bool reduced = true;
  for (FluidBase *f : phys) if (f && !f->dt_reduced) reduced = false;
  for (FluidBase *f : phys)
    if (f && !reduced && (nranks > 1 || SelfExchange())) { Comm::World().AllReduceMin(&dt, 1, f->stream); break; }
  if ((time < tlim) && ((time + dt) > tlim)) dt = tlim - time;
}

// ---- physics ----------------------------------------------------------------------------------
static int ReconFlag(const std::string &r) {
  if (r == "dc") return AKMI_RECON_DC;
  if (r == "plm") return AKMI_RECON_PLM;
  if (r == "ppm4") return AKMI_RECON_PPM4;
  if (r == "ppmx") return AKMI_RECON_PPMX;
  if (r == "wenoz") return AKMI_RECON_WENOZ;
  if (r == "teno") return AKMI_RECON_TENO;
  AKMI_FATAL("reconstruct = '" + r + "' not implemented on this path");
}

FluidBase::FluidBase(MeshBlockPack *pp, ParameterInput *pin, const std::string &blk) : pmy_pack(pp) {
  peos = new EquationOfState;
  EOS_Data &e = peos->eos_data;
  const std::string eqn_of_state = pin->GetString(blk, "eos");     // hydro.cpp:52-72
  if (eqn_of_state == "ideal") {
    e.is_ideal = true; e.gamma = pin->GetReal(blk, "gamma"); e.iso_cs = 0.0;
  } else if (eqn_of_state == "isothermal") {
    e.is_ideal = false; e.gamma = 0.0; e.iso_cs = pin->GetReal(blk, "iso_sound_speed");
  } else {
    AKMI_FATAL("<" + blk + ">/eos = '" + eqn_of_state + "' not implemented");
  }
  e.dfloor = pin->GetOrAddReal(blk, "dfloor", static_cast<Real>(FLT_MIN));   // eos.cpp:22-25
  e.pfloor = pin->GetOrAddReal(blk, "pfloor", static_cast<Real>(FLT_MIN));
  e.tfloor = pin->GetOrAddReal(blk, "tfloor", static_cast<Real>(FLT_MIN));
  e.sfloor = pin->GetOrAddReal(blk, "sfloor", static_cast<Real>(FLT_MIN));
  e.sigma_max = pin->GetOrAddReal(blk, "sigma_max", static_cast<Real>(FLT_MAX));
  const RegionIndcs &ind = pp->pmesh->mb_indcs;
  std::string rec = pin->GetOrAddString(blk, "reconstruct", "plm");
  recon_method = ReconFlag(rec);
  if (recon_method >= AKMI_RECON_PPM4 && ind.ng < 3)
    AKMI_FATAL("PPM/WENOZ reconstruction requires at least 3 ghost zones");
  nscalars = pin->GetOrAddInteger(blk, "nscalars", 0);
  nfluid = e.is_ideal ? 5 : 4;                 // no energy variable with the isothermal EOS
  nvars = nfluid + nscalars;                   // scalars follow the fluid variables
  // diffusion objects: hydro.cpp:77-98, mhd.cpp:104-130 (constant isotropic coefficients)
  for (const char *n : {"nu_aniso", "alpha_aniso", "alpha_spitzer"})
    if (pin->DoesParameterExist(blk, n))
      AKMI_FATAL(std::string("<") + blk + ">/" + n + " is not on this path");
  if (pin->DoesParameterExist(blk, "nu_iso")) { has_visc = true; nu_iso = pin->GetReal(blk, "nu_iso"); }
  if (pin->DoesParameterExist(blk, "alpha_iso")) {
    if (!e.is_ideal) AKMI_FATAL("Thermal conduction requires ideal gas EOS");
    has_cond = true; alpha_iso = pin->GetReal(blk, "alpha_iso"); dtmin_cond.Realloc(1);
  }
  if (blk == "mhd" && (pin->DoesParameterExist(blk, "eta_ohm") || pin->DoesParameterExist(blk, "eta_ad"))) {
    has_resist = true;                                       // mhd.cpp:121-130, resistivity.cpp:24-36
    eta_ohm = pin->GetOrAddReal(blk, "eta_ohm", 0.0);
    eta_ad = pin->GetOrAddReal(blk, "eta_ad", 0.0);
    dtmin_cond.Realloc(1);                                   // scratch double of the cell reductions
  }
  // <hydro|mhd>/fused_stage = true | false | auto (default): an explicit true / false is kept as it is
  std::string fs = pin->GetOrAddString(blk, "fused_stage", "auto");
  for (char &c : fs) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
  // (the boolean spellings of ParameterInput::GetBoolean, src/parameter_input.cpp: "true" / "false" in any case, or an
  //  integer, non-zero = true -- decks written for the boolean this key was before round 4 keep working)
  const bool fs_int = !fs.empty() && fs.find_first_not_of("0123456789") == std::string::npos;
  if (fs != "auto" && fs != "true" && fs != "false" && !fs_int)
    AKMI_FATAL("<" + blk + ">/fused_stage = " + fs + ": true, false, an integer or auto");
  const bool fused_given = fs != "auto";
  fused = fs_int ? std::stol(fs) != 0 : fs != "false";
  // small 3-D MHD packs: the task-granular chain (one thread per face) beats the marching kernels of the fused stage
  // (64^3: 1 113 against 1 042 Mcell-updates/s, equal at 72^3); hydro packs keep the fused stage at every size (64^3: 2 838
  // against 1 854) -- profiles/r06_small_packs.txt; same bits either way.
  // <hydro|mhd>/small_pack_tasks = false keeps the fused kernels.  AKMI_SMALL_PACK_TASKS=0: off
  {
    // (read without adding it to the deck: the parameter dump of the output files stays what the reference's is)
    const bool small_ok = pin->DoesParameterExist(blk, "small_pack_tasks") ? pin->GetBoolean(blk, "small_pack_tasks") : true;
    const char *sp = std::getenv("AKMI_SMALL_PACK_TASKS");
    const long ncell_pack = static_cast<long>(pp->nmb_thispack)*ind.nx1*ind.nx2*ind.nx3;
    if (blk == "mhd" && small_ok && !fused_given && fused && ind.nx3 > 1 && nscalars == 0 && !(sp && std::atoi(sp) == 0) &&
        ncell_pack <= AKMI_SMALL_PACK_CELLS)
      fused = false;                   // (not with passive scalars: the task path's sweeps do not carry them)
  }
  // the fused stage kernels cover both equations of state and carry passive scalars along; extra fluxes
  // (diffusion), FOFC and refined meshes use the task-granular kernels
  if (has_visc || has_cond || has_resist) fused = false;
  pack_c.nmb = pp->nmb_thispack; pack_c.nvar = nvars;
  pack_c.nx1 = ind.nx1; pack_c.nx2 = ind.nx2; pack_c.nx3 = ind.nx3; pack_c.ng = ind.ng;
  pack_c.dx = pp->pmb->d_dx.p;
  pack_c.gamma = e.gamma; pack_c.dfloor = e.dfloor; pack_c.pfloor = e.pfloor;
  pack_c.tfloor = e.tfloor; pack_c.sfloor = e.sfloor; pack_c.sigma_max = e.sigma_max;
  pack_c.iso_cs = e.iso_cs; pack_c.is_ideal = e.is_ideal ? 1 : 0;
  const size_t n1 = ind.nx1 + 2*ind.ng, n2 = ind.nx2 > 1 ? ind.nx2 + 2*ind.ng : 1,
               n3 = ind.nx3 > 1 ? ind.nx3 + 2*ind.ng : 1;
  const size_t ncc = static_cast<size_t>(pp->nmb_thispack)*nvars*n3*n2*n1;
  u0.Realloc(ncc); w0.Realloc(ncc); u1.Realloc(ncc);
  counters.Realloc(3); dt3.Realloc(3);
  multilevel = pp->pmesh->multilevel;
  const bool fused_req = fused;
  if (multilevel) {
    // restricted fluxes replace face fluxes between Fluxes and RKUpdate: the task-granular flux arrays
    fused = false;
    cpack_c = pack_c;
    cpack_c.nx1 = ind.nx1/2; cpack_c.nx2 = ind.nx2 > 1 ? ind.nx2/2 : 1; cpack_c.nx3 = ind.nx3 > 1 ? ind.nx3/2 : 1;
    const size_t c1 = cpack_c.nx1 + 2*ind.ng, c2 = ind.nx2 > 1 ? cpack_c.nx2 + 2*ind.ng : 1,
                 c3 = ind.nx3 > 1 ? cpack_c.nx3 + 2*ind.ng : 1;
    coarse_u0.Realloc(static_cast<size_t>(pp->nmb_thispack)*nvars*c3*c2*c1);
    psmr = new MeshBoundaryValuesSMR(pp, nvars);
    psmr->BuildLists(&pack_c, stream);
    psmr->BuildCcMap(&pack_c, stream);
    if (blk == "mhd") psmr->BuildFcMaps(&pack_c, stream);
  }
  if (!multilevel && (pp->pmesh->nranks > 1 || SelfExchange()))
    pbval = new MeshBoundaryValues(pp, &pack_c, nvars, blk == "mhd");
  // <mesh_refinement>/prolong_primitives converts with SingleC2P_IdealHyd / _IdealMHD whatever the EOS of the run is
  // (prolong_prims.cpp:35-186); offered for the ideal gas only -- said at construction, not by the first Prolongate
  if (pp->pmesh->multilevel && pp->pmesh->prolong_prims && !e.is_ideal)
    AKMI_FATAL("<mesh_refinement>/prolong_primitives = true needs the ideal-gas EOS (<" + blk + ">/eos = " +
               pin->GetString(blk, "eos") + ")");
  use_fofc = pin->GetOrAddBoolean(blk, "fofc", false);     // hydro.cpp:153-190, mhd.cpp:199-235
  if (use_fofc) {
    const int need = recon_method == AKMI_RECON_PLM ? 3 : (recon_method >= AKMI_RECON_PPM4 ? 4 : 2);
    if (ind.ng < need)
      AKMI_FATAL("FOFC and this reconstruction require at least " + std::to_string(need) +
                 " ghost zones, but <mesh>/nghost=" + std::to_string(ind.ng));
    if (nscalars > 0 || (blk == "mhd" && !e.is_ideal))
      AKMI_FATAL("<" + blk + ">/fofc with passive scalars or (MHD) the isothermal EOS is not on this path");
    fused = false;                      // FOFC works on the flux arrays of the task-granular path
    fofc.Realloc(static_cast<size_t>(pp->nmb_thispack)*n3*n2*n1);    // zero-filled
    nfofc.Realloc(1);
  }
}
FluidBase::~FluidBase() {
  u0.Free(); w0.Free(); u1.Free(); w1.Free(); counters.Free(); dt3.Free(); ws.Free(); fofc.Free(); nfofc.Free();
  dtmin_cond.Free(); coarse_u0.Free(); coarse_w0.Free();
  delete psmr;
  delete pbval;
  delete peos;
}
// may the ghost zones of the primitives be filled like those of the conserved variables (akmi_hydro_ghost_uw)?  Only where
// every boundary's value rule commutes with ConsToPrim: neighbour / periodic copies, outflow, reflect
bool FluidBase::BcsCommuteWithC2P() const {
  for (int f : pmy_pack->pmb->mb_bcs)
    if (f != AKMI_BC_BLOCK && f != AKMI_BC_PERIODIC && f != AKMI_BC_OUTFLOW && f != AKMI_BC_REFLECT) return false;
  return true;
}
bool FluidBase::FoldBCs() {
  const char *e = getenv("AKMI_FOLD_BCS");      // read per call: tests switch it inside one process
  return !(e && atoi(e) == 0);
}
// same-rank gather of the conserved variables; on meshes with physical boundaries the boundary functions ride along
void FluidBase::GatherU(Driver *d, int stage) {
  if (want_ghost_c2p_) {
    // the stage kernel has converted the active cells (akmi_hydro_stage_w): ghost zones of u0 AND of the new primitive array
    // by the same gather + boundary functions, one launch, nothing converted twice (see akmi_hydro_ghost_uw)
    want_ghost_c2p_ = false;
    AKCHK(akmi_hydro_ghost_uw(&pack_c, pmy_pack->pmb->d_nghbr.p, pmy_pack->pmb->d_bcs.p, pmy_pack->pmb->mb_bcs.data(), u0.p,
                              w0.p, ws.p, counters.p, stream));
    u_bcs_done_ = true; shell_done_ = true;
    return;
  }
  want_ghost_c2p_ = false;
  if (FoldBCs() && !pmy_pack->pmesh->strictly_periodic) {
    // the last stage of the fused path: the launch also resets the CFL minima its ConsToPrim scans into
    // (not when the stage call has converted the active cells already: their scan is in dt3 by now)
    Real *reset = (fused && !interior_done_ && !d->ra_active && stage >= 1 && stage == d->nexp_stages) ? dt3.p : nullptr;
    AKCHK(akmi_bvals_cc_local_bcs(&pack_c, nvars, pmy_pack->pmb->d_nghbr.p, pmy_pack->pmb->d_bcs.p, nullptr, u0.p, reset,
                                  stream));
    u_bcs_done_ = true;
    dt3_reset_ = reset != nullptr;
  } else {
    AKCHK(akmi_bvals_cc_local(&pack_c, nvars, pmy_pack->pmb->d_nghbr.p, u0.p, stream));
  }
}
void FluidBase::FinishNewDt() {        // hydro_newdt.cpp:121-124
  Real d[3];
  Mesh *pm = pmy_pack->pmesh;
  // several ranks, RCCL transport, no diffusion time steps: the minimum over the ranks (mesh.cpp:634-637) is taken
  // of the three device-side values before they are read back -- ncclAllReduce(min) in place on the compute stream
  // -- so a cycle has ONE host synchronisation (this read-back, which the reference has too: hydro_newdt.cpp:121)
  // instead of read-back + H2D + all-reduce + read-back.  min_r(min(2 dt_old, cfl dt_r)) == min(2 dt_old, cfl min_r dt_r).
  dt_reduced = false;
  const bool several = pm->nranks > 1 || SelfExchange();
  if (several) Comm::World().ProfMark(Comm::kDtReduce, stream);
  if (several && !has_visc && !has_cond && !has_resist)
    dt_reduced = Comm::World().AllReduceMinDevice(dt3.p, 3, stream);
  HIPCHK(hipMemcpyAsync(d, dt3.p, sizeof(d), hipMemcpyDeviceToHost, stream));
  if (several) Comm::World().ProfMark(Comm::kDtReduce, stream);
  HIPCHK(hipStreamSynchronize(stream));
  dtnew = d[0];
  if (pm->multi_d) dtnew = std::min(dtnew, d[1]);
  if (pm->three_d) dtnew = std::min(dtnew, d[2]);
}

void FluidBase::AddDiffusionFluxes(DvceFaceFld &flx, int fs) {   // hydro_tasks.cpp:183-189, mhd_tasks.cpp:198-203
  if (has_cond && alpha_iso != 0.0)
    AKCHK(akmi_heat_fluxes(&pack_c, alpha_iso, w0.p, flx.x1f.p, flx.x2f.p, flx.x3f.p, fs, stream));
  if (has_visc && nu_iso != 0.0)
    AKCHK(akmi_viscous_fluxes(&pack_c, nu_iso, w0.p, flx.x1f.p, flx.x2f.p, flx.x3f.p, fs, stream));
}
void FluidBase::DiffusionNewDt() {     // viscosity.cpp:232-251, conduction.cpp:314-377, resistivity.cpp:291-311
  Mesh *pm = pmy_pack->pmesh;
  const Real fac = pm->three_d ? 1.0/6.0 : (pm->two_d ? 0.25 : 0.5);
  auto const_dt = [&](Real coeff) {
    Real dt = static_cast<Real>(FLT_MAX);
    for (const RegionSize &sz : pmy_pack->pmb->mb_size) {
      dt = std::min(dt, fac*(sz.dx1*sz.dx1)/coeff);
      if (pm->multi_d) dt = std::min(dt, fac*(sz.dx2*sz.dx2)/coeff);
      if (pm->three_d) dt = std::min(dt, fac*(sz.dx3*sz.dx3)/coeff);
    }
    return dt;
  };
  if (has_visc) dt_visc = nu_iso != 0.0 ? const_dt(nu_iso) : static_cast<Real>(FLT_MAX);
  if (has_resist && eta_ad == 0.0) {                         // resistivity.cpp:298-311
    dt_resist = eta_ohm > 0.0 ? const_dt(eta_ohm) : static_cast<Real>(FLT_MAX);
  } else if (has_resist) {                                   // resistivity.cpp:313-345
    AKCHK(akmi_resistive_newdt(&pack_c, eta_ohm, eta_ad, bcc_cells, dtmin_cond.p, stream));
    Real d;
    HIPCHK(hipMemcpyAsync(&d, dtmin_cond.p, sizeof(d), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    dt_resist = d*fac;
  }
  if (has_cond) {
    if (alpha_iso != 0.0) {
      AKCHK(akmi_conduction_newdt(&pack_c, alpha_iso, w0.p, dtmin_cond.p, stream));
      Real d;
      HIPCHK(hipMemcpyAsync(&d, dtmin_cond.p, sizeof(d), hipMemcpyDeviceToHost, stream));
      HIPCHK(hipStreamSynchronize(stream));
      dt_cond = d*fac;
    } else {
      dt_cond = static_cast<Real>(FLT_MAX)*fac;
    }
  }
}

static void FaceAlloc(DvceFaceFld &f, size_t nmb, size_t nv, size_t n3, size_t n2, size_t n1, int fs) {
  f.x1f.Realloc(nmb*nv*n3*n2*(n1 + fs)); f.x2f.Realloc(nmb*nv*n3*(n2 + fs)*n1);
  f.x3f.Realloc(nmb*nv*(n3 + fs)*n2*n1);
}
static void FaceFree(DvceFaceFld &f) { f.x1f.Free(); f.x2f.Free(); f.x3f.Free(); }

namespace hydro {
Hydro::Hydro(MeshBlockPack *pp, ParameterInput *pin) : FluidBase(pp, pin, "hydro") {
  const std::string rs = pin->GetString("hydro", "rsolver");
  // dynamic problems: llf/hlle/hllc/roe; kinematic problems: advect (hydro.cpp:244-278)
  kinematic = pin->GetOrAddString("time", "evolution", "dynamic") == "kinematic";
  if (kinematic) {
    if (rs != "advect") AKMI_FATAL("<hydro> rsolver = '" + rs + "' not implemented for kinematic problems");
    rsolver_method = AKMI_RS_ADVECT; fused = false;
  } else if (rs == "llf") rsolver_method = AKMI_RS_LLF;
  else if (rs == "hlle") rsolver_method = AKMI_RS_HLLE;
  else if (rs == "hllc") rsolver_method = AKMI_RS_HLLC;
  else if (rs == "roe") rsolver_method = AKMI_RS_ROE;
  else AKMI_FATAL("<hydro> rsolver = '" + rs + "' not implemented (llf, hlle, hllc, roe)");
  const RegionIndcs &ind = pp->pmesh->mb_indcs;
  const size_t n1 = ind.nx1 + 2*ind.ng, n2 = ind.nx2 > 1 ? ind.nx2 + 2*ind.ng : 1,
               n3 = ind.nx3 > 1 ? ind.nx3 + 2*ind.ng : 1;
  if (fused) ws.Realloc(static_cast<size_t>(akmi_stage_workspace_bytes(&pack_c, 0)));
  else FaceAlloc(uflx, pp->nmb_thispack, nvars, n3, n2, n1, 0);  // hydro.cpp:290-292
}
Hydro::~Hydro() { FaceFree(uflx); }

void Hydro::AssembleHydroTasks(std::map<std::string, std::shared_ptr<TaskList>> tl) {
  TaskID none(0);                                                  // hydro_tasks.cpp:48-80
  tl["before_stagen"]->AddTask(&Hydro::InitRecv, this, none);
  auto &s = tl["stagen"];
  TaskID copyu = s->AddTask(&Hydro::CopyCons, this, none);
  TaskID flux = s->AddTask(&Hydro::Fluxes, this, copyu);
  TaskID sendf = s->AddTask(&Hydro::SendFlux, this, flux);
  TaskID recvf = s->AddTask(&Hydro::RecvFlux, this, sendf);
  TaskID rkupdt = s->AddTask(&Hydro::RKUpdate, this, recvf);
  TaskID srctrms = s->AddTask(&Hydro::HydroSrcTerms, this, rkupdt);
  TaskID restu = s->AddTask(&Hydro::RestrictU, this, srctrms);
  TaskID sendu = s->AddTask(&Hydro::SendU, this, restu);
  TaskID recvu = s->AddTask(&Hydro::RecvU, this, sendu);
  TaskID prol = s->AddTask(&Hydro::Prolongate, this, recvu);
  TaskID bcs = s->AddTask(&Hydro::ApplyPhysicalBCs, this, prol);
  TaskID c2p = s->AddTask(&Hydro::ConToPrim, this, bcs);
  s->AddTask(&Hydro::NewTimeStep, this, c2p);
  TaskID csend = tl["after_stagen"]->AddTask(&Hydro::ClearSend, this, none);
  tl["after_stagen"]->AddTask(&Hydro::ClearRecv, this, csend);
}
}  // namespace hydro

namespace mhd {
MHD::MHD(MeshBlockPack *pp, ParameterInput *pin) : FluidBase(pp, pin, "mhd") {
  const std::string rs = pin->GetString("mhd", "rsolver");
  // dynamic problems: llf/hlle/hlld; kinematic problems: advect (mhd.cpp:292-326)
  kinematic = pin->GetOrAddString("time", "evolution", "dynamic") == "kinematic";
  if (kinematic) {
    if (rs != "advect") AKMI_FATAL("<mhd> rsolver = '" + rs + "' not implemented for kinematic problems");
    rsolver_method = AKMI_RS_ADVECT; fused = false;
  } else if (rs == "llf") rsolver_method = AKMI_RS_LLF;
  else if (rs == "hlle") rsolver_method = AKMI_RS_HLLE;
  else if (rs == "hlld") rsolver_method = AKMI_RS_HLLD;
  else AKMI_FATAL("<mhd> rsolver = '" + rs + "' not implemented (llf, hlle, hlld)");
  const RegionIndcs &ind = pp->pmesh->mb_indcs;
  const size_t nmb = pp->nmb_thispack;
  const size_t n1 = ind.nx1 + 2*ind.ng, n2 = ind.nx2 > 1 ? ind.nx2 + 2*ind.ng : 1,
               n3 = ind.nx3 > 1 ? ind.nx3 + 2*ind.ng : 1;
  bcc0.Realloc(nmb*3*n3*n2*n1);
  bcc_cells = bcc0.p;
  FaceAlloc(b0, nmb, 1, n3, n2, n1, 1);
  FaceAlloc(b1, nmb, 1, n3, n2, n1, 1);
  if (fused) {
    ws.Realloc(static_cast<size_t>(akmi_stage_workspace_bytes(&pack_c, 1)));
  } else {
    FaceAlloc(uflx, nmb, nvars, n3, n2, n1, 1);                    // mhd.cpp:341-343
    efld.x1e.Realloc(nmb*(n3 + 1)*(n2 + 1)*n1); efld.x2e.Realloc(nmb*(n3 + 1)*n2*(n1 + 1));
    efld.x3e.Realloc(nmb*n3*(n2 + 1)*(n1 + 1));
    for (DvceArray<Real> *a : {&e3x1, &e2x1, &e1x2, &e3x2, &e2x3, &e1x3}) a->Realloc(nmb*n3*n2*n1);
  }
  if (multilevel) {                                                // mhd.cpp:368-380
    const size_t c1 = cpack_c.nx1 + 2*ind.ng, c2 = ind.nx2 > 1 ? cpack_c.nx2 + 2*ind.ng : 1,
                 c3 = ind.nx3 > 1 ? cpack_c.nx3 + 2*ind.ng : 1;
    FaceAlloc(coarse_b0, nmb, 1, c3, c2, c1, 1);
  }
}
MHD::~MHD() {
  bcc0.Free(); FaceFree(b0); FaceFree(b1); FaceFree(uflx); FaceFree(coarse_b0);
  efld.x1e.Free(); efld.x2e.Free(); efld.x3e.Free();
  for (DvceArray<Real> *a : {&e3x1, &e2x1, &e1x2, &e3x2, &e2x3, &e1x3}) a->Free();
}

void MHD::AssembleMHDTasks(std::map<std::string, std::shared_ptr<TaskList>> tl) {
  TaskID none(0);                                                  // mhd_tasks.cpp:38-84
  tl["before_timeintegrator"]->AddTask(&MHD::SaveMHDState, this, none);
  tl["before_stagen"]->AddTask(&MHD::InitRecv, this, none);
  auto &s = tl["stagen"];
  TaskID copyu = s->AddTask(&MHD::CopyCons, this, none);
  TaskID flux = s->AddTask(&MHD::Fluxes, this, copyu);
  TaskID sendf = s->AddTask(&MHD::SendFlux, this, flux);
  TaskID recvf = s->AddTask(&MHD::RecvFlux, this, sendf);
  TaskID rkupdt = s->AddTask(&MHD::RKUpdate, this, recvf);
  TaskID srctrms = s->AddTask(&MHD::MHDSrcTerms, this, rkupdt);
  TaskID restu = s->AddTask(&MHD::RestrictU, this, srctrms);
  TaskID sendu = s->AddTask(&MHD::SendU, this, restu);
  TaskID recvu = s->AddTask(&MHD::RecvU, this, sendu);
  TaskID efld_ = s->AddTask(&MHD::EField, this, recvu);
  TaskID sende = s->AddTask(&MHD::SendE, this, efld_);
  TaskID recve = s->AddTask(&MHD::RecvE, this, sende);
  TaskID ct = s->AddTask(&MHD::CT, this, recve);
  TaskID restb = s->AddTask(&MHD::RestrictB, this, ct);
  TaskID sendb = s->AddTask(&MHD::SendB, this, restb);
  TaskID recvb = s->AddTask(&MHD::RecvB, this, sendb);
  TaskID prol = s->AddTask(&MHD::Prolongate, this, recvb);
  TaskID bcs = s->AddTask(&MHD::ApplyPhysicalBCs, this, prol);
  TaskID c2p = s->AddTask(&MHD::ConToPrim, this, bcs);
  s->AddTask(&MHD::NewTimeStep, this, c2p);
  TaskID csend = tl["after_stagen"]->AddTask(&MHD::ClearSend, this, none);
  tl["after_stagen"]->AddTask(&MHD::ClearRecv, this, csend);
}
}  // namespace mhd

// ---- Driver -----------------------------------------------------------------------------------
Driver::Driver(ParameterInput *pin, Mesh *pmesh) {       // driver.cpp:85-162
  {
    const std::string ev = pin->GetOrAddString("time", "evolution", "dynamic");
    if (ev != "dynamic" && ev != "kinematic")
      AKMI_FATAL("<time> evolution = '" + ev + "' is not on this path (dynamic, kinematic)");
  }
  integrator = pin->GetOrAddString("time", "integrator", "rk2");
  tlim = pin->GetReal("time", "tlim");
  nlim = pin->GetOrAddInteger("time", "nlim", -1);
  for (int q = 0; q < 4; ++q) gam0[q] = gam1[q] = beta[q] = delta[q] = 0.0;
  if (integrator == "rk1") {
    nexp_stages = 1; gam0[0] = 0.0; gam1[0] = 1.0; beta[0] = 1.0;
  } else if (integrator == "rk2") {
    nexp_stages = 2;
    gam0[0] = 0.0; gam1[0] = 1.0; beta[0] = 1.0;
    gam0[1] = 0.5; gam1[1] = 0.5; beta[1] = 0.5;
  } else if (integrator == "rk4") {   // RK4()4[2S], driver.cpp:131-160
    nexp_stages = 4;
    gam0[0] = 0.0; gam1[0] = 1.0; beta[0] = 1.193743905974738;
    gam0[1] = 0.121098479554482; gam1[1] = 0.721781678111411; beta[1] = 0.099279895495783;
    gam0[2] = -3.843833699660025; gam1[2] = 2.121209265338722; beta[2] = 1.131678018054042;
    gam0[3] = 0.546370891121863; gam1[3] = 0.198653035682705; beta[3] = 0.310665766509336;
    delta[0] = 1.0; delta[1] = 0.217683334308543; delta[2] = 1.065841341361089; delta[3] = 0.0;
  } else if (integrator == "rk3") {
    nexp_stages = 3;
    gam0[0] = 0.0; gam1[0] = 1.0; beta[0] = 1.0;
    gam0[1] = 0.25; gam1[1] = 0.75; beta[1] = 0.25;
    gam0[2] = 2.0/3.0; gam1[2] = 1.0/3.0; beta[2] = 2.0/3.0;
  } else {
    AKMI_FATAL("integrator=" + integrator + " not implemented. Valid choices are [rk1,rk2,rk3,rk4].");
  }
  // cycle graph: every physics object on the fused stage, no off-rank neighbours, no levels, and a
  // stream that can be captured (the legacy null stream cannot)
  // auto: 1-D packs only.  Measured (profiles/r02_small_packs.txt): 1-D MHD 109 -> 83 us per cycle, 1-D
  // hydro 68 -> 64 us; from 256^2 upwards the kernels are long enough for the host to stay ahead and
  // the graph's node scheduling costs 5-8 %
  const std::string cg = pin->GetOrAddString("time", "cycle_graph", "auto");
  if (cg != "auto" && cg != "true" && cg != "false") AKMI_FATAL("<time>/cycle_graph = auto, true or false");
  use_graph = cg == "true" || (cg == "auto" && pmesh->one_d);
  if (const char *e = std::getenv("AKMI_CYCLE_GRAPH")) use_graph = std::atoi(e) != 0;
  FluidBase *phys[2] = {pmesh->pmb_pack->phydro, pmesh->pmb_pack->pmhd};
  int nphys = 0;
  for (FluidBase *f : phys) {
    if (!f) continue;
    ++nphys;
    if (!f->fused || f->multilevel || f->kinematic || f->stream == nullptr) use_graph = false;
  }
  if (pmesh->nranks > 1 || nphys != 1 || SelfExchange()) use_graph = false;
  // run-ahead cycles (akmi_host.hpp): the same eligibility, plus no diffusion time steps (they are reduced on the host)
  const std::string ra = pin->GetOrAddString("time", "run_ahead", "auto");
  if (ra != "auto" && ra != "true" && ra != "false") AKMI_FATAL("<time>/run_ahead = auto, true or false");
  run_ahead = ra != "false";
  if (const char *e = std::getenv("AKMI_RUN_AHEAD")) run_ahead = std::atoi(e) != 0;
  for (FluidBase *f : phys) {
    if (!f) continue;
    if (!f->fused || f->multilevel || f->kinematic || f->stream == nullptr || f->has_visc || f->has_cond || f->has_resist)
      run_ahead = false;
  }
  if (pmesh->nranks > 1 || nphys != 1 || SelfExchange() || use_graph) run_ahead = false;
  if (use_graph || run_ahead) {
    d_dt.Realloc(2);
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&h_dt), 5*sizeof(Real)));
    for (FluidBase *f : phys) if (f) f->dt_dev = d_dt.p;
  }
  if (run_ahead) {
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&ra_slot), 6*sizeof(Real)));
    for (hipEvent_t &e : ra_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
}
Driver::~Driver() {
  for (hipEvent_t e : prof_ev) (void)hipEventDestroy(e);
  if (cycle_exec) (void)hipGraphExecDestroy(cycle_exec);
  if (h_dt) (void)hipHostFree(h_dt);
  if (ra_slot) (void)hipHostFree(ra_slot);
  for (hipEvent_t e : ra_ev) if (e) (void)hipEventDestroy(e);
  d_dt.Free();
}

// Mesh::NewTimeStep (mesh.cpp:573-643) for one physics module without diffusion, on the device: st = {dt, time}
__global__ void k_mesh_newdt(Real *__restrict__ dt3, Real *__restrict__ st, Real tlim, Real cfl_no, int multi_d,
                             int three_d, Real *__restrict__ slot) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Real dt = st[0], time = st[1];
  time = time + dt;                                         // driver.cpp:444
  Real dtnew = dt3[0];                                      // hydro_newdt.cpp:121-124
  if (multi_d) dtnew = (dt3[1] < dtnew) ? dt3[1] : dtnew;
  if (three_d) dtnew = (dt3[2] < dtnew) ? dt3[2] : dtnew;
  dt = 2.0*dt;                                              // mesh.cpp:577
  const Real c = cfl_no*dtnew;
  dt = (c < dt) ? c : dt;
  if ((time < tlim) && ((time + dt) > tlim)) dt = tlim - time;
  st[0] = dt; st[1] = time;
  slot[0] = dt; slot[1] = time; slot[2] = dtnew;
  dt3[0] = dt3[1] = dt3[2] = static_cast<Real>(FLT_MAX);    // the next cycle's scan starts from here (no k_init_dt3 launch)
}
void Driver::EnqueueMeshNewDt(FluidBase *f) {
  Mesh *pm = f->pmy_pack->pmesh;
  const int s = static_cast<int>(ra_cycle & 1);
  k_mesh_newdt<<<1, 64, 0, f->stream>>>(f->dt3.p, d_dt.p, tlim, pm->cfl_no, pm->multi_d ? 1 : 0, pm->three_d ? 1 : 0,
                                        ra_slot + 3*s);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ra_ev[s], f->stream));
}

// event pair k = (prof_ev[2k], prof_ev[2k+1]); nothing is recorded while a cycle graph is captured or replayed
void Driver::ProfMark(hipStream_t st) {
  if (!prof_on || capturing || use_graph) return;
  if (prof_used == prof_ev.size()) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    prof_ev.push_back(e);
  }
  HIPCHK(hipEventRecord(prof_ev[prof_used++], st));
}
int Driver::ProfRead(double *ms_total, long long *calls) {
  double tot = 0.0;
  const size_t np = prof_used/2;
  for (size_t k = 0; k < np; ++k) {
    HIPCHK(hipEventSynchronize(prof_ev[2*k + 1]));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, prof_ev[2*k], prof_ev[2*k + 1]));
    tot += ms;
  }
  if (ms_total) *ms_total = tot;
  if (calls) *calls = static_cast<long long>(np);
  prof_used = 0;
  return AKMI_COMPLETE;
}

void Driver::ExecuteTaskList(Mesh *pm, const std::string &tl, int stage) {   // driver.cpp:290-307
  auto &t = pm->pmb_pack->tl_map[tl];
  if (t->Empty()) return;
  t->Reset();
  while (!t->IsComplete())
    if (t->DoAvailable(this, stage) == TaskListStatus::complete) break;
}

void Driver::InitBoundaryValuesAndPrimitives(Mesh *pm) {   // driver.cpp:569-653
  if (pm->pmb_pack->phydro) pm->pmb_pack->phydro->BeginStage();
  if (pm->pmb_pack->pmhd) pm->pmb_pack->pmhd->BeginStage();
  if (auto *ph = pm->pmb_pack->phydro) {
    ph->RestrictU(this, 0);
    ph->SendU(this, 0); ph->RecvU(this, 0); ph->Prolongate(this, 0);
    ph->ApplyPhysicalBCs(this, 0); ph->ConToPrim(this, 0);
  }
  if (auto *pm_ = pm->pmb_pack->pmhd) {
    pm_->RestrictU(this, 0); pm_->RestrictB(this, 0);
    pm_->SendU(this, 0); pm_->RecvU(this, 0); pm_->SendB(this, 0); pm_->RecvB(this, 0);
    pm_->Prolongate(this, 0);
    pm_->ApplyPhysicalBCs(this, 0); pm_->ConToPrim(this, 0);
  }
}

void Driver::Initialize(Mesh *pm) {                        // driver.cpp:314-371
  InitBoundaryValuesAndPrimitives(pm);
  if (pm->pmb_pack->phydro) pm->pmb_pack->phydro->NewTimeStep(this, nexp_stages);
  if (pm->pmb_pack->pmhd) pm->pmb_pack->pmhd->NewTimeStep(this, nexp_stages);
  pm->NewTimeStep(tlim);
  nmb_updated_ = 0;
}

void Driver::RunStages(Mesh *pm) {                         // driver.cpp:398-423
  ExecuteTaskList(pm, "before_timeintegrator", 0);
  for (int stage = 1; stage <= nexp_stages; ++stage) {
    if (pm->pmb_pack->phydro) pm->pmb_pack->phydro->BeginStage();
    if (pm->pmb_pack->pmhd) pm->pmb_pack->pmhd->BeginStage();
    ExecuteTaskList(pm, "before_stagen", stage);
    ExecuteTaskList(pm, "stagen", stage);
    ExecuteTaskList(pm, "after_stagen", stage);
  }
  ExecuteTaskList(pm, "after_timeintegrator", 1);
}

int Driver::Execute(Mesh *pm, int max_cycles) {            // driver.cpp:380-459
  int n = 0;
  if (run_ahead) {
    FluidBase *f = pm->pmb_pack->phydro ? static_cast<FluidBase *>(pm->pmb_pack->phydro)
                                        : static_cast<FluidBase *>(pm->pmb_pack->pmhd);
    h_dt[0] = pm->dt; h_dt[1] = pm->time;
    h_dt[2] = h_dt[3] = h_dt[4] = static_cast<Real>(FLT_MAX);
    HIPCHK(hipMemcpyAsync(d_dt.p, h_dt, 2*sizeof(Real), hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipMemcpyAsync(f->dt3.p, h_dt + 2, 3*sizeof(Real), hipMemcpyHostToDevice, f->stream));
    bool pending = false;                // a cycle is enqueued whose results the host has not read yet
    auto collect = [&](long long cyc) {  // results of cycle `cyc` (counted like ra_cycle): dt of the cycle after it
      const int s = static_cast<int>(cyc & 1);
      HIPCHK(hipEventSynchronize(ra_ev[s]));
      pm->dtold = pm->dt;
      pm->dt = ra_slot[3*s];
      f->dtnew = ra_slot[3*s + 2];
      if (ra_slot[3*s + 1] != pm->time) AKMI_THROW("run-ahead: the device clock left the host clock");
    };
    while ((pm->time < tlim) && (pm->ncycle < nlim || nlim < 0)) {       // pm->time: start of the cycle to enqueue, exact
      if (max_cycles >= 0 && n >= max_cycles) break;
      ra_active = true;                  // (Initialize's NewTimeStep takes the synchronous path)
      // the host's dt is the one of the cycle BEFORE the one being enqueued: every eligible task takes dt from device
      // memory (stage_phase_dt / stage_fused_dt); a task that formed beta*pm->dt on the host would silently use the old
      // value, so the field holds a NaN while the cycle is enqueued -- such a use shows in the first comparison
      const Real dt_host = pm->dt;
      pm->dt = std::numeric_limits<Real>::quiet_NaN();
      RunStages(pm);                     // the kernels read dt from d_dt; NewTimeStep enqueues k_mesh_newdt
      pm->dt = dt_host;
      ra_active = false;
      // dt of the cycle just enqueued is the result of the cycle before it, which has finished by now or will long
      // before the one just enqueued does: the host needs it only here, to advance its clock
      if (pending) collect(ra_cycle - 1);
      ++ra_cycle;
      pending = true;
      pm->time = pm->time + pm->dt;
      pm->ncycle++;
      nmb_updated_ += pm->nmb_total;
      ++n;
    }
    if (pending) collect(ra_cycle - 1);  // the last cycle: dt of the next one, and the clocks compared once more
    HIPCHK(hipStreamSynchronize(f->stream));
    if (pm->pmb_pack->phydro) pm->pmb_pack->phydro->RestoreRegisters();
    if (pm->pmb_pack->pmhd) pm->pmb_pack->pmhd->RestoreRegisters();
    return n;
  }
  while ((pm->time < tlim) && (pm->ncycle < nlim || nlim < 0)) {
    if (max_cycles >= 0 && n >= max_cycles) break;
    if (use_graph) {
      FluidBase *f = pm->pmb_pack->phydro ? static_cast<FluidBase *>(pm->pmb_pack->phydro)
                                          : static_cast<FluidBase *>(pm->pmb_pack->pmhd);
      if (!cycle_exec) {
        // record one cycle; the calls enqueue nothing while the stream is being captured
        hipGraph_t graph;
        HIPCHK(hipStreamBeginCapture(f->stream, hipStreamCaptureModeRelaxed));
        capturing = true;
        RunStages(pm);
        capturing = false;
        HIPCHK(hipStreamEndCapture(f->stream, &graph));
        HIPCHK(hipGraphInstantiate(&cycle_exec, graph, nullptr, nullptr, 0));
        HIPCHK(hipGraphDestroy(graph));
      }
      *h_dt = pm->dt;
      HIPCHK(hipMemcpyAsync(d_dt.p, h_dt, sizeof(Real), hipMemcpyHostToDevice, f->stream));
      HIPCHK(hipGraphLaunch(cycle_exec, f->stream));
      f->FinishNewDtPublic();           // dt3 -> host (the one synchronisation of the cycle)
    } else {
      RunStages(pm);
    }
    pm->time = pm->time + pm->dt;
    pm->ncycle++;
    nmb_updated_ += pm->nmb_total;
    pm->NewTimeStep(tlim);
    ++n;
  }
  // an odd number of out-of-place first stages leaves u0 / b0 in the buffers that used to be u1 / b1: copy
  // back once per call, so that device pointers handed out by akmi_sim_array stay valid across akmi_sim_execute
  if (pm->pmb_pack->phydro) pm->pmb_pack->phydro->RestoreRegisters();
  if (pm->pmb_pack->pmhd) pm->pmb_pack->pmhd->RestoreRegisters();
  return n;
}

// copy_u1 of include/akmi.h: the first stage writes its result into the second register and the
// registers are swapped afterwards (no CopyCons traffic).  The C2P part alone sees swapped pointers
// already; RK4's second register is updated by CopyCons itself; a captured cycle graph has the
// pointers baked in -- those keep the folded copy.
static int CopyFlag(const Driver *d, int stage, int phases) {
  if (stage != 1) return 0;
  static const bool off = std::getenv("AKMI_OUT_OF_PLACE") && std::atoi(std::getenv("AKMI_OUT_OF_PLACE")) == 0;   // A/B switch
  if (d->integrator == "rk4" || d->use_graph || off) return 1;
  return (phases & (AKMI_PHASE_SWEEPS | AKMI_PHASE_EMF_CT)) ? 2 : 0;
}
static bool MergeC2P() {      // A/B switch, profiles/r03_whatif_merge_c2p.txt
  static const bool on = !(std::getenv("AKMI_MERGE_C2P") && std::atoi(std::getenv("AKMI_MERGE_C2P")) == 0);
  return on;
}
template <typename T> static void SwapArr(DvceArray<T> &a, DvceArray<T> &b) { std::swap(a.p, b.p); std::swap(a.n, b.n); }
// hydro: ConsToPrim of the active cells inside the stage kernel (akmi_hydro_stage_w); AKMI_FUSE_C2P=0: A/B switch
static bool FuseC2P() {
  static const bool on = !(std::getenv("AKMI_FUSE_C2P") && std::atoi(std::getenv("AKMI_FUSE_C2P")) == 0);
  return on;
}
// Task-granular path, first stage: CopyCons folded into an out-of-place RKUpdate / CT (akmi_rk_update_oop,
// akmi_mhd_ct_oop), registers swapped afterwards -- no copy traffic.  Not with FOFC (its trial update reads u1/b1
// before RKUpdate), RK4 (CopyCons updates the second register itself), the update-in-the-sweeps option.
// A/B switch AKMI_TASK_OOP=0.
bool FluidBase::OopFirst(const Driver *d, int stage) const {
  static const bool off = std::getenv("AKMI_TASK_OOP") && std::atoi(std::getenv("AKMI_TASK_OOP")) == 0;
  return stage == 1 && !fused && !use_fofc && d->integrator != "rk4" && !off;
}

void FluidBase::RestoreRegisters() {
  if (w_swapped) {             // the primitives back into the buffer akmi_sim_array handed out
    HIPCHK(hipMemcpyAsync(w1.p, w0.p, w0.n*sizeof(Real), hipMemcpyDeviceToDevice, stream));
    SwapArr(w0, w1);
    w_swapped = false;
    if (!u_swapped) HIPCHK(hipStreamSynchronize(stream));
  }
  if (!u_swapped) return;
  // u1 (the creation-time u0 buffer) holds a state nothing reads any more: the first stage of the next cycle
  // overwrites it.  Current state -> that buffer, then the names trade places again.
  HIPCHK(hipMemcpyAsync(u1.p, u0.p, u0.n*sizeof(Real), hipMemcpyDeviceToDevice, stream));
  SwapArr(u0, u1);
  u_swapped = false;
  // akmi_sim_execute returns with every array complete, as it did when each cycle ended in the dt read-back: a
  // caller may read the akmi_sim_array pointers on a stream of its own
  HIPCHK(hipStreamSynchronize(stream));
}

namespace mhd {
void MHD::RestoreRegisters() {
  FluidBase::RestoreRegisters();
  if (!b_swapped) return;
  DvceArray<Real> *cur[3] = {&b0.x1f, &b0.x2f, &b0.x3f}, *old[3] = {&b1.x1f, &b1.x2f, &b1.x3f};
  for (int q = 0; q < 3; ++q) {
    HIPCHK(hipMemcpyAsync(old[q]->p, cur[q]->p, cur[q]->n*sizeof(Real), hipMemcpyDeviceToDevice, stream));
    SwapArr(*cur[q], *old[q]);
  }
  b_swapped = false;
  HIPCHK(hipStreamSynchronize(stream));
}
}  // namespace mhd

// ---- task bodies: one C-ABI call each ------------------------------------------------------------
namespace hydro {
TaskStatus Hydro::CopyCons(Driver *d, int stage) {         // hydro_tasks.cpp:130-152
  if (stage == 1 && !fused && !OopFirst(d, stage)) AKCHK(akmi_copy_cons(&pack_c, u0.p, u1.p, stream));
  if (stage > 1 && d->integrator == "rk4")
    AKCHK(akmi_rk4_copy_cons(&pack_c, d->delta[stage - 1], u0.p, u1.p, stream));
  return TaskStatus::complete;
}
TaskStatus Hydro::Fluxes(Driver *d, int stage) {           // hydro_tasks.cpp:159-201
  if (fused) return TaskStatus::complete;
  if (use_fofc)                                             // hydro_fluxes.cpp:92-101
    AKCHK(akmi_hydro_fluxes_fofc(&pack_c, recon_method, rsolver_method, w0.p, uflx.x1f.p,
                                 uflx.x2f.p, uflx.x3f.p, 0, stream));
  else
    AKCHK(akmi_hydro_fluxes(&pack_c, recon_method, rsolver_method, w0.p, uflx.x1f.p, uflx.x2f.p,
                            uflx.x3f.p, 0, stream));
  AddDiffusionFluxes(uflx, 0);                              // hydro_tasks.cpp:183-189
  if (use_fofc) {                                           // hydro_tasks.cpp:192-194
    AKCHK(akmi_hydro_fofc(&pack_c, d->gam0[stage - 1], d->gam1[stage - 1],
                          d->beta[stage - 1]*pmy_pack->pmesh->dt, w0.p, u0.p, u1.p, uflx.x1f.p,
                          uflx.x2f.p, uflx.x3f.p, 0, fofc.p, nfofc.p, stream));
  }
  return TaskStatus::complete;
}
TaskStatus Hydro::RKUpdate(Driver *d, int stage) {         // hydro_update.cpp:23-83
  Real beta_dt = d->beta[stage - 1]*pmy_pack->pmesh->dt;
  if (fused && peers()) {
    // off-rank neighbours: only the sweeps + update here, so that SendU can post the halo messages
    // before the c2p of the active cells is enqueued
    StagePhase(d, stage, AKMI_PHASE_SWEEPS);
  } else if (fused && !d->use_graph && FuseC2P() && BcsCommuteWithC2P() &&
             akmi_hydro_stage_w_eligible(&pack_c, recon_method, rsolver_method)) {
    // the stage kernel converts the cells it finishes (their new state is in its registers) into the second primitive
    // array; ConToPrim then only has the ghost shell left (after the ghost fill): no pass that reads u0 back
    const int do_dt = (stage == d->nexp_stages);
    const int copy = CopyFlag(d, stage, AKMI_PHASE_ALL);
    if (!w1.p) w1.Realloc(w0.n);
    int wrote = 0;
    d->ProfMark(stream);
    AKCHK(akmi_hydro_stage_w(&pack_c, recon_method, rsolver_method, d->gam0[stage - 1], d->gam1[stage - 1],
                             dt_dev ? d->beta[stage - 1] : beta_dt, dt_dev, copy, w0.p, w1.p, u0.p, u1.p,
                             do_dt ? (d->ra_active ? 2 : 1) : 0, counters.p, dt3.p, ws.p, stream, &wrote));
    d->ProfMark(stream);
    if (copy == 2) { SwapArr(u0, u1); u_swapped = !u_swapped; }
    if (wrote) { SwapArr(w0, w1); w_swapped = !w_swapped; }
    interior_done_ = true; dt_ready_ = do_dt;
    want_ghost_c2p_ = true;
  } else if (fused && !d->use_graph && MergeC2P()) {
    // no off-rank neighbour: ONE ConsToPrim over all cells after the ghost fill (ConToPrim) instead of c2p of the
    // active cells here + c2p of the ghost shell there (thin slabs): 512 blocks of 32^3 1518 -> 1726 Mcell-updates/s
    StagePhase(d, stage, AKMI_PHASE_SWEEPS);
  } else if (fused) {
    int do_dt = (stage == d->nexp_stages);
    const int copy = CopyFlag(d, stage, AKMI_PHASE_ALL);
    d->ProfMark(stream);
    if (dt_dev)
      AKCHK(akmi_hydro_stage_fused_dt(&pack_c, recon_method, rsolver_method, d->gam0[stage - 1],
                                      d->gam1[stage - 1], d->beta[stage - 1], dt_dev, copy, w0.p, u0.p,
                                      u1.p, do_dt, counters.p, dt3.p, ws.p, stream));
    else
    AKCHK(akmi_hydro_stage_fused(&pack_c, recon_method, rsolver_method, d->gam0[stage - 1],
                                 d->gam1[stage - 1], beta_dt, copy, w0.p, u0.p, u1.p, do_dt,
                                 counters.p, dt3.p, ws.p, stream));
    d->ProfMark(stream);
    if (copy == 2) { SwapArr(u0, u1); u_swapped = !u_swapped; }
    interior_done_ = true; dt_ready_ = do_dt;
  } else if (OopFirst(d, stage)) {
    AKCHK(akmi_rk_update_oop(&pack_c, d->gam0[stage - 1], d->gam1[stage - 1], beta_dt, u0.p, u1.p,
                             uflx.x1f.p, uflx.x2f.p, uflx.x3f.p, 0, stream));
    SwapArr(u0, u1); u_swapped = !u_swapped;
  } else {
    AKCHK(akmi_rk_update(&pack_c, d->gam0[stage - 1], d->gam1[stage - 1], beta_dt, u0.p, u1.p,
                         uflx.x1f.p, uflx.x2f.p, uflx.x3f.p, 0, stream));
  }
  return TaskStatus::complete;
}
void Hydro::StagePhase(Driver *d, int stage, int phases) {
  // stage 0 = Driver::InitBoundaryValuesAndPrimitives: only the c2p part may run then
  if (stage < 1 && phases != AKMI_PHASE_C2P) AKMI_FATAL("stage 0 has no RK weights");
  const Real g0 = stage >= 1 ? d->gam0[stage - 1] : 1.0, g1 = stage >= 1 ? d->gam1[stage - 1] : 0.0;
  const Real beta_dt = stage >= 1 ? d->beta[stage - 1]*pmy_pack->pmesh->dt : 0.0;
  const int do_dt = (stage == d->nexp_stages);
  const int copy = CopyFlag(d, stage, phases);
  d->ProfMark(stream);
  if (dt_dev && stage >= 1)
    AKCHK(akmi_hydro_stage_phase_dt(&pack_c, recon_method, rsolver_method, g0, g1, d->beta[stage - 1], dt_dev, copy, w0.p,
                                    u0.p, u1.p, do_dt, counters.p, dt3.p, phases, ws.p, stream));
  else
  AKCHK(akmi_hydro_stage_phase(&pack_c, recon_method, rsolver_method, g0, g1, beta_dt, copy, w0.p,
                               u0.p, u1.p, do_dt, counters.p, dt3.p, phases, ws.p, stream));
  d->ProfMark(stream);
  if (copy == 2) { SwapArr(u0, u1); u_swapped = !u_swapped; }
  if (phases & AKMI_PHASE_C2P) { interior_done_ = true; dt_ready_ = do_dt; }
}
TaskStatus Hydro::SendU(Driver *d, int stage) {            // hydro_tasks.cpp:308-320
  if (multilevel) {
    if (!psmr->cc_map_on)
      AKCHK(akmi_smr_pack_cc(&pack_c, &psmr->smr_c, nvars, u0.p, coarse_u0.p, psmr->buf[0].p, stream));
    psmr->Post(0, stream);
  } else if (pbval) {
    pbval->PackAndSendCC(u0.p, stream);
    // the messages are in flight on the communicator's stream: convert the active cells (they do
    // not depend on the halo) underneath them
    if (fused && peers()) StagePhase(d, stage, AKMI_PHASE_C2P);
  } else {
    GatherU(d, stage);
  }
  return TaskStatus::complete;
}
TaskStatus Hydro::RecvU(Driver *d, int stage) {            // hydro_tasks.cpp:327-339
  if (multilevel) {
    psmr->Wait(0, stream);
    if (psmr->cc_map_on) {
      AKCHK(akmi_smr_cc_copy(&pack_c, nvars, psmr->d_cc_map.p, psmr->cc_np, psmr->cc_tail, u0.p, coarse_u0.p, stream));
    } else {
      AKCHK(akmi_smr_unpack_cc(&pack_c, &psmr->smr_c, nvars, psmr->buf[0].p, u0.p, coarse_u0.p, stream));
      if (psmr->smr_c.direct_same) AKCHK(akmi_bvals_cc_local(&pack_c, nvars, psmr->d_same.p, u0.p, stream));
    }
  } else if (pbval) {
    pbval->RecvAndUnpackCC(u0.p, stream);
  }
  return TaskStatus::complete;
}
TaskStatus Hydro::RecvFlux(Driver *d, int stage) {         // hydro_tasks.cpp:222-232
  if (multilevel) {
    psmr->Wait(1, stream);
    AKCHK(akmi_smr_unpack_flux_cc(&pack_c, &psmr->smr_c, nvars, 0, psmr->buf[1].p, uflx.x1f.p, uflx.x2f.p,
                                  uflx.x3f.p, stream));
  }
  return TaskStatus::complete;
}
TaskStatus Hydro::SendFlux(Driver *d, int stage) {         // hydro_tasks.cpp:206-215
  if (multilevel) {
    AKCHK(akmi_smr_pack_flux_cc(&pack_c, &psmr->smr_c, nvars, 0, uflx.x1f.p, uflx.x2f.p, uflx.x3f.p,
                                psmr->buf[1].p, stream));
    psmr->Post(1, stream);
  }
  return TaskStatus::complete;
}
TaskStatus Hydro::RestrictU(Driver *d, int stage) {        // hydro_tasks.cpp:291-300
  if (multilevel) AKCHK(akmi_restrict_cc_masked(&pack_c, nvars, psmr->smr_c.needs_coarse, u0.p, coarse_u0.p, stream));
  return TaskStatus::complete;
}
TaskStatus Hydro::Prolongate(Driver *d, int stage) {       // hydro_tasks.cpp:381-400
  if (!multilevel) return TaskStatus::complete;
  AKCHK(akmi_smr_fill_coarse_cc(&pack_c, &psmr->smr_c, nvars, u0.p, coarse_u0.p, stream));
  if (!pmy_pack->pmesh->strictly_periodic)               // HydroBCsCoarse: the BC helper on coarse indices
    AKCHK(akmi_hydro_bcs_dirs(&cpack_c, nvars, pmy_pack->pmb->d_bcs.p, pmy_pack->pmb->bc_dirs, nullptr, coarse_u0.p, stream));
  if (pmy_pack->pmesh->prolong_prims) {                   // hydro_tasks.cpp:388-392
    if (!coarse_w0.p) coarse_w0.Realloc(coarse_u0.n);
    AKCHK(akmi_smr_c2p_coarse(&pack_c, &psmr->smr_c, nvars, coarse_u0.p, nullptr, nullptr, nullptr, coarse_w0.p, stream));
    AKCHK(akmi_smr_prolong_cc(&pack_c, &psmr->smr_c, nvars, coarse_w0.p, w0.p, stream));
    AKCHK(akmi_smr_p2c_fine(&pack_c, &psmr->smr_c, nvars, w0.p, nullptr, nullptr, nullptr, u0.p, stream));
    return TaskStatus::complete;
  }
  AKCHK(akmi_smr_prolong_cc(&pack_c, &psmr->smr_c, nvars, coarse_u0.p, u0.p, stream));
  return TaskStatus::complete;
}
TaskStatus Hydro::ApplyPhysicalBCs(Driver *d, int stage) { // hydro_tasks.cpp:357-375
  if (pmy_pack->pmesh->strictly_periodic) return TaskStatus::complete;
  if (u_bcs_done_) { u_bcs_done_ = false; return TaskStatus::complete; }      // applied by the gather of SendU
  AKCHK(akmi_hydro_bcs_dirs(&pack_c, nvars, pmy_pack->pmb->d_bcs.p, pmy_pack->pmb->bc_dirs, nullptr, u0.p, stream));
  return TaskStatus::complete;
}
TaskStatus Hydro::ConToPrim(Driver *d, int stage) {        // hydro_tasks.cpp:404-412
  const RegionIndcs &ind = pmy_pack->pmesh->mb_indcs;
  const int n1 = ind.nx1 + 2*ind.ng, n2 = ind.nx2 > 1 ? ind.nx2 + 2*ind.ng : 1,
            n3 = ind.nx3 > 1 ? ind.nx3 + 2*ind.ng : 1;
  if (fused && interior_done_) {
    interior_done_ = false;
    if (!shell_done_) AKCHK(akmi_hydro_c2p_shell(&pack_c, u0.p, w0.p, counters.p, stream));
    shell_done_ = false;
  } else if (fused) {
    int do_dt = (stage == d->nexp_stages);
    d->ProfMark(stream);
    // (run-ahead cycles: k_mesh_newdt leaves the minima reset for the next cycle)
    AKCHK(akmi_hydro_c2p_newdt(&pack_c, u0.p, w0.p, do_dt ? ((dt3_reset_ || d->ra_active) ? 2 : 1) : 0, counters.p, dt3.p,
                               stream));
    dt3_reset_ = false;
    d->ProfMark(stream);
    dt_ready_ = do_dt;
  } else if (multilevel && !kinematic) {
    // refined meshes (task-granular chain): the last conversion carries the CFL scan of NewTimeStep along; the others go
    // through the same entry (large packs: its two-cells-per-thread kernel, 7 % faster than akmi_hydro_c2p's)
    const int do_dt = (stage == d->nexp_stages);
    AKCHK(akmi_hydro_c2p_newdt(&pack_c, u0.p, w0.p, do_dt, counters.p, dt3.p, stream));
    dt_ready_ = do_dt;
  } else {
    AKCHK(akmi_hydro_c2p(&pack_c, u0.p, w0.p, 0, n1 - 1, 0, n2 - 1, 0, n3 - 1, counters.p, stream));
  }
  return TaskStatus::complete;
}
TaskStatus Hydro::NewTimeStep(Driver *d, int stage) {      // hydro_newdt.cpp:30-139
  if (stage != d->nexp_stages) return TaskStatus::complete;
  if (kinematic) AKCHK(akmi_kinematic_newdt(&pack_c, w0.p, dt3.p, stream));      // hydro_newdt.cpp:55-72
  else if (!dt_ready_) AKCHK(akmi_hydro_newdt(&pack_c, w0.p, dt3.p, stream));
  dt_ready_ = false;
  if (d->capturing) return TaskStatus::complete;   // the Driver reads dt3 after the graph launch
  if (d->ra_active) { d->EnqueueMeshNewDt(this); return TaskStatus::complete; }
  FinishNewDt();
  DiffusionNewDt();
  return TaskStatus::complete;
}
}  // namespace hydro

namespace mhd {
TaskStatus MHD::CopyCons(Driver *d, int stage) {           // mhd_tasks.cpp:162-170
  if (stage == 1 && !fused && !OopFirst(d, stage)) {
    AKCHK(akmi_copy_cons(&pack_c, u0.p, u1.p, stream));
    HIPCHK(hipMemcpyAsync(b1.x1f.p, b0.x1f.p, sizeof(Real)*b0.x1f.n, hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemcpyAsync(b1.x2f.p, b0.x2f.p, sizeof(Real)*b0.x2f.n, hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemcpyAsync(b1.x3f.p, b0.x3f.p, sizeof(Real)*b0.x3f.n, hipMemcpyDeviceToDevice, stream));
  }
  return TaskStatus::complete;
}
TaskStatus MHD::Fluxes(Driver *d, int stage) {             // mhd_tasks.cpp:177-216
  if (fused) return TaskStatus::complete;
  if (use_fofc)                                             // mhd_fluxes.cpp:100-105
    AKCHK(akmi_mhd_fluxes_fofc(&pack_c, recon_method, rsolver_method, w0.p, bcc0.p, b0.x1f.p,
                               b0.x2f.p, b0.x3f.p, uflx.x1f.p, uflx.x2f.p, uflx.x3f.p, e3x1.p,
                               e2x1.p, e1x2.p, e3x2.p, e2x3.p, e1x3.p, stream));
  else
    AKCHK(akmi_mhd_fluxes(&pack_c, recon_method, rsolver_method, w0.p, bcc0.p, b0.x1f.p, b0.x2f.p,
                          b0.x3f.p, uflx.x1f.p, uflx.x2f.p, uflx.x3f.p, e3x1.p, e2x1.p, e1x2.p,
                          e3x2.p, e2x3.p, e1x3.p, stream));
  AddDiffusionFluxes(uflx, 1);                              // mhd_tasks.cpp:198-203
  if (has_resist && eta_ohm != 0.0 && peos->eos_data.is_ideal)   // mhd_tasks.cpp:204-206
    AKCHK(akmi_resistive_fluxes(&pack_c, eta_ohm, b0.x1f.p, b0.x2f.p, b0.x3f.p, uflx.x1f.p,
                                uflx.x2f.p, uflx.x3f.p, stream));
  if (has_resist && eta_ad != 0.0 && peos->eos_data.is_ideal)    // resistivity.cpp:67-69
    AKCHK(akmi_ambipolar_fluxes(&pack_c, eta_ad, bcc0.p, b0.x1f.p, b0.x2f.p, b0.x3f.p, uflx.x1f.p,
                                uflx.x2f.p, uflx.x3f.p, stream));
  if (use_fofc) {                                           // mhd_tasks.cpp:209-211
    AKCHK(akmi_mhd_fofc(&pack_c, d->gam0[stage - 1], d->gam1[stage - 1],
                        d->beta[stage - 1]*pmy_pack->pmesh->dt, w0.p, bcc0.p, b0.x1f.p, b0.x2f.p,
                        b0.x3f.p, b1.x1f.p, b1.x2f.p, b1.x3f.p, u0.p, u1.p, uflx.x1f.p, uflx.x2f.p,
                        uflx.x3f.p, e3x1.p, e2x1.p, e1x2.p, e3x2.p, e2x3.p, e1x3.p, fofc.p, nfofc.p,
                        stream));
  }
  return TaskStatus::complete;
}
TaskStatus MHD::RKUpdate(Driver *d, int stage) {           // mhd_update.cpp:24-84
  Real beta_dt = d->beta[stage - 1]*pmy_pack->pmesh->dt;
  if (fused && peers()) {
    StagePhase(d, stage, AKMI_PHASE_SWEEPS);
  } else if (fused && !d->use_graph && MergeC2P()) {
    StagePhase(d, stage, AKMI_PHASE_SWEEPS | AKMI_PHASE_EMF_CT);      // see Hydro::RKUpdate
  } else if (fused) {
    int do_dt = (stage == d->nexp_stages);
    const int copy = CopyFlag(d, stage, AKMI_PHASE_ALL);
    d->ProfMark(stream);
    if (dt_dev)
      AKCHK(akmi_mhd_stage_fused_dt(&pack_c, recon_method, rsolver_method, d->gam0[stage - 1],
                                    d->gam1[stage - 1], d->beta[stage - 1], dt_dev, copy, w0.p, bcc0.p,
                                    u0.p, u1.p, b0.x1f.p, b0.x2f.p, b0.x3f.p, b1.x1f.p, b1.x2f.p, b1.x3f.p,
                                    do_dt, counters.p, dt3.p, ws.p, stream));
    else
    AKCHK(akmi_mhd_stage_fused(&pack_c, recon_method, rsolver_method, d->gam0[stage - 1],
                               d->gam1[stage - 1], beta_dt, copy, w0.p, bcc0.p, u0.p, u1.p,
                               b0.x1f.p, b0.x2f.p, b0.x3f.p, b1.x1f.p, b1.x2f.p, b1.x3f.p, do_dt,
                               counters.p, dt3.p, ws.p, stream));
    d->ProfMark(stream);
    if (copy == 2) {
      SwapArr(u0, u1); u_swapped = !u_swapped;
      SwapArr(b0.x1f, b1.x1f); SwapArr(b0.x2f, b1.x2f); SwapArr(b0.x3f, b1.x3f); b_swapped = !b_swapped;
    }
    interior_done_ = true; dt_ready_ = do_dt;
  } else if (OopFirst(d, stage)) {
    AKCHK(akmi_rk_update_oop(&pack_c, d->gam0[stage - 1], d->gam1[stage - 1], beta_dt, u0.p, u1.p,
                             uflx.x1f.p, uflx.x2f.p, uflx.x3f.p, 1, stream));
    SwapArr(u0, u1); u_swapped = !u_swapped;
  } else {
    AKCHK(akmi_rk_update(&pack_c, d->gam0[stage - 1], d->gam1[stage - 1], beta_dt, u0.p, u1.p,
                         uflx.x1f.p, uflx.x2f.p, uflx.x3f.p, 1, stream));
  }
  return TaskStatus::complete;
}
void MHD::StagePhase(Driver *d, int stage, int phases) {
  if (stage < 1 && phases != AKMI_PHASE_C2P) AKMI_FATAL("stage 0 has no RK weights");
  const Real g0 = stage >= 1 ? d->gam0[stage - 1] : 1.0, g1 = stage >= 1 ? d->gam1[stage - 1] : 0.0;
  const Real beta_dt = stage >= 1 ? d->beta[stage - 1]*pmy_pack->pmesh->dt : 0.0;
  const int do_dt = (stage == d->nexp_stages);
  const int copy = CopyFlag(d, stage, phases);
  d->ProfMark(stream);
  if (dt_dev && stage >= 1)
    AKCHK(akmi_mhd_stage_phase_dt(&pack_c, recon_method, rsolver_method, g0, g1, d->beta[stage - 1], dt_dev, copy, w0.p,
                                  bcc0.p, u0.p, u1.p, b0.x1f.p, b0.x2f.p, b0.x3f.p, b1.x1f.p, b1.x2f.p,
                                  b1.x3f.p, do_dt, counters.p, dt3.p, phases, ws.p, stream));
  else
  AKCHK(akmi_mhd_stage_phase(&pack_c, recon_method, rsolver_method, g0, g1, beta_dt, copy, w0.p,
                             bcc0.p, u0.p, u1.p, b0.x1f.p, b0.x2f.p, b0.x3f.p, b1.x1f.p, b1.x2f.p,
                             b1.x3f.p, do_dt, counters.p, dt3.p, phases, ws.p, stream));
  d->ProfMark(stream);
  if (copy == 2) {
    if (phases & AKMI_PHASE_SWEEPS) { SwapArr(u0, u1); u_swapped = !u_swapped; }
    if (phases & AKMI_PHASE_EMF_CT) {
      SwapArr(b0.x1f, b1.x1f); SwapArr(b0.x2f, b1.x2f); SwapArr(b0.x3f, b1.x3f); b_swapped = !b_swapped;
    }
  }
  if (phases & AKMI_PHASE_C2P) { interior_done_ = true; dt_ready_ = do_dt; }
}
// Off-rank neighbours and the fused stage: sweeps+update -> pack+send U -> CornerE+CT -> pack+send B
// -> c2p of the active cells (+dt) -> wait/unpack U, B -> BCs -> c2p of the ghost shell: the reference's
// task order (mhd_tasks.cpp:48-75) with the transfers underneath the kernels that do not need them
TaskStatus MHD::SendU(Driver *d, int stage) {
  if (multilevel) {
    if (!psmr->cc_map_on)
      AKCHK(akmi_smr_pack_cc(&pack_c, &psmr->smr_c, nvars, u0.p, coarse_u0.p, psmr->buf[0].p, stream));
    psmr->Post(0, stream);
  } else if (pbval)
    pbval->PackAndSendCC(u0.p, stream);
  else
    GatherU(d, stage);
  return TaskStatus::complete;
}
TaskStatus MHD::RecvU(Driver *d, int stage) {
  if (multilevel) {
    psmr->Wait(0, stream);
    if (psmr->cc_map_on) {
      AKCHK(akmi_smr_cc_copy(&pack_c, nvars, psmr->d_cc_map.p, psmr->cc_np, psmr->cc_tail, u0.p, coarse_u0.p, stream));
    } else {
      AKCHK(akmi_smr_unpack_cc(&pack_c, &psmr->smr_c, nvars, psmr->buf[0].p, u0.p, coarse_u0.p, stream));
      if (psmr->smr_c.direct_same) AKCHK(akmi_bvals_cc_local(&pack_c, nvars, psmr->d_same.p, u0.p, stream));
    }
  } else if (pbval && !(fused && peers())) {
    pbval->RecvAndUnpackCC(u0.p, stream);                                     // else: in RecvB
  }
  return TaskStatus::complete;
}
TaskStatus MHD::RecvFlux(Driver *d, int stage) {           // mhd_tasks.cpp:240-250
  if (multilevel) {
    psmr->Wait(1, stream);
    AKCHK(akmi_smr_unpack_flux_cc(&pack_c, &psmr->smr_c, nvars, 1, psmr->buf[1].p, uflx.x1f.p, uflx.x2f.p,
                                  uflx.x3f.p, stream));
  }
  return TaskStatus::complete;
}
TaskStatus MHD::RecvE(Driver *d, int stage) {              // mhd_tasks.cpp:410-417
  if (multilevel) {
    psmr->Wait(3, stream);
    AKCHK(akmi_smr_unpack_emf(&pack_c, &psmr->smr_c, psmr->d_nflx.p, psmr->buf[3].p, efld.x1e.p, efld.x2e.p,
                              efld.x3e.p, stream));
  }
  return TaskStatus::complete;
}
TaskStatus MHD::RecvB(Driver *d, int stage) {
  if (multilevel) {
    psmr->Wait(2, stream);
    if (psmr->fc_map_on)
      AKCHK(akmi_smr_fc_copy(&pack_c, psmr->d_fc_map[0].p, psmr->fc_np[0], psmr->fc_tail[0], static_cast<long long>(psmr->buf[2].n),
                             b0.x1f.p, b0.x2f.p, b0.x3f.p, coarse_b0.x1f.p, coarse_b0.x2f.p, coarse_b0.x3f.p,
                             psmr->buf[2].p, stream));
    else
    AKCHK(akmi_smr_unpack_fc(&pack_c, &psmr->smr_c, psmr->buf[2].p, b0.x1f.p, b0.x2f.p, b0.x3f.p,
                             coarse_b0.x1f.p, coarse_b0.x2f.p, coarse_b0.x3f.p, stream));
    return TaskStatus::complete;
  }
  if (!pbval) return TaskStatus::complete;
  if (fused && peers()) pbval->RecvAndUnpackCC(u0.p, stream);
  pbval->RecvAndUnpackFC(b0, stream);
  return TaskStatus::complete;
}
TaskStatus MHD::SendFlux(Driver *d, int stage) {           // mhd_tasks.cpp:225-233
  if (multilevel) {
    AKCHK(akmi_smr_pack_flux_cc(&pack_c, &psmr->smr_c, nvars, 1, uflx.x1f.p, uflx.x2f.p, uflx.x3f.p,
                                psmr->buf[1].p, stream));
    psmr->Post(1, stream);
  }
  return TaskStatus::complete;
}
TaskStatus MHD::RestrictU(Driver *d, int stage) {          // mhd_tasks.cpp:315-322
  if (multilevel) AKCHK(akmi_restrict_cc_masked(&pack_c, nvars, psmr->smr_c.needs_coarse, u0.p, coarse_u0.p, stream));
  return TaskStatus::complete;
}
TaskStatus MHD::RestrictB(Driver *d, int stage) {          // mhd_tasks.cpp:691-697
  if (multilevel)
    AKCHK(akmi_restrict_fc_masked(&pack_c, psmr->smr_c.needs_coarse, b0.x1f.p, b0.x2f.p, b0.x3f.p, coarse_b0.x1f.p,
                                  coarse_b0.x2f.p, coarse_b0.x3f.p, stream));
  return TaskStatus::complete;
}
// SendE + RecvE (mhd_tasks.cpp:402-417).  Uniform mesh: every copy of a shared edge EMF is computed by
// the same deterministic kernel from identical inputs and the reference's sum and average return the
// value itself (2a/2 and ((2a+a)+a)/4 are exact), so nothing is exchanged; with levels this is the
// flux correction of the field
TaskStatus MHD::SendE(Driver *d, int stage) {
  if (multilevel) {
    AKCHK(akmi_smr_pack_emf(&pack_c, &psmr->smr_c, efld.x1e.p, efld.x2e.p, efld.x3e.p, psmr->buf[3].p, stream));
    psmr->Post(3, stream);
  }
  return TaskStatus::complete;
}
TaskStatus MHD::Prolongate(Driver *d, int stage) {         // mhd_tasks.cpp:527-552
  if (!multilevel) return TaskStatus::complete;
  const akmi_smr *t = &psmr->smr_c;
  AKCHK(akmi_smr_fill_coarse_cc(&pack_c, t, nvars, u0.p, coarse_u0.p, stream));
  AKCHK(akmi_smr_fill_coarse_fc(&pack_c, t, b0.x1f.p, b0.x2f.p, b0.x3f.p, coarse_b0.x1f.p, coarse_b0.x2f.p,
                                coarse_b0.x3f.p, stream));
  if (!pmy_pack->pmesh->strictly_periodic) {
    AKCHK(akmi_hydro_bcs_dirs(&cpack_c, nvars, pmy_pack->pmb->d_bcs.p, pmy_pack->pmb->bc_dirs, nullptr, coarse_u0.p, stream));
    AKCHK(akmi_bfield_bcs_dirs(&cpack_c, pmy_pack->pmb->d_bcs.p, pmy_pack->pmb->bc_dirs, nullptr, coarse_b0.x1f.p, coarse_b0.x2f.p, coarse_b0.x3f.p,
                          stream));
  }
  if (pmy_pack->pmesh->prolong_prims) {                   // mhd_tasks.cpp:539-544
    if (!coarse_w0.p) coarse_w0.Realloc(coarse_u0.n);
    AKCHK(akmi_smr_c2p_coarse(&pack_c, t, nvars, coarse_u0.p, coarse_b0.x1f.p, coarse_b0.x2f.p, coarse_b0.x3f.p,
                              coarse_w0.p, stream));
    AKCHK(akmi_smr_prolong_cc(&pack_c, t, nvars, coarse_w0.p, w0.p, stream));
    AKCHK(akmi_smr_prolong_fc(&pack_c, t, coarse_b0.x1f.p, coarse_b0.x2f.p, coarse_b0.x3f.p, b0.x1f.p, b0.x2f.p,
                              b0.x3f.p, stream));
    AKCHK(akmi_smr_p2c_fine(&pack_c, t, nvars, w0.p, b0.x1f.p, b0.x2f.p, b0.x3f.p, u0.p, stream));
    return TaskStatus::complete;
  }
  AKCHK(akmi_smr_prolong_cc(&pack_c, t, nvars, coarse_u0.p, u0.p, stream));
  AKCHK(akmi_smr_prolong_fc(&pack_c, t, coarse_b0.x1f.p, coarse_b0.x2f.p, coarse_b0.x3f.p, b0.x1f.p, b0.x2f.p,
                            b0.x3f.p, stream));
  return TaskStatus::complete;
}
TaskStatus MHD::EField(Driver *d, int stage) {             // mhd_corner_e.cpp:26-417
  if (!fused)
    AKCHK(akmi_mhd_corner_e(&pack_c, w0.p, bcc0.p, e3x1.p, e2x1.p, e1x2.p, e3x2.p, e2x3.p, e1x3.p,
                            uflx.x1f.p, uflx.x2f.p, uflx.x3f.p, efld.x1e.p, efld.x2e.p, efld.x3e.p,
                            stream));
  if (!fused && has_resist && eta_ohm != 0.0)               // mhd_tasks.cpp:381-383
    AKCHK(akmi_resistive_emfs(&pack_c, eta_ohm, b0.x1f.p, b0.x2f.p, b0.x3f.p, efld.x1e.p, efld.x2e.p,
                              efld.x3e.p, stream));
  if (!fused && has_resist && eta_ad != 0.0)                // resistivity.cpp:52-54
    AKCHK(akmi_ambipolar_emfs(&pack_c, eta_ad, bcc0.p, b0.x1f.p, b0.x2f.p, b0.x3f.p, efld.x1e.p,
                              efld.x2e.p, efld.x3e.p, stream));
  return TaskStatus::complete;
}
TaskStatus MHD::CT(Driver *d, int stage) {                 // mhd_ct.cpp:23-80
  if (fused && peers()) StagePhase(d, stage, AKMI_PHASE_EMF_CT);
  if (!fused && OopFirst(d, stage)) {
    AKCHK(akmi_mhd_ct_oop(&pack_c, d->gam0[stage - 1], d->gam1[stage - 1],
                          d->beta[stage - 1]*pmy_pack->pmesh->dt, efld.x1e.p, efld.x2e.p, efld.x3e.p,
                          b0.x1f.p, b0.x2f.p, b0.x3f.p, b1.x1f.p, b1.x2f.p, b1.x3f.p, stream));
    SwapArr(b0.x1f, b1.x1f); SwapArr(b0.x2f, b1.x2f); SwapArr(b0.x3f, b1.x3f); b_swapped = !b_swapped;
  } else if (!fused) {
    AKCHK(akmi_mhd_ct(&pack_c, d->gam0[stage - 1], d->gam1[stage - 1],
                      d->beta[stage - 1]*pmy_pack->pmesh->dt, efld.x1e.p, efld.x2e.p, efld.x3e.p,
                      b0.x1f.p, b0.x2f.p, b0.x3f.p, b1.x1f.p, b1.x2f.p, b1.x3f.p, stream));
  }
  return TaskStatus::complete;
}
TaskStatus MHD::SendB(Driver *d, int stage) {
  if (multilevel) {
    if (psmr->fc_map_on)
      AKCHK(akmi_smr_fc_copy(&pack_c, psmr->d_fc_map[1].p, psmr->fc_np[1], psmr->fc_tail[1], static_cast<long long>(psmr->buf[2].n),
                             b0.x1f.p, b0.x2f.p, b0.x3f.p, coarse_b0.x1f.p, coarse_b0.x2f.p, coarse_b0.x3f.p,
                             psmr->buf[2].p, stream));
    else
    AKCHK(akmi_smr_pack_fc(&pack_c, &psmr->smr_c, b0.x1f.p, b0.x2f.p, b0.x3f.p, coarse_b0.x1f.p,
                           coarse_b0.x2f.p, coarse_b0.x3f.p, psmr->buf[2].p, stream));
    psmr->Post(2, stream);
  } else if (pbval) {
    pbval->PackAndSendFC(b0, stream);
    if (fused && peers()) StagePhase(d, stage, AKMI_PHASE_C2P);
  } else if (FoldBCs() && !pmy_pack->pmesh->strictly_periodic) {
    AKCHK(akmi_bvals_fc_local_bcs(&pack_c, pmy_pack->pmb->d_nghbr.p, pmy_pack->pmb->d_bcs.p, nullptr, b0.x1f.p, b0.x2f.p,
                                  b0.x3f.p, stream));
    b_bcs_done_ = true;
  } else {
    AKCHK(akmi_bvals_fc_local(&pack_c, pmy_pack->pmb->d_nghbr.p, b0.x1f.p, b0.x2f.p, b0.x3f.p, stream));
  }
  return TaskStatus::complete;
}
TaskStatus MHD::ApplyPhysicalBCs(Driver *d, int stage) {   // mhd_tasks.cpp:501-520
  if (pmy_pack->pmesh->strictly_periodic) return TaskStatus::complete;
  if (!u_bcs_done_)
    AKCHK(akmi_hydro_bcs_dirs(&pack_c, nvars, pmy_pack->pmb->d_bcs.p, pmy_pack->pmb->bc_dirs, nullptr, u0.p, stream));
  if (!b_bcs_done_)
    AKCHK(akmi_bfield_bcs_dirs(&pack_c, pmy_pack->pmb->d_bcs.p, pmy_pack->pmb->bc_dirs, nullptr, b0.x1f.p, b0.x2f.p, b0.x3f.p, stream));
  u_bcs_done_ = b_bcs_done_ = false;
  return TaskStatus::complete;
}
TaskStatus MHD::ConToPrim(Driver *d, int stage) {
  const RegionIndcs &ind = pmy_pack->pmesh->mb_indcs;
  const int n1 = ind.nx1 + 2*ind.ng, n2 = ind.nx2 > 1 ? ind.nx2 + 2*ind.ng : 1,
            n3 = ind.nx3 > 1 ? ind.nx3 + 2*ind.ng : 1;
  if (fused && interior_done_) {
    interior_done_ = false;
    AKCHK(akmi_mhd_c2p_shell(&pack_c, u0.p, b0.x1f.p, b0.x2f.p, b0.x3f.p, w0.p, bcc0.p, counters.p, stream));
  } else if (fused) {
    int do_dt = (stage == d->nexp_stages);
    d->ProfMark(stream);
    AKCHK(akmi_mhd_c2p_newdt(&pack_c, u0.p, b0.x1f.p, b0.x2f.p, b0.x3f.p, w0.p, bcc0.p,
                             do_dt ? ((dt3_reset_ || d->ra_active) ? 2 : 1) : 0,
                             counters.p, dt3.p, stream));
    dt3_reset_ = false;
    d->ProfMark(stream);
    dt_ready_ = do_dt;
  } else if (multilevel && !kinematic) {                   // (see Hydro::ConToPrim)
    const int do_dt = (stage == d->nexp_stages);
    AKCHK(akmi_mhd_c2p_newdt(&pack_c, u0.p, b0.x1f.p, b0.x2f.p, b0.x3f.p, w0.p, bcc0.p, do_dt, counters.p, dt3.p,
                             stream));
    dt_ready_ = do_dt;
  } else {
    AKCHK(akmi_mhd_c2p(&pack_c, u0.p, b0.x1f.p, b0.x2f.p, b0.x3f.p, w0.p, bcc0.p, 0, n1 - 1, 0, n2 - 1,
                       0, n3 - 1, counters.p, stream));
  }
  return TaskStatus::complete;
}
TaskStatus MHD::NewTimeStep(Driver *d, int stage) {        // mhd_newdt.cpp:31-174
  if (stage != d->nexp_stages) return TaskStatus::complete;
  if (kinematic) AKCHK(akmi_kinematic_newdt(&pack_c, w0.p, dt3.p, stream));      // mhd_newdt.cpp:56-73
  else if (!dt_ready_) AKCHK(akmi_mhd_newdt(&pack_c, w0.p, bcc0.p, dt3.p, stream));
  dt_ready_ = false;
  if (d->capturing) return TaskStatus::complete;   // the Driver reads dt3 after the graph launch
  if (d->ra_active) { d->EnqueueMeshNewDt(this); return TaskStatus::complete; }
  FinishNewDt();
  DiffusionNewDt();
  return TaskStatus::complete;
}
}  // namespace mhd

// ---- the simulation object behind akmi_sim_* --------------------------------------------------------
struct Sim {
  ParameterInput pin;
  Mesh *pmesh = nullptr;
  Driver *pdriver = nullptr;
  hipStream_t own_stream = nullptr;     // the caller passed the null stream, which cannot be captured
  ~Sim() { delete pdriver; delete pmesh; if (own_stream) (void)hipStreamDestroy(own_stream); }
  // work the caller enqueued elsewhere (initial conditions written on the null stream) comes first
  void Enter() { if (own_stream) HIPCHK(hipDeviceSynchronize()); }
};

}  // namespace host
}  // namespace akmi

namespace akmi { void set_error(const char *fmt, ...); }
using namespace akmi::host;

extern "C" {

void *akmi_sim_create(const char *deck_text, void *stream) {
  Sim *s = nullptr;
  try {
    s = new Sim;
    s->pin.LoadFromString(deck_text);
    s->pmesh = new Mesh(&s->pin, Comm::World().rank, Comm::World().nranks);
    s->pmesh->pmb_pack->AddPhysics(&s->pin);
    if (!stream) {
      // a stream of our own (the null stream cannot be captured into a graph), created with the DEFAULT flags: the same
      // kernels run measurably slower on a hipStreamNonBlocking stream on this stack -- hydro 256^3 with ConsToPrim inside the
      // stage kernel 6 880-6 930 against 7 900-8 170 Mcell-updates/s, the null stream 7 790-7 930 (profiles/r06_stream_kind.txt;
      // AKMI_STREAM_NONBLOCKING=1 brings the old kind back for A/B runs)
      static const bool nb = std::getenv("AKMI_STREAM_NONBLOCKING") && std::atoi(std::getenv("AKMI_STREAM_NONBLOCKING")) != 0;
      HIPCHK(hipStreamCreateWithFlags(&s->own_stream, nb ? hipStreamNonBlocking : hipStreamDefault));
      stream = s->own_stream;
    }
    if (auto *ph = s->pmesh->pmb_pack->phydro) ph->stream = (hipStream_t)stream;
    if (auto *pm = s->pmesh->pmb_pack->pmhd) pm->stream = (hipStream_t)stream;
    return s;
  } catch (...) {
    NoteException("akmi_sim_create");
    try { delete s; } catch (...) {}      // what the half-built objects own is released by their destructors
    return nullptr;
  }
}

/* the Driver reads <time>/tlim, which the linear-wave problem generator rescales: create it
 * after the initial conditions have been uploaded */
int akmi_sim_initialize(void *h, double tlim_override) {
  AKMI_C_ENTRY("akmi_sim_initialize", AKMI_FAIL,
    Sim *s = static_cast<Sim *>(h);
    s->Enter();
    if (tlim_override > 0.0) s->pin.SetReal("time", "tlim", tlim_override);
    delete s->pdriver;
    s->pdriver = nullptr;
    s->pdriver = new Driver(&s->pin, s->pmesh);
    s->pdriver->Initialize(s->pmesh);
    return AKMI_COMPLETE;
  )
}

int akmi_sim_execute(void *h, int max_cycles) {
  AKMI_C_ENTRY("akmi_sim_execute", AKMI_FAIL,
    Sim *s = static_cast<Sim *>(h);
    if (!s->pdriver) { akmi::set_error("akmi_sim_execute: call akmi_sim_initialize first"); return AKMI_FAIL; }
    s->Enter();
    return s->pdriver->Execute(s->pmesh, max_cycles);
  )
}

/* live timing of the fused-stage launch group: on != 0 starts recording a HIP event pair on the launch stream
 * around every akmi_*_stage_fused / akmi_*_stage_phase call; akmi_sim_profile_read waits for the recorded
 * events, returns their summed duration and count, and clears the record */
int akmi_sim_profile(void *h, int on) {
  Sim *s = static_cast<Sim *>(h);
  if (!s->pdriver) { akmi::set_error("akmi_sim_profile: call akmi_sim_initialize first"); return AKMI_FAIL; }
  s->pdriver->prof_on = on != 0;
  if (on) s->pdriver->prof_used = 0;
  return AKMI_COMPLETE;
}
int akmi_sim_profile_read(void *h, double *ms_total, long long *calls) {
  AKMI_C_ENTRY("akmi_sim_profile_read", AKMI_FAIL,
    Sim *s = static_cast<Sim *>(h);
    if (!s->pdriver) { akmi::set_error("akmi_sim_profile_read: call akmi_sim_initialize first"); return AKMI_FAIL; }
    return s->pdriver->ProfRead(ms_total, calls);
  )
}

void akmi_sim_destroy(void *h) { try { delete static_cast<Sim *>(h); } catch (...) { NoteException("akmi_sim_destroy"); } }

double akmi_sim_time(void *h) { return static_cast<Sim *>(h)->pmesh->time; }
double akmi_sim_dt(void *h) { return static_cast<Sim *>(h)->pmesh->dt; }
double akmi_sim_tlim(void *h) { Sim *s = static_cast<Sim *>(h); return s->pdriver ? s->pdriver->tlim : s->pin.GetReal("time", "tlim"); }
int akmi_sim_ncycle(void *h) { return static_cast<Sim *>(h)->pmesh->ncycle; }
int akmi_sim_nmb(void *h) { return static_cast<Sim *>(h)->pmesh->nmb_total; }

/* device pointer + element count of a named array: u0 w0 u1 bcc0 b0x1f b0x2f b0x3f b1x1f b1x2f
 * b1x3f dx ; lloc (host int[nmb][3]) via akmi_sim_lloc */
void *akmi_sim_array(void *h, const char *name, long long *count) {
  AKMI_C_ENTRY("akmi_sim_array", nullptr,
  Sim *s = static_cast<Sim *>(h);
  MeshBlockPack *pk = s->pmesh->pmb_pack;
  FluidBase *f = pk->phydro ? static_cast<FluidBase *>(pk->phydro) : static_cast<FluidBase *>(pk->pmhd);
  std::string n(name);
  DvceArray<Real> *a = nullptr;
  if (n == "u0") a = &f->u0; else if (n == "w0") a = &f->w0; else if (n == "u1") a = &f->u1;
  else if (n == "dx") a = &pk->pmb->d_dx;
  else if (pk->pmhd) {
    auto *m = pk->pmhd;
    if (n == "bcc0") a = &m->bcc0;
    else if (n == "b0x1f") a = &m->b0.x1f; else if (n == "b0x2f") a = &m->b0.x2f; else if (n == "b0x3f") a = &m->b0.x3f;
    else if (n == "b1x1f") a = &m->b1.x1f; else if (n == "b1x2f") a = &m->b1.x2f; else if (n == "b1x3f") a = &m->b1.x3f;
  }
  if (!a) { if (count) *count = 0; return nullptr; }
  if (count) *count = static_cast<long long>(a->n);
  return a->p;
  )
}

const int *akmi_sim_lloc(void *h) { return static_cast<Sim *>(h)->pmesh->lloc_eachmb.data(); }
int akmi_sim_gids(void *h) { return static_cast<Sim *>(h)->pmesh->pmb_pack->gids; }
int akmi_sim_nmb_thisrank(void *h) { return static_cast<Sim *>(h)->pmesh->pmb_pack->nmb_thispack; }

}  // extern "C"
