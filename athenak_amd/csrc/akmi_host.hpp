// akmi_host.hpp -- C++ host mirror of the reference's operator surface for the hot path.
//
// Same class and member names as the reference so that a task body reads like its original:
//   ParameterInput   src/parameter_input.hpp:67-127
//   TaskStatus/TaskID/Task/TaskList   src/tasklist/task_list.hpp:30-236
//   RegionSize/RegionIndcs/Mesh/MeshBlock/MeshBlockPack   src/mesh/mesh.hpp:25-185,
//                                                          src/mesh/meshblock_pack.hpp:44-97
//   hydro::Hydro / mhd::MHD (arrays + task member functions)   src/hydro/hydro.hpp:73-154,
//                                                              src/mhd/mhd.hpp:93-199
//   Driver   src/driver/driver.cpp:93-162,290-307,314-459
// Every task body is ONE call through the C ABI of include/akmi.h.  One process per GPU: the
// Z-ordered MeshBlock list is cut into one pack per rank (Mesh::LoadBalance), off-rank halos
// travel as one message per peer and variable class through Comm (RCCL ncclSend/ncclRecv on
// its own stream; akmi_host_comm.cpp), dt is reduced with ncclAllReduce(min).
#ifndef AKMI_HOST_HPP_
#define AKMI_HOST_HPP_
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>
#include <functional>
#include <array>
#include <list>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/akmi.h"

namespace akmi {
namespace host {

using Real = double;

// ---------------------------------------------------------------------------------------
class ParameterInput {
 public:
  ParameterInput() = default;
  explicit ParameterInput(const std::string &text) { LoadFromString(text); }
  void LoadFromString(const std::string &text);
  void ModifyFromCmdline(const std::vector<std::string> &args);
  bool DoesBlockExist(const std::string &b) const { return blocks_.count(b) != 0; }
  bool DoesParameterExist(const std::string &b, const std::string &n) const;
  std::string GetString(const std::string &b, const std::string &n) const;
  int GetInteger(const std::string &b, const std::string &n) const;
  Real GetReal(const std::string &b, const std::string &n) const;
  bool GetBoolean(const std::string &b, const std::string &n) const;
  std::string GetOrAddString(const std::string &b, const std::string &n, const std::string &d);
  int GetOrAddInteger(const std::string &b, const std::string &n, int d);
  Real GetOrAddReal(const std::string &b, const std::string &n, Real d);
  bool GetOrAddBoolean(const std::string &b, const std::string &n, bool d);
  void SetReal(const std::string &b, const std::string &n, Real v);
  std::vector<std::string> BlockNames() const {
    std::vector<std::string> v;
    for (const auto &b : blocks_) v.push_back(b.first);
    return v;
  }
 private:
  std::map<std::string, std::map<std::string, std::string>> blocks_;
};

[[noreturn]] void Fatal(const char *file, int line, const std::string &msg);
#define AKMI_FATAL(msg) ::akmi::host::Fatal(__FILE__, __LINE__, (msg))

// Errors of the INPUT DECK keep the reference's convention: "### FATAL ERROR in <file> at line <n>" + exit (AKMI_FATAL; e.g.
// src/eos/eos.cpp:37-39, src/mesh/mesh.cpp:234).  Run-time failures -- a device allocation, a HIP / RCCL call, a library entry
// that returned AKMI_FAIL -- are thrown as HostError; together with whatever else the C++ runtime throws (std::bad_alloc,
// std::invalid_argument out of std::stoi on a malformed deck value, ...) they stop at the C entry points, which return
// AKMI_FAIL / NULL with the message in akmi_last_error() (include/akmi.h:24-27; the TaskStatus::fail of
// src/tasklist/task_list.hpp:30).  No C++ exception crosses `extern "C"`.
struct HostError : std::runtime_error { using std::runtime_error::runtime_error; };
[[noreturn]] void Throw(const char *file, int line, const std::string &msg);
#define AKMI_THROW(msg) ::akmi::host::Throw(__FILE__, __LINE__, (msg))
// the body of a C entry point: what it returns on an exception, then the statements
void NoteException(const char *entry) noexcept;      // message of the exception in flight -> akmi_last_error()
#define AKMI_C_ENTRY(entry, on_error, ...)                               \
  try { __VA_ARGS__ }                                                    \
  catch (...) { ::akmi::host::NoteException(entry); return on_error; }

// ---------------------------------------------------------------------------------------
class Driver;
enum class TaskStatus {fail, complete, incomplete};
enum class TaskListStatus {running, stuck, complete, nothing_to_do};

class TaskID {   // task_list.hpp:37-81
 public:
  TaskID() = default;
  explicit TaskID(unsigned int id) { bits_ = (id == 0) ? 0ull : (1ull << (id - 1)); }
  void Clear() { bits_ = 0; }
  bool CheckDependencies(const TaskID &dep) const { return (bits_ & dep.bits_) == dep.bits_; }
  void SetComplete(const TaskID &rhs) { bits_ |= rhs.bits_; }
  bool operator==(const TaskID &rhs) const { return bits_ == rhs.bits_; }
  bool operator!=(const TaskID &rhs) const { return bits_ != rhs.bits_; }
  TaskID operator|(const TaskID &rhs) const { TaskID r; r.bits_ = bits_ | rhs.bits_; return r; }
 private:
  std::uint64_t bits_ = 0;
};

class Task {     // task_list.hpp:88-110
 public:
  Task(TaskID id, TaskID dep, std::function<TaskStatus(Driver *, int)> func)
      : myid_(id), dep_(dep), func_(func) {}
  TaskStatus operator()(Driver *d, int s) { return func_(d, s); }
  TaskID GetID() { return myid_; }
  TaskID GetDependency() { return dep_; }
  void SetComplete() { complete_ = true; }
  void SetIncomplete() { complete_ = false; }
  bool IsComplete() { return complete_; }
 private:
  TaskID myid_, dep_;
  bool complete_ = false;
  std::function<TaskStatus(Driver *, int)> func_;
};

class TaskList {  // task_list.hpp:117-236
 public:
  bool IsComplete();
  bool Empty() { return task_list_.empty(); }
  int Size() { return static_cast<int>(task_list_.size()); }
  void Reset();
  TaskListStatus DoAvailable(Driver *d, int s);
  template <class F, class T>
  TaskID AddTask(F func, T *obj, TaskID &dep) {
    TaskID id(static_cast<unsigned int>(task_list_.size()) + 1);
    task_list_.push_back(Task(id, dep, [=](Driver *d, int s) mutable -> TaskStatus {
      return (obj->*func)(d, s);
    }));
    return id;
  }
 private:
  std::list<Task> task_list_;
  TaskID tasks_completed_;
};

// ---------------------------------------------------------------------------------------
struct RegionSize {     // mesh.hpp:25-29
  Real x1min, x2min, x3min, x1max, x2max, x3max, dx1, dx2, dx3;
};
struct RegionIndcs {    // mesh.hpp:35-41
  int ng, nx1, nx2, nx3, is, ie, js, je, ks, ke;
};

template <typename T>
struct DvceArray {      // flat device array (layout contract: include/akmi.h)
  T *p = nullptr;
  size_t n = 0;
  void Realloc(size_t count);
  void Free();
  T *data() const { return p; }
};
struct DvceFaceFld { DvceArray<Real> x1f, x2f, x3f; };
struct DvceEdgeFld { DvceArray<Real> x1e, x2e, x3e; };

class Mesh;
class MeshBlockPack;

// ---- ranks: global_variable::my_rank / nranks and the MPI calls of the reference --------------
// (src/globals.hpp, bvals.cpp:134-310 one message per peer rank, mesh.cpp:634-637 dt reduction).
// Transports: RCCL (one GPU per rank; librccl is resolved at run time so that the library loads
// on machines without it), or caller-supplied callbacks that move HOST buffers (another MPI, or
// gloo in the tests where two ranks have to share the only GPU, which RCCL refuses).
class Comm {
 public:
  static Comm &World();
  enum class Kind { none, rccl, callback };
  struct Msg { int peer; Real *send; long long nsend; Real *recv; long long nrecv; };
  int rank = 0, nranks = 1;
  Kind kind = Kind::none;
  void InitRCCL(int rank, int nranks, const char id[128]);
  void InitCallbacks(int rank, int nranks, akmi_comm_exchange_fn ex, akmi_comm_allreduce_min_fn ar,
                     void *user);
  void Finalize();
  // all messages of one variable class (uniform meshes: channel 0 = cell-centred, 1 = face-centred;
  // refined meshes: 0 cc variables, 1 cc fluxes, 2 fc variables, 3 edge EMFs).  The data is
  // ready on `compute` when Post is called; Wait makes `compute` wait for the receives.
  void Post(const std::vector<Msg> &m, hipStream_t compute, int channel);
  void Wait(hipStream_t compute, int channel);
  void AllReduceMin(Real *host_vals, int n, hipStream_t compute);
  // the same on values that are already in device memory, in place, asynchronous on `compute` (RCCL transport only;
  // returns false -- nothing done -- for the callback transport or without peers)
  bool AllReduceMinDevice(Real *dev_vals, int n, hipStream_t compute);
  static void GetUniqueId(char id[128]);
  // exchange profile (akmi_comm_profile): event pairs on the compute stream per category
  enum ProfCat { kPack = 0, kWait = 1, kUnpack = 2, kDtReduce = 3, kNCat = 4 };
  bool prof_on = false;
  void ProfMark(int cat, hipStream_t st);          // two calls = one pair
  void ProfReset();
  int ProfRead(double *out, int n);
 private:
  std::vector<hipEvent_t> prof_ev_[kNCat];
  size_t prof_used_[kNCat] = {0, 0, 0, 0};
  long long prof_bytes_ = 0, prof_posts_ = 0;
  int prof_peers_ = 0;
  void *nccl_ = nullptr;                 // ncclComm_t
  hipStream_t comm_stream_ = nullptr;
  hipEvent_t ready_[4] = {nullptr, nullptr, nullptr, nullptr}, done_[4] = {nullptr, nullptr, nullptr, nullptr};
  Real *d_scratch_ = nullptr;            // device doubles for the dt reduction
  akmi_comm_exchange_fn ex_ = nullptr;
  akmi_comm_allreduce_min_fn ar_ = nullptr;
  void *user_ = nullptr;
  struct Staged { std::vector<Msg> m; std::vector<Real *> hs, hr; std::vector<long long> cs, cr; } staged_[4];
};

// who sends what to whom on a uniform mesh (bvals.py MeshBoundaryValues.__init__): pure host data
struct ExchangePlan {
  std::vector<int> peers;                                     // sorted ranks this rank talks to
  std::vector<int> tab;                                       // [nmb][27] as include/akmi.h describes
  std::map<int, std::vector<std::array<int, 2>>> recv_items;  // peer -> (my gid, direction o), sorted
  std::map<int, std::vector<std::array<int, 4>>> send_items;  // peer -> (recv gid, recv o, my m, d), sorted
};
struct ExchangeChannel {                                      // one variable class (bvals.py _plan)
  int nsend = 0;
  std::vector<int> send_tab;                                  // [nsend][2] {m, d}
  std::vector<long long> send_off, seg_off;
  std::map<int, std::pair<long long, long long>> send_slices, recv_slices;   // peer -> [a, b) doubles
  long long nsendbuf = 0, nrecvbuf = 0;
};
ExchangeChannel PlanChannel(const ExchangePlan &pl, const std::function<long long(int)> &segsize);
class MeshBlock;
void BuildMeshBlockPlan(MeshBlock *pmb, int my_rank, int gids);
bool SelfExchange();   // AKMI_SELF_EXCHANGE=1 + RCCL communicator: same-rank neighbours through ncclSend/ncclRecv to self

// ---- static mesh refinement (akmi_host_smr.cpp) ----------------------------------------------
struct LogicalLocation { int lx1, lx2, lx3, level; };          // mesh.hpp:54-60
int NeighborIndex(int ix, int iy, int iz, int n1, int n2);      // nghbr_index.hpp:28-54

class MeshBlockTree {   // meshblock_tree.hpp:27-62, meshblock_tree.cpp
 public:
  struct Node {
    LogicalLocation lloc;
    std::vector<std::unique_ptr<Node>> leaf;   // empty: a leaf; entries may be null in the root grid
    int gid;
  };
  MeshBlockTree(const int nmb_root[3], const bool periodic[6], int ndim);
  void AddNode(const LogicalLocation &rloc);
  void Refine(Node *node);
  std::vector<LogicalLocation> CreateZOrderedLLList();
  Node *FindNeighbor(const LogicalLocation &myloc, int ox1, int ox2, int ox3);
  int root_level = 0;
 private:
  Node *MakeChild(Node *node, int n);
  void CreateRoot(Node *node);
  void Walk(Node *node, std::vector<LogicalLocation> &out);
  bool Wrap(int &l, int d, int level) const;
  std::unique_ptr<Node> root_;
  int nmb_root_[3], ndim_, nleaf_;
  bool periodic_[6];
};

class MeshBlock {       // meshblock.cpp:25-131
 public:
  MeshBlock(MeshBlockPack *ppack, int igids, int nmb);
  ~MeshBlock();
  void SetNeighborsSMR(Mesh *pm);   // meshblock.cpp:142-425
  int nmb;
  std::vector<int> mb_lev;        // logical level of each block
  std::vector<int> nghbr_smr;     // multilevel: [nmb][56][3] {index in the pack | nmb: another rank | -1, level, dest slot}
  std::vector<int> nghbr_smr_gid; // [nmb][56] global id of that neighbour
  std::vector<int> mb_gid;
  std::vector<RegionSize> mb_size;
  std::vector<int> mb_bcs;        // [nmb][6]
  int bc_dirs = 7;                // bit d: some block has a physical boundary across direction d (akmi_*_bcs_dirs)
  std::vector<int> nghbr_gid, nghbr_rank;   // [nmb][27] same-level neighbour: gid (-1 none) and its rank
  ExchangePlan plan;              // plan.tab = the device neighbour table (local index | -1 | remote slot)
  DvceArray<Real> d_dx;           // [nmb][3]
  DvceArray<int> d_bcs, d_nghbr;
};

namespace hydro { class Hydro; }
namespace mhd { class MHD; }

class MeshBlockPack {   // meshblock_pack.hpp:44-97
 public:
  MeshBlockPack(Mesh *pm, int igids, int igide);
  ~MeshBlockPack();
  void AddPhysics(ParameterInput *pin);
  Mesh *pmesh;
  int gids, gide, nmb_thispack;
  MeshBlock *pmb = nullptr;
  hydro::Hydro *phydro = nullptr;
  mhd::MHD *pmhd = nullptr;
  std::map<std::string, std::shared_ptr<TaskList>> tl_map;
};

class Mesh {            // mesh.hpp:92-185
 public:
  // host_only: tables only, no device memory (the exchange-plan unit test runs without a GPU)
  explicit Mesh(ParameterInput *pin, int my_rank = 0, int nranks = 1, bool host_only = false);
  ~Mesh();
  void LoadBalance(const std::vector<float> &clist);   // load_balance.cpp:38-88
  int my_rank, nranks;
  bool host_only;
  std::vector<int> rank_eachmb, gids_eachrank, nmb_eachrank;
  void NewTimeStep(const Real tlim);   // mesh.cpp:573-643
  int NumberOfMeshBlockCells() const { return mb_indcs.nx1*mb_indcs.nx2*mb_indcs.nx3; }
  RegionSize mesh_size;
  RegionIndcs mesh_indcs, mb_indcs;
  int mesh_bcs[6];
  bool one_d, two_d, three_d, multi_d, strictly_periodic;
  int nmb_rootx1, nmb_rootx2, nmb_rootx3, nmb_total;
  bool multilevel = false;        // <mesh_refinement>/refinement = static
  bool prolong_prims = false;     // <mesh_refinement>/prolong_primitives (mesh_refinement.cpp:52)
  int root_level = 0, max_level = 0;
  std::unique_ptr<MeshBlockTree> ptree;
  std::vector<LogicalLocation> lloc_tree;   // multilevel: Z-ordered leaves with their levels
  void BuildTreeFromScratch(ParameterInput *pin);   // build_tree.cpp:32-258 (static refinement)
  std::vector<int> lloc_eachmb;   // [nmb_total][3], Z-ordered
  Real time, dt, dtold, cfl_no;
  int ncycle;
  MeshBlockPack *pmb_pack = nullptr;
};

// level-aware boundary values of one pack (src/bvals/bvals.hpp:134-267 on a multilevel mesh): the
// buffer index tables on the device + the receive buffers; task bodies = akmi_smr_* calls
// same-level exchange with off-rank neighbours (uniform meshes): buffers + device tables of one pack
class MeshBoundaryValues {     // bvals.hpp:134-267, the off-rank part of bvals_cc.cpp / bvals_fc.cpp
 public:
  MeshBoundaryValues(MeshBlockPack *pp, const akmi_pack *pack, int nvar, bool with_fc);
  ~MeshBoundaryValues();
  MeshBlockPack *pmy_pack;
  const akmi_pack *pack_c;
  int nvar;
  bool HasPeers() const;
  void PackAndSendCC(Real *u, hipStream_t st);
  void RecvAndUnpackCC(Real *u, hipStream_t st);
  void PackAndSendFC(DvceFaceFld &b, hipStream_t st);
  void RecvAndUnpackFC(DvceFaceFld &b, hipStream_t st);
  ExchangeChannel ch[2];
 private:
  void Post(int c, hipStream_t st);
  DvceArray<int> d_send_tab[2];
  DvceArray<long long> d_send_off[2], d_seg_off[2];
  DvceArray<Real> sendbuf[2], recvbuf[2];
};

class MeshBoundaryValuesSMR {
 public:
  MeshBoundaryValuesSMR(MeshBlockPack *pp, int nvar);
  ~MeshBoundaryValuesSMR();
  MeshBlockPack *pmy_pack;
  int nvar, nnghbr;
  akmi_smr smr_c{};
  DvceArray<int> d_nghbr, d_lev, d_cc, d_fc, d_ndat, d_ox, d_nflx;
  DvceArray<int> d_same;          // [nmb][27] same-level neighbours in this pack: filled by akmi_bvals_cc_local
  DvceArray<unsigned char> d_needs;   // [nmb] block has a coarser neighbour (akmi_smr::needs_coarse)
  DvceArray<int> d_lists;         // akmi_smr::lists (work lists of (block, slot) pairs)
  void BuildLists(const akmi_pack *pk, hipStream_t st);
  // the face-field exchange as lists of element copies (akmi_smr_fc_map): [0] what the unpack does, [1] the outgoing
  // messages; fc_map_on == false: akmi_smr_pack_fc + akmi_smr_unpack_fc (AKMI_SMR_FC_MAP=0)
  DvceArray<int> d_fc_map[2];
  long long fc_np[2] = {0, 0}, fc_tail[2] = {0, 0};
  bool fc_map_on = false;
  void BuildFcMaps(const akmi_pack *pk, hipStream_t st);
  // one rank: the cell-centred exchange as a copy list of variable 0 (akmi_smr_cc_map); AKMI_SMR_CC_MAP=0: off
  DvceArray<int> d_cc_map;
  long long cc_np = 0, cc_tail = 0;
  bool cc_map_on = false;
  void BuildCcMap(const akmi_pack *pk, hipStream_t st);
  DvceArray<long long> d_layout, d_soff, d_roff;
  DvceArray<Real> buf[4];         // cc vars, cc flux, fc vars, fc flux
  // ranks: peers and the slices of buf[cls] that travel (akmi_smr::soff/roff address the segments)
  std::vector<int> peers;
  std::map<int, std::pair<long long, long long>> send_slices[4], recv_slices[4];
  void Post(int cls, hipStream_t st);
  void Wait(int cls, hipStream_t st);
};

// physics base: what Hydro and MHD share ------------------------------------------------
struct EOS_Data { Real gamma, iso_cs, dfloor, pfloor, tfloor, sfloor, sigma_max; bool is_ideal; };
struct EquationOfState { EOS_Data eos_data; };

class FluidBase {
 public:
  FluidBase(MeshBlockPack *pp, ParameterInput *pin, const std::string &blk);
  virtual ~FluidBase();
  MeshBlockPack *pmy_pack;
  EquationOfState *peos;
  int recon_method, rsolver_method;
  bool fused;
  akmi_pack pack_c;
  DvceArray<Real> u0, w0, u1;
  DvceArray<Real> w1;                   // second primitive array (allocated on first use): akmi_hydro_stage_w writes the new
                                        // primitives of the active cells there while neighbours still read w0; then the two trade places
  bool w_swapped = false;               // w0 lives in the buffer that was w1 when the arrays were created
  DvceArray<int> counters;
  DvceArray<Real> dt3;
  DvceArray<char> ws;
  int nfluid = 5, nscalars = 0, nvars = 5;   // nhydro|nmhd (4 isothermal), passive scalars
  // constant-coefficient diffusion (src/diffusion): objects exist when the parameter is in the deck
  bool has_visc = false, has_cond = false, has_resist = false;
  Real nu_iso = 0.0, alpha_iso = 0.0, eta_ohm = 0.0, eta_ad = 0.0;
  const Real *bcc_cells = nullptr;     // MHD: cell-centred field, for the ambipolar time step
  bool kinematic = false;              // <time>/evolution = kinematic: advect solvers, velocity-only dt
  Real dt_visc = static_cast<Real>(FLT_MAX), dt_cond = static_cast<Real>(FLT_MAX),
       dt_resist = static_cast<Real>(FLT_MAX);
  DvceArray<Real> dtmin_cond;
  bool use_fofc = false;                // hydro.hpp:116-117, mhd.hpp
  bool OopFirst(const Driver *d, int stage) const;     // first stage of the task path out of place (akmi_host.cpp)
  bool dt_reduced = false;              // dtnew is already the minimum over all ranks (FinishNewDt)
  DvceArray<unsigned char> fofc;
  DvceArray<int> nfofc;                 // EventCounters::nfofc (mesh.hpp:71), kept on the device
  Real dtnew = static_cast<Real>(FLT_MAX);
  hipStream_t stream = nullptr;
  // static mesh refinement: coarse buffers (hydro.cpp:300-310, mhd.cpp:368-380) + boundary values
  bool multilevel = false;
  akmi_pack cpack_c;                    // the coarse buffers as a pack of nx/2 cells (coarse BCs)
  DvceArray<Real> coarse_u0, coarse_w0;  // coarse_w0: <mesh_refinement>/prolong_primitives = true only
  MeshBoundaryValuesSMR *psmr = nullptr;
  MeshBoundaryValues *pbval = nullptr;  // off-rank neighbours (uniform meshes, nranks > 1)
  bool peers() const { return pbval && pbval->HasPeers(); }
  const Real *dt_dev = nullptr;         // set by the Driver when cycles are replayed from a hipGraph
  // the out-of-place first stage trades the two registers; true while u0 / b0 live in the buffers that were
  // u1 / b1 when the arrays were created (akmi_sim_execute copies back so that akmi_sim_array pointers stay valid)
  bool u_swapped = false, b_swapped = false;
  virtual void RestoreRegisters();
 public:
  void FinishNewDtPublic() { FinishNewDt(); }
 protected:
  void FinishNewDt();
  void AddDiffusionFluxes(DvceFaceFld &flx, int face_shaped);   // hydro_tasks.cpp:183-189
  void DiffusionNewDt();                                        // hydro_newdt.cpp:128-133
  bool interior_done_ = false, dt_ready_ = false;
  // single-rank uniform meshes: the gather of SendU / SendB applies the physical boundary functions as well
  // (akmi_bvals_*_local_bcs); ApplyPhysicalBCs then has nothing left to do for that array.  AKMI_FOLD_BCS=0: the separate
  // kernels (A/B runs).  dt3_reset_: that launch also reset the CFL minima of the last stage.
  // They are hand-shakes inside ONE stage (set by SendU / SendB, consumed by ApplyPhysicalBCs / ConToPrim of the same
  // stage): the Driver clears them when a stage begins (BeginStage), so a task list that drops the consumer cannot leave one
  // standing for a later stage.
  bool u_bcs_done_ = false, b_bcs_done_ = false, dt3_reset_ = false;
  // hydro, after akmi_hydro_stage_w: the gather of SendU also converts the ghost shell (akmi_hydro_ghost_c2p) -- want_:
  // RKUpdate asks for it, shell_done_: ConToPrim has nothing left to do
  bool want_ghost_c2p_ = false, shell_done_ = false;
 public:
  void BeginStage() { u_bcs_done_ = b_bcs_done_ = dt3_reset_ = want_ghost_c2p_ = shell_done_ = false; }
 protected:
  static bool FoldBCs();
  bool BcsCommuteWithC2P() const;
  void GatherU(Driver *d, int stage);
};

namespace hydro {
class Hydro : public FluidBase {    // hydro.hpp:73-154
 public:
  Hydro(MeshBlockPack *pp, ParameterInput *pin);
  ~Hydro() override;
  DvceFaceFld uflx;
  void AssembleHydroTasks(std::map<std::string, std::shared_ptr<TaskList>> tl);
  void StagePhase(Driver *d, int stage, int phases);   // akmi_hydro_stage_phase
  TaskStatus InitRecv(Driver *d, int stage) { return TaskStatus::complete; }
  TaskStatus CopyCons(Driver *d, int stage);
  TaskStatus Fluxes(Driver *d, int stage);
  TaskStatus SendFlux(Driver *d, int stage);
  TaskStatus RecvFlux(Driver *d, int stage);
  TaskStatus RKUpdate(Driver *d, int stage);
  TaskStatus HydroSrcTerms(Driver *d, int stage) { return TaskStatus::complete; }
  TaskStatus RestrictU(Driver *d, int stage);
  TaskStatus SendU(Driver *d, int stage);
  TaskStatus RecvU(Driver *d, int stage);
  TaskStatus Prolongate(Driver *d, int stage);
  TaskStatus ApplyPhysicalBCs(Driver *d, int stage);
  TaskStatus ConToPrim(Driver *d, int stage);
  TaskStatus NewTimeStep(Driver *d, int stage);
  TaskStatus ClearSend(Driver *d, int stage) { return TaskStatus::complete; }
  TaskStatus ClearRecv(Driver *d, int stage) { return TaskStatus::complete; }
};
}  // namespace hydro

namespace mhd {
class MHD : public FluidBase {      // mhd.hpp:93-199
 public:
  MHD(MeshBlockPack *pp, ParameterInput *pin);
  ~MHD() override;
  DvceArray<Real> bcc0;
  DvceFaceFld b0, b1, uflx, coarse_b0;
  DvceEdgeFld efld;
  DvceArray<Real> e3x1, e2x1, e1x2, e3x2, e2x3, e1x3;
  void AssembleMHDTasks(std::map<std::string, std::shared_ptr<TaskList>> tl);
  void StagePhase(Driver *d, int stage, int phases);   // akmi_mhd_stage_phase
  void RestoreRegisters() override;
  TaskStatus SaveMHDState(Driver *d, int stage) { return TaskStatus::complete; }
  TaskStatus InitRecv(Driver *d, int stage) { return TaskStatus::complete; }
  TaskStatus CopyCons(Driver *d, int stage);
  TaskStatus Fluxes(Driver *d, int stage);
  TaskStatus SendFlux(Driver *d, int stage);
  TaskStatus RecvFlux(Driver *d, int stage);
  TaskStatus RKUpdate(Driver *d, int stage);
  TaskStatus MHDSrcTerms(Driver *d, int stage) { return TaskStatus::complete; }
  TaskStatus RestrictU(Driver *d, int stage);
  TaskStatus SendU(Driver *d, int stage);
  TaskStatus RecvU(Driver *d, int stage);
  TaskStatus EField(Driver *d, int stage);
  TaskStatus SendE(Driver *d, int stage);      // identity on uniform meshes, EMF correction with levels
  TaskStatus RecvE(Driver *d, int stage);
  TaskStatus CT(Driver *d, int stage);
  TaskStatus RestrictB(Driver *d, int stage);
  TaskStatus SendB(Driver *d, int stage);
  TaskStatus RecvB(Driver *d, int stage);
  TaskStatus Prolongate(Driver *d, int stage);
  TaskStatus ApplyPhysicalBCs(Driver *d, int stage);
  TaskStatus ConToPrim(Driver *d, int stage);
  TaskStatus NewTimeStep(Driver *d, int stage);
  TaskStatus ClearSend(Driver *d, int stage) { return TaskStatus::complete; }
  TaskStatus ClearRecv(Driver *d, int stage) { return TaskStatus::complete; }
};
}  // namespace mhd

class Driver {          // driver.cpp
 public:
  Driver(ParameterInput *pin, Mesh *pmesh);
  void ExecuteTaskList(Mesh *pm, const std::string &tl, int stage);
  void InitBoundaryValuesAndPrimitives(Mesh *pm);
  void Initialize(Mesh *pm);
  int Execute(Mesh *pm, int max_cycles);
  std::string integrator;
  Real tlim;
  int nlim, nexp_stages;
  Real gam0[4], gam1[4], beta[4], delta[4];
  std::int64_t nmb_updated_ = 0;
  ~Driver();
  // One cycle (all stages) captured into a hipGraph and replayed: on small packs a cycle is a chain of
  // dependent launches of a few microseconds each and the host cannot issue them fast enough.  Nothing
  // in the captured calls changes from cycle to cycle except dt, which the kernels read from d_dt
  // (akmi_*_stage_fused_dt).  Eligible: fused stage, one rank, uniform mesh.  <time>/cycle_graph = auto
  // (1-D packs) | true | false; AKMI_CYCLE_GRAPH=0/1 overrides.
  bool use_graph = false, capturing = false;
  // akmi_sim_profile: live timing of the fused-stage launch group -- a HIP event pair on the launch stream
  // around every akmi_*_stage_fused / akmi_*_stage_phase call of the cycles that follow (bench.py's roofline entry)
  bool prof_on = false;
  std::vector<hipEvent_t> prof_ev;
  size_t prof_used = 0;
  void ProfMark(hipStream_t st);
  int ProfRead(double *ms_total, long long *calls);
  hipGraphExec_t cycle_exec = nullptr;
  DvceArray<Real> d_dt;              // [0] dt of the cycle being enqueued, [1] its start time (run-ahead mode)
  Real *h_dt = nullptr;              // pinned
  // Run-ahead cycles.  The reference reads the new time step back at the end of every cycle (hydro_newdt.cpp:121-124,
  // mesh.cpp:573-643) and the device idles while the host wakes up, computes dt and issues the next cycle's first
  // launches -- 30-40 us, 10 % of a cycle of the 128^3 hydro deck.  Here Mesh::NewTimeStep runs ON the device at the end
  // of the cycle (k_mesh_newdt: the same operations in the same order on (dt3, dt, time, tlim, cfl_no)), every kernel takes
  // dt from device memory, and the host enqueues cycle n+1 while cycle n runs: to do that it needs time_(n+1) =
  // time_n + dt_n only, and dt_n is the result of cycle n-1.  The results of a cycle reach the host through a pinned
  // slot + event, one cycle late; Execute drains them before it returns, so (time, dt, ncycle) are exact at every
  // akmi_sim_* call.  Eligible like the cycle graph: fused stage, one rank, uniform mesh, no diffusion time steps.
  // <time>/run_ahead = auto (on when eligible) | true | false; AKMI_RUN_AHEAD=0/1 overrides.
  bool run_ahead = false, ra_active = false;     // ra_active: inside the run-ahead loop of Execute
  Real *ra_slot = nullptr;           // pinned, device-visible: 2 slots x {dt, time, dtnew}
  hipEvent_t ra_ev[2] = {nullptr, nullptr};
  long long ra_cycle = 0;            // cycles enqueued in run-ahead mode (slot = ra_cycle & 1)
  void EnqueueMeshNewDt(FluidBase *f);
 private:
  void RunStages(Mesh *pm);
};

}  // namespace host
}  // namespace akmi
#endif  // AKMI_HOST_HPP_
