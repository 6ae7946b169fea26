// akmi_host_comm.cpp -- ranks in the C++ host: the communicator (RCCL called directly, or
// caller-supplied callbacks), the same-level exchange plan of a rank, and MeshBoundaryValues for
// off-rank neighbours.  Reference: one MPI_Isend/Irecv per peer rank and variable class
// (src/bvals/bvals.cpp:134-310, bvals_cc.cpp:247-303, bvals_fc.cpp), MPI_Allreduce(MIN) of dt
// (src/mesh/mesh.cpp:634-637), block -> rank by Mesh::LoadBalance (src/mesh/load_balance.cpp:38-88).
//
// MI355X mapping.  One process per GPU.  A rank's messages of one variable class are ONE grouped
// RCCL call (ncclGroupStart .. ncclSend/ncclRecv per peer .. ncclGroupEnd) on the communicator's own
// HIP stream: xGMI is point to point, so a rank talks to its 1-7 peers over separate links at once,
// and the transfer runs underneath the kernels that do not need the halo (CornerE + CT + the c2p of
// the active cells, see the task bodies in akmi_host.cpp).  Ordering against the compute stream is by
// events only -- no host synchronisation on the data path.  The segment order inside a message is
// (receiver gid, receiver direction), which both sides derive from the block tables alone: no header
// exchange (the reference's one-shot handshake, bvals.cpp:248-270, has no equivalent here).
#include <arpa/inet.h>
#include <dlfcn.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>

#include <poll.h>
#include <rccl/rccl.h>

#include "akmi_host.hpp"

namespace akmi {
void set_error(const char *fmt, ...);
namespace host {

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  (void)hipGetLastError(); AKMI_THROW(std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)

// ---- RCCL entry points, resolved at run time ----------------------------------------------------
// libakmi.so has no link-time dependency on librccl: a Python process has torch's copy loaded
// already (the same library must serve both, two copies of RCCL in one process is asking for
// trouble), a plain C++ program gets /opt/rocm/lib/librccl.so.1.
namespace {
struct Rccl {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;      // optional (the exchange profile reports it)
  bool ok = false;
  std::string where;
};

Rccl &rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r;
  tried = true;
  void *h = nullptr;
  if (dlsym(RTLD_DEFAULT, "ncclCommInitRank")) { h = RTLD_DEFAULT; r.where = "process scope"; }
  const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (int q = 0; q < 4 && !h; ++q)
    if ((h = dlopen(names[q], RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) r.where = std::string(names[q]) + " (already loaded)";
  for (int q = 0; q < 4 && !h; ++q)
    if ((h = dlopen(names[q], RTLD_NOW | RTLD_GLOBAL))) r.where = names[q];
  if (!h) { r.where = "librccl not found"; return r; }
#define SYM(f) r.f = reinterpret_cast<decltype(r.f)>(dlsym(h, "nccl" #f))
  SYM(GetUniqueId); SYM(CommInitRank); SYM(CommDestroy); SYM(Send); SYM(Recv); SYM(GroupStart);
  SYM(GroupEnd); SYM(AllReduce); SYM(GetErrorString); SYM(CommCount);
#undef SYM
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Send && r.Recv && r.GroupStart &&
         r.GroupEnd && r.AllReduce && r.GetErrorString;
  return r;
}
#define NCCLCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) \
  AKMI_THROW(std::string(#x) + ": " + rccl().GetErrorString(r_)); } while (0)
}  // namespace

Comm &Comm::World() {
  static Comm c;
  return c;
}

void Comm::GetUniqueId(char id[128]) {
  if (!rccl().ok) AKMI_FATAL("RCCL is not available: " + rccl().where);
  ncclUniqueId u;
  NCCLCHK(rccl().GetUniqueId(&u));
  static_assert(sizeof(u) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(id, &u, 128);
}

void Comm::InitRCCL(int rank_, int nranks_, const char id[128]) {
  if (kind != Kind::none) Finalize();
  if (!rccl().ok) AKMI_FATAL("RCCL is not available: " + rccl().where);
  ncclUniqueId u;
  std::memcpy(&u, id, 128);
  ncclComm_t c;
  NCCLCHK(rccl().CommInitRank(&c, nranks_, u, rank_));
  nccl_ = c;
  rank = rank_; nranks = nranks_; kind = Kind::rccl;
  // highest priority: a transfer that queues behind nothing starts as soon as its data is ready
  int lo = 0, hi = 0;
  HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  HIPCHK(hipStreamCreateWithPriority(&comm_stream_, hipStreamNonBlocking, hi));
  for (int q = 0; q < 4; ++q) {
    HIPCHK(hipEventCreateWithFlags(&ready_[q], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&done_[q], hipEventDisableTiming));
  }
  HIPCHK(hipMalloc(&d_scratch_, 8*sizeof(Real)));
}

void Comm::InitCallbacks(int rank_, int nranks_, akmi_comm_exchange_fn ex, akmi_comm_allreduce_min_fn ar,
                         void *user) {
  if (kind != Kind::none) Finalize();
  rank = rank_; nranks = nranks_; kind = Kind::callback;
  ex_ = ex; ar_ = ar; user_ = user;
}

void Comm::Finalize() {
  if (kind == Kind::rccl) {
    HIPCHK(hipStreamSynchronize(comm_stream_));
    rccl().CommDestroy(static_cast<ncclComm_t>(nccl_));
    for (int q = 0; q < 4; ++q) { hipEventDestroy(ready_[q]); hipEventDestroy(done_[q]); }
    hipStreamDestroy(comm_stream_);
    hipFree(d_scratch_);
    nccl_ = nullptr; comm_stream_ = nullptr; d_scratch_ = nullptr;
  }
  for (auto &s : staged_) {
    for (Real *p : s.hs) hipHostFree(p);
    for (Real *p : s.hr) hipHostFree(p);
    s.hs.clear(); s.hr.clear(); s.cs.clear(); s.cr.clear(); s.m.clear();
  }
  for (int c = 0; c < kNCat; ++c) { for (hipEvent_t e : prof_ev_[c]) (void)hipEventDestroy(e); prof_ev_[c].clear(); }
  ProfReset(); prof_on = false;
  rank = 0; nranks = 1; kind = Kind::none;
  ex_ = nullptr; ar_ = nullptr; user_ = nullptr;
}

// ---- exchange profile ---------------------------------------------------------------------------
void Comm::ProfMark(int cat, hipStream_t st) {
  if (!prof_on) return;
  auto &ev = prof_ev_[cat];
  if (prof_used_[cat] == ev.size()) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    ev.push_back(e);
  }
  HIPCHK(hipEventRecord(ev[prof_used_[cat]++], st));
}
void Comm::ProfReset() {
  for (int c = 0; c < kNCat; ++c) prof_used_[c] = 0;
  prof_bytes_ = 0; prof_posts_ = 0; prof_peers_ = 0;
}
int Comm::ProfRead(double *out, int n) {
  if (!out || n < 12) { akmi::set_error("comm_profile_read: out needs room for 12 doubles"); return AKMI_FAIL; }
  for (int c = 0; c < kNCat; ++c) {
    double tot = 0.0;
    const size_t np = prof_used_[c]/2;
    for (size_t k = 0; k < np; ++k) {
      HIPCHK(hipEventSynchronize(prof_ev_[c][2*k + 1]));
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, prof_ev_[c][2*k], prof_ev_[c][2*k + 1]));
      tot += ms;
    }
    out[2*c] = tot; out[2*c + 1] = static_cast<double>(np);
  }
  out[8] = static_cast<double>(prof_bytes_); out[9] = static_cast<double>(prof_posts_); out[10] = prof_peers_;
  int cnt = 0;
  if (kind == Kind::rccl && rccl().CommCount) NCCLCHK(rccl().CommCount(static_cast<ncclComm_t>(nccl_), &cnt));
  out[11] = cnt;
  ProfReset();
  return AKMI_COMPLETE;
}

void Comm::Post(const std::vector<Msg> &m, hipStream_t compute, int c) {
  if (m.empty()) return;
  if (prof_on) {
    for (const Msg &x : m) prof_bytes_ += x.nsend*static_cast<long long>(sizeof(Real));
    ++prof_posts_;
    prof_peers_ = std::max(prof_peers_, static_cast<int>(m.size()));
  }
  if (kind == Kind::rccl) {
    ncclComm_t comm = static_cast<ncclComm_t>(nccl_);
    HIPCHK(hipEventRecord(ready_[c], compute));                 // the pack kernel has been enqueued
    HIPCHK(hipStreamWaitEvent(comm_stream_, ready_[c], 0));
    NCCLCHK(rccl().GroupStart());
    for (const Msg &x : m)
      if (x.nrecv > 0) NCCLCHK(rccl().Recv(x.recv, static_cast<size_t>(x.nrecv), ncclDouble, x.peer, comm, comm_stream_));
    for (const Msg &x : m)
      if (x.nsend > 0) NCCLCHK(rccl().Send(x.send, static_cast<size_t>(x.nsend), ncclDouble, x.peer, comm, comm_stream_));
    NCCLCHK(rccl().GroupEnd());
    HIPCHK(hipEventRecord(done_[c], comm_stream_));
    return;
  }
  if (kind != Kind::callback) AKMI_FATAL("off-rank neighbours without a communicator (akmi_comm_init_*)");
  // host-staged transport: device -> pinned host now, the blocking exchange in Wait (so that the
  // kernels enqueued in between still overlap with the copies)
  Staged &s = staged_[c];
  bool fits = s.hs.size() == m.size();
  for (size_t q = 0; fits && q < m.size(); ++q) fits = m[q].nsend <= s.cs[q] && m[q].nrecv <= s.cr[q];
  if (!fits) {
    for (Real *p : s.hs) hipHostFree(p);
    for (Real *p : s.hr) hipHostFree(p);
    s.hs.assign(m.size(), nullptr); s.hr.assign(m.size(), nullptr);
    s.cs.assign(m.size(), 0); s.cr.assign(m.size(), 0);
    for (size_t q = 0; q < m.size(); ++q) {
      s.cs[q] = std::max<long long>(m[q].nsend, 1); s.cr[q] = std::max<long long>(m[q].nrecv, 1);
      HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&s.hs[q]), sizeof(Real)*s.cs[q]));
      HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&s.hr[q]), sizeof(Real)*s.cr[q]));
    }
  }
  s.m = m;
  for (size_t q = 0; q < m.size(); ++q)
    if (m[q].nsend > 0)
      HIPCHK(hipMemcpyAsync(s.hs[q], m[q].send, sizeof(Real)*m[q].nsend, hipMemcpyDeviceToHost, compute));
  HIPCHK(hipStreamSynchronize(compute));
}

void Comm::Wait(hipStream_t compute, int c) {
  if (kind == Kind::rccl) {
    // the two marks straddle nothing but the wait: their distance is the time the compute stream stalls for the receives
    ProfMark(kWait, compute);
    HIPCHK(hipStreamWaitEvent(compute, done_[c], 0));
    ProfMark(kWait, compute);
    return;
  }
  Staged &s = staged_[c];
  if (s.m.empty()) return;
  const int n = static_cast<int>(s.m.size());
  std::vector<int> peers(n);
  std::vector<const double *> sp(n);
  std::vector<double *> rp(n);
  std::vector<long long> sc(n), rc(n);
  for (int q = 0; q < n; ++q) {
    peers[q] = s.m[q].peer; sp[q] = s.hs[q]; rp[q] = s.hr[q]; sc[q] = s.m[q].nsend; rc[q] = s.m[q].nrecv;
  }
  if (ex_(user_, n, peers.data(), sp.data(), sc.data(), rp.data(), rc.data()) != 0)
    AKMI_FATAL("the exchange callback reported a failure");
  for (int q = 0; q < n; ++q)
    if (s.m[q].nrecv > 0)
      HIPCHK(hipMemcpyAsync(s.m[q].recv, s.hr[q], sizeof(Real)*s.m[q].nrecv, hipMemcpyHostToDevice, compute));
  s.m.clear();
}

void Comm::AllReduceMin(Real *v, int n, hipStream_t compute) {
  if (kind == Kind::none) return;
  if (kind == Kind::callback) {
    if (ar_(user_, v, n) != 0) AKMI_FATAL("the allreduce callback reported a failure");
    return;
  }
  if (n > 8) AKMI_FATAL("Comm::AllReduceMin: at most 8 values");
  HIPCHK(hipMemcpyAsync(d_scratch_, v, sizeof(Real)*n, hipMemcpyHostToDevice, compute));
  NCCLCHK(rccl().AllReduce(d_scratch_, d_scratch_, static_cast<size_t>(n), ncclDouble, ncclMin,
                           static_cast<ncclComm_t>(nccl_), compute));
  HIPCHK(hipMemcpyAsync(v, d_scratch_, sizeof(Real)*n, hipMemcpyDeviceToHost, compute));
  HIPCHK(hipStreamSynchronize(compute));
}

bool Comm::AllReduceMinDevice(Real *d, int n, hipStream_t compute) {
  if (kind != Kind::rccl) return false;
  NCCLCHK(rccl().AllReduce(d, d, static_cast<size_t>(n), ncclDouble, ncclMin, static_cast<ncclComm_t>(nccl_), compute));
  return true;
}

// ---- block -> rank ------------------------------------------------------------------------------
void Mesh::LoadBalance(const std::vector<float> &clist) {       // load_balance.cpp:38-88
  const int nb = static_cast<int>(clist.size());
  double totalcost = 0.0;
  for (float c : clist) totalcost += c;
  int j = nranks - 1;
  double targetcost = totalcost/nranks, mycost = 0.0;
  rank_eachmb.assign(nb, 0);
  for (int i = nb - 1; i >= 0; --i) {
    if (targetcost == 0.0)
      AKMI_FATAL("There is at least one process which has no MeshBlock; decrease the number of processes "
                 "or use smaller MeshBlocks.");
    mycost += clist[i];
    rank_eachmb[i] = j;
    if (mycost >= targetcost && j > 0) {
      --j;
      totalcost -= mycost;
      mycost = 0.0;
      targetcost = totalcost/(j + 1);
    }
  }
  gids_eachrank.assign(nranks, 0); nmb_eachrank.assign(nranks, 0);
  j = 0;
  for (int i = 1; i < nb; ++i)
    if (rank_eachmb[i] != rank_eachmb[i - 1]) {
      nmb_eachrank[j] = i - gids_eachrank[j];
      gids_eachrank[++j] = i;
    }
  nmb_eachrank[j] = nb - gids_eachrank[j];
  if (j != nranks - 1)
    AKMI_FATAL("There is at least one process which has no MeshBlock; decrease the number of processes "
               "or use smaller MeshBlocks.");
}

// ---- the plan -----------------------------------------------------------------------------------
// remote ghost regions are numbered ("slots") in (peer, my gid, direction) order; a message to a peer
// carries the segments in (receiver gid, receiver direction) order -- the same order seen from the
// other side
// AKMI_SELF_EXCHANGE=1 with an RCCL communicator (functional check of the transport on however many GPUs there
// are, one included): neighbours on THIS rank are treated like neighbours on another rank, with the rank itself
// as the peer -- their ghost zones travel pack kernel -> send buffer -> ncclSend/ncclRecv to self inside the
// group call -> receive buffer -> unpack kernel instead of the same-rank gather.  Results must not change.
bool SelfExchange() {
  static const bool on = std::getenv("AKMI_SELF_EXCHANGE") && std::atoi(std::getenv("AKMI_SELF_EXCHANGE")) != 0;
  return on && Comm::World().kind == Comm::Kind::rccl;
}

static ExchangePlan BuildPlan(int my_rank, int gids, int nmb, const std::vector<int> &ngid,
                              const std::vector<int> &nrank) {
  ExchangePlan pl;
  pl.tab.assign(27*nmb, -1);
  const bool self = SelfExchange();
  for (int m = 0; m < nmb; ++m)
    for (int d = 0; d < 27; ++d) {
      const int g = ngid[27*m + d], r = nrank[27*m + d];
      if (g < 0) continue;
      if (r == my_rank && !self) { pl.tab[27*m + d] = g - gids; continue; }
      pl.recv_items[r].push_back({gids + m, d});
      pl.send_items[r].push_back({g, 26 - d, m, d});
    }
  for (auto &kv : pl.recv_items) { std::sort(kv.second.begin(), kv.second.end()); pl.peers.push_back(kv.first); }
  for (auto &kv : pl.send_items) {
    std::sort(kv.second.begin(), kv.second.end());
    if (!pl.recv_items.count(kv.first)) pl.peers.push_back(kv.first);
  }
  std::sort(pl.peers.begin(), pl.peers.end());
  int slot = 0;
  for (int r : pl.peers)
    if (pl.recv_items.count(r))
      for (const auto &it : pl.recv_items[r]) pl.tab[27*(it[0] - gids) + it[1]] = -(slot++ + 2);
  return pl;
}

ExchangeChannel PlanChannel(const ExchangePlan &pl, const std::function<long long(int)> &segsize) {
  ExchangeChannel ch;
  long long off = 0;
  for (int r : pl.peers) {
    const long long start = off;
    auto it = pl.recv_items.find(r);
    if (it != pl.recv_items.end())
      for (const auto &x : it->second) { ch.seg_off.push_back(off); off += segsize(x[1]); }
    ch.recv_slices[r] = {start, off};
  }
  ch.nrecvbuf = off;
  off = 0;
  for (int r : pl.peers) {
    const long long start = off;
    auto it = pl.send_items.find(r);
    if (it != pl.send_items.end())
      for (const auto &x : it->second) {
        ch.send_tab.push_back(x[2]); ch.send_tab.push_back(x[3]);
        ch.send_off.push_back(off);
        off += segsize(26 - x[3]);        // the receiver's region for its direction o = 26 - d
      }
    ch.send_slices[r] = {start, off};
  }
  ch.nsendbuf = off;
  ch.nsend = static_cast<int>(ch.send_off.size());
  return ch;
}

// called by the MeshBlock constructor once nghbr_gid / nghbr_rank are known
void BuildMeshBlockPlan(MeshBlock *pmb, int my_rank, int gids) {
  pmb->plan = BuildPlan(my_rank, gids, pmb->nmb, pmb->nghbr_gid, pmb->nghbr_rank);
}

// ---- MeshBoundaryValues (off-rank part) -------------------------------------------------------------
template <typename T>
static void Upload(DvceArray<T> &d, const std::vector<T> &h) {
  d.Realloc(std::max<size_t>(h.size(), 1));
  if (!h.empty()) HIPCHK(hipMemcpy(d.p, h.data(), sizeof(T)*h.size(), hipMemcpyHostToDevice));
}

MeshBoundaryValues::MeshBoundaryValues(MeshBlockPack *pp, const akmi_pack *pack, int nvar_, bool with_fc)
    : pmy_pack(pp), pack_c(pack), nvar(nvar_) {
  const ExchangePlan &pl = pp->pmb->plan;
  ch[0] = PlanChannel(pl, [&](int d) { return nvar*akmi_bvals_cc_segsize(pack, d); });
  if (with_fc) ch[1] = PlanChannel(pl, [&](int d) { return akmi_bvals_fc_segsize(pack, d); });
  for (int c = 0; c < (with_fc ? 2 : 1); ++c) {
    Upload(d_send_tab[c], ch[c].send_tab);
    Upload(d_send_off[c], ch[c].send_off);
    Upload(d_seg_off[c], ch[c].seg_off);
    sendbuf[c].Realloc(std::max<long long>(ch[c].nsendbuf, 1));
    recvbuf[c].Realloc(std::max<long long>(ch[c].nrecvbuf, 1));
  }
}
MeshBoundaryValues::~MeshBoundaryValues() {
  for (int c = 0; c < 2; ++c) {
    d_send_tab[c].Free(); d_send_off[c].Free(); d_seg_off[c].Free(); sendbuf[c].Free(); recvbuf[c].Free();
  }
}
bool MeshBoundaryValues::HasPeers() const { return !pmy_pack->pmb->plan.peers.empty(); }

void MeshBoundaryValues::Post(int c, hipStream_t st) {
  std::vector<Comm::Msg> m;
  for (int r : pmy_pack->pmb->plan.peers) {
    const auto s = ch[c].send_slices.at(r), v = ch[c].recv_slices.at(r);
    m.push_back({r, sendbuf[c].p + s.first, s.second - s.first, recvbuf[c].p + v.first, v.second - v.first});
  }
  Comm::World().Post(m, st, c);
}

#define AKCHK(x) do { if ((x) < 0) AKMI_THROW(std::string(#x) + ": " + akmi_last_error()); } while (0)

void MeshBoundaryValues::PackAndSendCC(Real *u, hipStream_t st) {
  AKCHK(akmi_bvals_cc_local(pack_c, nvar, pmy_pack->pmb->d_nghbr.p, u, st));
  if (!HasPeers()) return;
  Comm::World().ProfMark(Comm::kPack, st);
  AKCHK(akmi_bvals_cc_pack(pack_c, nvar, ch[0].nsend, d_send_tab[0].p, d_send_off[0].p, u, sendbuf[0].p, st));
  Comm::World().ProfMark(Comm::kPack, st);
  Post(0, st);
}
void MeshBoundaryValues::RecvAndUnpackCC(Real *u, hipStream_t st) {
  if (!HasPeers()) return;
  Comm::World().Wait(st, 0);
  Comm::World().ProfMark(Comm::kUnpack, st);
  AKCHK(akmi_bvals_cc_unpack(pack_c, nvar, pmy_pack->pmb->d_nghbr.p, d_seg_off[0].p, recvbuf[0].p, u, st));
  Comm::World().ProfMark(Comm::kUnpack, st);
}
void MeshBoundaryValues::PackAndSendFC(DvceFaceFld &b, hipStream_t st) {
  AKCHK(akmi_bvals_fc_local(pack_c, pmy_pack->pmb->d_nghbr.p, b.x1f.p, b.x2f.p, b.x3f.p, st));
  if (!HasPeers()) return;
  Comm::World().ProfMark(Comm::kPack, st);
  AKCHK(akmi_bvals_fc_pack(pack_c, ch[1].nsend, d_send_tab[1].p, d_send_off[1].p, b.x1f.p, b.x2f.p, b.x3f.p,
                           sendbuf[1].p, st));
  Comm::World().ProfMark(Comm::kPack, st);
  Post(1, st);
}
void MeshBoundaryValues::RecvAndUnpackFC(DvceFaceFld &b, hipStream_t st) {
  if (!HasPeers()) return;
  Comm::World().Wait(st, 1);
  Comm::World().ProfMark(Comm::kUnpack, st);
  AKCHK(akmi_bvals_fc_unpack(pack_c, pmy_pack->pmb->d_nghbr.p, d_seg_off[1].p, recvbuf[1].p, b.x1f.p, b.x2f.p,
                             b.x3f.p, st));
  Comm::World().ProfMark(Comm::kUnpack, st);
}

// ---- bootstrap over TCP for jobs started the torchrun way -------------------------------------------
static bool SendAll(int fd, const char *p, size_t n) {
  while (n) { ssize_t k = ::send(fd, p, n, 0); if (k <= 0) return false; p += k; n -= k; }
  return true;
}
static bool RecvAll(int fd, char *p, size_t n) {
  while (n) { ssize_t k = ::recv(fd, p, n, 0); if (k <= 0) return false; p += k; n -= k; }
  return true;
}

static bool BootstrapId(int rank, int nranks, char id[128], std::string &err) {
  const char *addr = std::getenv("MASTER_ADDR"), *port = std::getenv("MASTER_PORT");
  const int base = port ? std::atoi(port) : 29500;
  const char *ov = std::getenv("AKMI_BOOTSTRAP_PORT");
  const int p = ov ? std::atoi(ov) : base + 1;
  if (rank == 0) {
    Comm::GetUniqueId(id);
    int ls = ::socket(AF_INET, SOCK_STREAM, 0);
    int one = 1;
    setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in a{};
    a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_ANY); a.sin_port = htons(static_cast<uint16_t>(p));
    if (::bind(ls, reinterpret_cast<sockaddr *>(&a), sizeof(a)) != 0 || ::listen(ls, nranks) != 0) {
      err = "rank 0 cannot listen on port " + std::to_string(p); ::close(ls); return false;
    }
    // every rank announces itself with its number; only ranks 1..nranks-1, each once, get the id.  Anything
    // else that connects to the port is dropped, and a rank that never shows up costs AKMI_BOOTSTRAP_TIMEOUT
    // seconds (default 120), not a hang.
    const char *te = std::getenv("AKMI_BOOTSTRAP_TIMEOUT");
    const int timeout_ms = 1000*(te ? std::atoi(te) : 120);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<bool> seen(nranks, false);
    int served = 0;
    while (served < nranks - 1) {
      const int left = timeout_ms - static_cast<int>(std::chrono::duration_cast<std::chrono::milliseconds>(
                                         std::chrono::steady_clock::now() - t0).count());
      pollfd pf{ls, POLLIN, 0};
      if (left <= 0 || ::poll(&pf, 1, left) <= 0) {
        err = "rank 0: " + std::to_string(nranks - 1 - served) + " rank(s) did not ask for the id within " +
              std::to_string(timeout_ms/1000) + " s";
        ::close(ls); return false;
      }
      int fd = ::accept(ls, nullptr, nullptr);
      if (fd < 0) continue;
      timeval tv{5, 0};
      setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
      std::int32_t who = -1;
      if (RecvAll(fd, reinterpret_cast<char *>(&who), sizeof(who)) && who >= 1 && who < nranks && !seen[who] &&
          SendAll(fd, id, 128)) {
        seen[who] = true;
        ++served;
      }
      ::close(fd);
    }
    ::close(ls);
    return true;
  }
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_INET; hints.ai_socktype = SOCK_STREAM;
  if (getaddrinfo(addr ? addr : "127.0.0.1", std::to_string(p).c_str(), &hints, &res) != 0 || !res) {
    err = "cannot resolve MASTER_ADDR"; return false;
  }
  for (int attempt = 0; attempt < 600; ++attempt) {           // rank 0 may not be listening yet
    int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    if (::connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
      const std::int32_t who = rank;
      timeval tv{30, 0};
      setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
      const bool ok = SendAll(fd, reinterpret_cast<const char *>(&who), sizeof(who)) && RecvAll(fd, id, 128);
      ::close(fd); freeaddrinfo(res);
      if (!ok) err = "receiving the id from rank 0 failed";
      return ok;
    }
    ::close(fd);
    std::this_thread::sleep_for(std::chrono::milliseconds(100));
  }
  freeaddrinfo(res);
  err = "rank 0 did not answer on port " + std::to_string(p);
  return false;
}

}  // namespace host
}  // namespace akmi

using namespace akmi::host;

extern "C" {

// librccl is resolved at run time: a missing library is an error return of these entries, not an exit
static bool RcclAvailable(const char *who) {
  if (rccl().ok) return true;
  akmi::set_error("%s: RCCL is not available (%s)", who, rccl().where.c_str());
  return false;
}
int akmi_comm_unique_id(char id[128]) {
  if (!RcclAvailable("comm_unique_id")) return AKMI_FAIL;
  AKMI_C_ENTRY("akmi_comm_unique_id", AKMI_FAIL, Comm::GetUniqueId(id); return AKMI_COMPLETE;)
}
int akmi_comm_init_rccl(int rank, int nranks, const char id[128]) {
  if (rank < 0 || nranks < 1 || rank >= nranks) { akmi::set_error("comm_init_rccl: rank %d of %d", rank, nranks); return AKMI_FAIL; }
  if (!RcclAvailable("comm_init_rccl")) return AKMI_FAIL;
  AKMI_C_ENTRY("akmi_comm_init_rccl", AKMI_FAIL, Comm::World().InitRCCL(rank, nranks, id); return AKMI_COMPLETE;)
}
int akmi_comm_init_env(void) {
  const char *r = std::getenv("RANK"), *w = std::getenv("WORLD_SIZE"), *l = std::getenv("LOCAL_RANK");
  const int rank = r ? std::atoi(r) : 0, nranks = w ? std::atoi(w) : 1;
  if (l) { if (hipSetDevice(std::atoi(l)) != hipSuccess) { akmi::set_error("comm_init_env: hipSetDevice(LOCAL_RANK) failed"); return AKMI_FAIL; } }
  if (!RcclAvailable("comm_init_env")) return AKMI_FAIL;
  AKMI_C_ENTRY("akmi_comm_init_env", AKMI_FAIL,
    char id[128];
    std::string err;
    if (!BootstrapId(rank, nranks, id, err)) { akmi::set_error("comm_init_env: %s", err.c_str()); return AKMI_FAIL; }
    Comm::World().InitRCCL(rank, nranks, id);
    return AKMI_COMPLETE;
  )
}
int akmi_comm_init_callbacks(int rank, int nranks, akmi_comm_exchange_fn exchange,
                             akmi_comm_allreduce_min_fn allreduce_min, void *user) {
  if (rank < 0 || nranks < 1 || rank >= nranks || !exchange || !allreduce_min) {
    akmi::set_error("comm_init_callbacks: bad arguments"); return AKMI_FAIL;
  }
  AKMI_C_ENTRY("akmi_comm_init_callbacks", AKMI_FAIL,
    Comm::World().InitCallbacks(rank, nranks, exchange, allreduce_min, user);
    return AKMI_COMPLETE;
  )
}
int akmi_comm_finalize(void) { AKMI_C_ENTRY("akmi_comm_finalize", AKMI_FAIL, Comm::World().Finalize(); return AKMI_COMPLETE;) }
int akmi_comm_allreduce_min(double *vals, int n, void *stream) {
  AKMI_C_ENTRY("akmi_comm_allreduce_min", AKMI_FAIL,
    Comm::World().AllReduceMin(vals, n, static_cast<hipStream_t>(stream));
    return AKMI_COMPLETE;
  )
}
int akmi_comm_rank(void) { return Comm::World().rank; }
int akmi_comm_nranks(void) { return Comm::World().nranks; }
int akmi_comm_profile(int on) {
  Comm::World().prof_on = on != 0;
  if (on) Comm::World().ProfReset();
  return AKMI_COMPLETE;
}
int akmi_comm_profile_read(double *out, int n) {
  AKMI_C_ENTRY("akmi_comm_profile_read", AKMI_FAIL, return Comm::World().ProfRead(out, n);)
}

long long akmi_host_exchange_plan(const char *deck_text, int rank, int nranks, int nvar, int fc,
                                  long long *out, long long cap) {
  if (rank < 0 || nranks < 1 || rank >= nranks) { akmi::set_error("host_exchange_plan: rank %d of %d", rank, nranks); return -1; }
  AKMI_C_ENTRY("akmi_host_exchange_plan", -1,
  ParameterInput pin;
  pin.LoadFromString(deck_text);
  Mesh mesh(&pin, rank, nranks, true);
  const RegionIndcs &ind = mesh.mb_indcs;
  akmi_pack pk{};
  pk.nmb = mesh.pmb_pack->nmb_thispack; pk.nvar = nvar;
  pk.nx1 = ind.nx1; pk.nx2 = ind.nx2; pk.nx3 = ind.nx3; pk.ng = ind.ng;
  const ExchangePlan &pl = mesh.pmb_pack->pmb->plan;
  ExchangeChannel ch = fc ? PlanChannel(pl, [&](int d) { return akmi_bvals_fc_segsize(&pk, d); })
                          : PlanChannel(pl, [&](int d) { return nvar*akmi_bvals_cc_segsize(&pk, d); });
  std::vector<long long> v;
  v.push_back(static_cast<long long>(pl.peers.size()));
  for (int r : pl.peers) v.push_back(r);
  for (int r : pl.peers) {
    v.push_back(ch.send_slices[r].first); v.push_back(ch.send_slices[r].second);
    v.push_back(ch.recv_slices[r].first); v.push_back(ch.recv_slices[r].second);
  }
  v.push_back(pk.nmb);
  for (int t : pl.tab) v.push_back(t);
  v.push_back(ch.nsend);
  for (int t : ch.send_tab) v.push_back(t);
  for (long long t : ch.send_off) v.push_back(t);
  v.push_back(static_cast<long long>(ch.seg_off.size()));
  for (long long t : ch.seg_off) v.push_back(t);
  const long long n = static_cast<long long>(v.size());
  if (out && n <= cap) std::memcpy(out, v.data(), sizeof(long long)*n);
  return n;
  )
}

}  // extern "C"
