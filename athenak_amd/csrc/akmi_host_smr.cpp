// akmi_host_smr.cpp -- static mesh refinement in the C++ host: the MeshBlockTree with its 2:1 rule
// (src/mesh/meshblock_tree.cpp:64-465), Mesh::BuildTreeFromScratch with <refined_region*> blocks
// (src/mesh/build_tree.cpp:32-258), the 56-slot neighbour table of MeshBlock::SetNeighbors
// (src/mesh/meshblock.cpp:142-425), the index ranges of the boundary buffers
// (src/bvals/buffs_cc.cpp, buffs_fc.cpp) as flat device tables, and MeshBoundaryValuesSMR, whose task
// bodies are the akmi_smr_* entry points of include/akmi.h.
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "akmi_host.hpp"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  (void)hipGetLastError(); AKMI_THROW(std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)

namespace akmi {
namespace host {

static Real LeftEdgeXs(int ith, int n, Real xmin, Real xmax) {   // cell_locations.hpp:23-28
  Real x = static_cast<Real>(ith)/static_cast<Real>(n);
  return (x*xmax - x*xmin) - (0.5*xmax - 0.5*xmin) + (0.5*xmin + 0.5*xmax);
}

int NeighborIndex(int ix, int iy, int iz, int n1, int n2) {      // nghbr_index.hpp:28-54
  if (std::abs(ix) + std::abs(iy) + std::abs(iz) == 0 || std::abs(ix*iy*iz) > 1) return -1;
  if (iz == 0) {
    if (ix*iy == 0) return std::abs(ix)*2*(ix + 1) + std::abs(iy)*2*(iy + 5) + n1 + 2*n2;
    return 16 + (ix + 1) + 2*(iy + 1) + n1;
  }
  if (ix*iy == 0) return 24 + std::abs(ix)*(ix + 9) + std::abs(iy)*(iy + 17) + 2*(iz + 1) + n1 + 2*n2;
  return 48 + (ix + 1)/2 + (iy + 1) + 2*(iz + 1);
}

// ---- MeshBlockTree ---------------------------------------------------------------------------
MeshBlockTree::MeshBlockTree(const int nmb_root[3], const bool periodic[6], int ndim) : ndim_(ndim) {
  for (int d = 0; d < 3; ++d) nmb_root_[d] = nmb_root[d];
  for (int q = 0; q < 6; ++q) periodic_[q] = periodic[q];
  nleaf_ = 1 << ndim;
  int nmax = std::max(nmb_root[0], std::max(nmb_root[1], nmb_root[2]));
  root_level = 0;
  while ((1 << root_level) < nmax) ++root_level;                  // build_tree.cpp:44
  root_.reset(new Node{LogicalLocation{0, 0, 0, 0}, {}, -1});
  CreateRoot(root_.get());
}

MeshBlockTree::Node *MeshBlockTree::MakeChild(Node *node, int n) {
  const LogicalLocation &l = node->lloc;
  return new Node{LogicalLocation{l.lx1*2 + (n & 1), l.lx2*2 + ((n >> 1) & 1), l.lx3*2 + ((n >> 2) & 1),
                                  l.level + 1}, {}, -1};
}

void MeshBlockTree::CreateRoot(Node *node) {     // CreateRootGrid: the root grid may be incomplete
  if (node->lloc.level == root_level) return;
  node->leaf.resize(nleaf_);
  const int levfac = 1 << (root_level - node->lloc.level - 1);
  for (int n = 0; n < nleaf_; ++n) {
    std::unique_ptr<Node> c(MakeChild(node, n));
    if (c->lloc.lx3*levfac < nmb_root_[2] && c->lloc.lx2*levfac < nmb_root_[1] &&
        c->lloc.lx1*levfac < nmb_root_[0]) {
      CreateRoot(c.get());
      node->leaf[n] = std::move(c);
    }
  }
}

static int LeafIndex(const LogicalLocation &r, int level) {
  const int sh = r.level - level - 1;
  return ((r.lx1 >> sh) & 1) + (((r.lx2 >> sh) & 1) << 1) + (((r.lx3 >> sh) & 1) << 2);
}

void MeshBlockTree::AddNode(const LogicalLocation &rloc) {
  Node *node = root_.get();
  while (node->lloc.level != rloc.level) {
    if (node->leaf.empty()) Refine(node);
    node = node->leaf[LeafIndex(rloc, node->lloc.level)].get();
    if (node == nullptr) AKMI_FATAL("MeshBlockTree::AddNode outside the root grid");
  }
}

bool MeshBlockTree::Wrap(int &l, int d, int level) const {
  const int n = nmb_root_[d] << (level - root_level);
  if (l < 0) { if (!periodic_[2*d]) return false; l = n - 1; }
  else if (l >= n) { if (!periodic_[2*d + 1]) return false; l = 0; }
  return true;
}

void MeshBlockTree::Refine(Node *node) {         // with the 2:1 rule: the same-level neighbours exist afterwards
  if (!node->leaf.empty()) return;
  node->leaf.resize(nleaf_);
  for (int n = 0; n < nleaf_; ++n) node->leaf[n].reset(MakeChild(node, n));
  const LogicalLocation l = node->lloc;
  const int r1 = 1, r2 = ndim_ > 1 ? 1 : 0, r3 = ndim_ > 2 ? 1 : 0;
  for (int oz = -r3; oz <= r3; ++oz) {
    int z = l.lx3 + oz;
    if (!Wrap(z, 2, l.level)) continue;
    for (int oy = -r2; oy <= r2; ++oy) {
      int y = l.lx2 + oy;
      if (!Wrap(y, 1, l.level)) continue;
      for (int ox = -r1; ox <= r1; ++ox) {
        if (ox == 0 && oy == 0 && oz == 0) continue;
        int x = l.lx1 + ox;
        if (!Wrap(x, 0, l.level)) continue;
        AddNode(LogicalLocation{x, y, z, l.level});
      }
    }
  }
  node->gid = -1;
}

void MeshBlockTree::Walk(Node *node, std::vector<LogicalLocation> &out) {
  if (node->leaf.empty()) { node->gid = static_cast<int>(out.size()); out.push_back(node->lloc); return; }
  for (auto &c : node->leaf) if (c) Walk(c.get(), out);
}
std::vector<LogicalLocation> MeshBlockTree::CreateZOrderedLLList() {
  std::vector<LogicalLocation> out;
  Walk(root_.get(), out);
  return out;
}

// the block touching myloc in direction (ox1,ox2,ox3): itself when it is a leaf of the same or the
// coarser level, its parent node when the neighbours are finer; nullptr at a mesh boundary
MeshBlockTree::Node *MeshBlockTree::FindNeighbor(const LogicalLocation &my, int ox1, int ox2, int ox3) {
  const int ll = my.level;
  int lx = my.lx1 + ox1, ly = my.lx2 + ox2, lz = my.lx3 + ox3;
  if (!Wrap(lx, 0, ll) || !Wrap(ly, 1, ll) || !Wrap(lz, 2, ll)) return nullptr;
  if (ll < 1) return root_.get();
  Node *bt = root_.get();
  for (int level = 0; level < ll; ++level) {
    if (bt->leaf.empty()) {
      if (level == ll - 1) return bt;                         // coarser neighbour
      AKMI_FATAL("Neighbor search failed; MeshBlockTree broken.");
    }
    const int sh = ll - level - 1;
    bt = bt->leaf[((lx >> sh) & 1) + (((ly >> sh) & 1) << 1) + (((lz >> sh) & 1) << 2)].get();
    if (bt == nullptr) AKMI_FATAL("Neighbor search failed; MeshBlockTree broken.");
  }
  return bt;
}

// ---- Mesh::BuildTreeFromScratch (static refinement) ----------------------------------------------
void Mesh::BuildTreeFromScratch(ParameterInput *pin) {
  const int ndim = three_d ? 3 : (multi_d ? 2 : 1);
  const int nmb_root[3] = {nmb_rootx1, nmb_rootx2, nmb_rootx3};
  const char *names[6] = {"ix1_bc", "ox1_bc", "ix2_bc", "ox2_bc", "ix3_bc", "ox3_bc"};
  bool periodic[6];
  for (int q = 0; q < 6; ++q) {
    const std::string v = pin->GetString("mesh", names[q]);
    periodic[q] = q < 2*ndim && (v == "periodic" || v == "shear_periodic");
  }
  ptree.reset(new MeshBlockTree(nmb_root, periodic, ndim));
  root_level = ptree->root_level;
  max_level = root_level;
  const int mbn[3] = {mb_indcs.nx1, mb_indcs.nx2, mb_indcs.nx3};
  for (int d = 0; d < ndim; ++d)
    if (mbn[d] % 2) AKMI_FATAL("Number of cells in MeshBlock must be divisible by 2 with SMR or AMR.");
  if (mb_indcs.ng % 2) AKMI_FATAL("Number of ghost cells must be divisible by two for SMR/AMR calculations");
  const Real xmin[3] = {mesh_size.x1min, mesh_size.x2min, mesh_size.x3min};
  const Real xmax[3] = {mesh_size.x1max, mesh_size.x2max, mesh_size.x3max};
  for (const std::string &name : pin->BlockNames()) {
    if (name.compare(0, 14, "refined_region") != 0) continue;
    Real rmin[3] = {xmin[0], xmin[1], xmin[2]}, rmax[3] = {xmax[0], xmax[1], xmax[2]};
    for (int d = 0; d < ndim; ++d) {
      rmin[d] = pin->GetReal(name, "x" + std::to_string(d + 1) + "min");
      rmax[d] = pin->GetReal(name, "x" + std::to_string(d + 1) + "max");
    }
    const int phy = pin->GetInteger(name, "level");
    if (phy < 1) AKMI_FATAL("<refined_region> level must be larger than 0 (root level=0)");
    for (int d = 0; d < 3; ++d) {
      if (rmin[d] > rmax[d]) AKMI_FATAL("Invalid <refined_region> (xmax < xmin in one direction).");
      if (rmin[d] < xmin[d] || rmax[d] > xmax[d]) AKMI_FATAL("<refined_region> must be fully contained within root mesh");
    }
    const int log = phy + root_level;
    max_level = std::max(max_level, log);
    int lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1};
    for (int d = 0; d < ndim; ++d) {                            // build_tree.cpp:139-205
      const int lxmax = nmb_root[d]*(1 << phy);
      int a = 0;
      while (a < lxmax && !(LeftEdgeXs(a + 1, lxmax, xmin[d], xmax[d]) > rmin[d])) ++a;
      int b = a;
      while (b < lxmax && !(LeftEdgeXs(b + 1, lxmax, xmin[d], xmax[d]) >= rmax[d])) ++b;
      if (a % 2 == 1) --a;
      if (b % 2 == 0) ++b;
      lo[d] = a; hi[d] = b;
    }
    for (int k = lo[2]; k < hi[2]; k += 2)
      for (int j = lo[1]; j < hi[1]; j += 2)
        for (int i = lo[0]; i < hi[0]; i += 2) ptree->AddNode(LogicalLocation{i, j, k, log});
  }
  lloc_tree = ptree->CreateZOrderedLLList();
  nmb_total = static_cast<int>(lloc_tree.size());
  lloc_eachmb.resize(3*nmb_total);
  for (int m = 0; m < nmb_total; ++m) {
    lloc_eachmb[3*m] = lloc_tree[m].lx1; lloc_eachmb[3*m + 1] = lloc_tree[m].lx2; lloc_eachmb[3*m + 2] = lloc_tree[m].lx3;
  }
}

// ---- MeshBlock::SetNeighbors with levels (meshblock.cpp:142-425), written once for all directions:
// the free directions of an offset index the sub-blocks -- of a finer neighbour all of them, of a
// coarser neighbour the one this block occupies on its parent; a coarser edge/corner neighbour exists
// only where this block sits in that corner of its parent.
void MeshBlock::SetNeighborsSMR(Mesh *pm) {
  nghbr_smr.assign(static_cast<size_t>(nmb)*56*3, -1);
  nghbr_smr_gid.assign(static_cast<size_t>(nmb)*56, -1);
  const int gids = mb_gid[0];
  MeshBlockTree &tree = *pm->ptree;
  const int ndim = pm->three_d ? 3 : (pm->multi_d ? 2 : 1);
  for (int m = 0; m < nmb; ++m) {
    const LogicalLocation &ll = pm->lloc_tree[mb_gid[m]];
    const int my[3] = {ll.lx1, ll.lx2, ll.lx3};
    int myf[3], myo[3];
    for (int d = 0; d < 3; ++d) { myf[d] = my[d] & 1; myo[d] = d < ndim ? (my[d] & 1)*2 - 1 : 0; }
    const int r[3] = {1, ndim > 1 ? 1 : 0, ndim > 2 ? 1 : 0};
    for (int oz = -r[2]; oz <= r[2]; ++oz) for (int oy = -r[1]; oy <= r[1]; ++oy) for (int ox = -r[0]; ox <= r[0]; ++ox) {
      const int o[3] = {ox, oy, oz};
      if (ox == 0 && oy == 0 && oz == 0) continue;
      MeshBlockTree::Node *nt = tree.FindNeighbor(ll, ox, oy, oz);
      if (nt == nullptr) continue;
      int fr[3], nfree = 0;
      for (int d = 0; d < 3; ++d) if (o[d] == 0) fr[nfree++] = d;
      auto slot = [&](const int off[3], int f1, int f2) { return NeighborIndex(off[0], off[1], off[2], f1, f2); };
      const int neg[3] = {-ox, -oy, -oz};
      auto set = [&](int n, const MeshBlockTree::Node *c, int dest) {
        int *q = &nghbr_smr[(static_cast<size_t>(m)*56 + n)*3];
        // index in this pack, or nmb for a block of another rank (then only tested for existence: the
        // segments of such neighbours are addressed through akmi_smr::soff/roff)
        q[0] = pm->rank_eachmb[c->gid] == pm->my_rank ? c->gid - gids : nmb;
        q[1] = c->lloc.level; q[2] = dest;
        nghbr_smr_gid[static_cast<size_t>(m)*56 + n] = c->gid;
      };
      if (!nt->leaf.empty()) {                                  // finer: every touching child
        const int nf1 = nfree > 0 ? (fr[0] < ndim ? 2 : 1) : 1, nf2 = nfree > 1 ? (fr[1] < ndim ? 2 : 1) : 1;
        for (int f2 = 0; f2 < nf2; ++f2) for (int f1 = 0; f1 < nf1; ++f1) {
          int idx[3];
          for (int d = 0; d < 3; ++d) idx[d] = o[d] != 0 ? 1 - (o[d] + 1)/2 : 0;
          if (nfree > 0) idx[fr[0]] = f1;
          if (nfree > 1) idx[fr[1]] = f2;
          const MeshBlockTree::Node *c = nt->leaf[idx[0] + (idx[1] << 1) + (idx[2] << 2)].get();
          set(slot(o, f1, f2), c, slot(neg, f1, f2));
        }
      } else if (nt->lloc.level == ll.level) {
        set(slot(o, 0, 0), nt, slot(neg, 0, 0));
      } else {                                                  // coarser
        if (nfree < 2) {
          bool corner = true;
          for (int d = 0; d < 3; ++d) if (o[d] != 0 && myo[d] != o[d]) corner = false;
          if (!corner) continue;
        }
        const int f1 = nfree > 0 ? myf[fr[0]] : 0, f2 = nfree > 1 ? myf[fr[1]] : 0;
        set(slot(o, f1, f2), nt, slot(neg, f1, f2));
      }
    }
  }
}

// ---- index ranges of the 56 buffers --------------------------------------------------------------
namespace {
struct Dir { int s, e, cs, ce, cnx, ng; bool act; };
enum Kind { SAME = 0, COAR = 1, FINE = 2, PROL = 3, FLXS = 4, FLXC = 5 };

// one direction of one table (buffs_cc.cpp:28-483, buffs_fc.cpp:29-937).  o: offset of the slot along
// the direction; f: the sub-block flag that applies to it; a: 1 where a face-field component has its
// extra face; st: 1 where an edge-field component is staggered; ml_oth: multilevel mesh and the slot
// is offset in another direction (face fields only)
void Interval(Kind kind, bool send, const Dir &d, int o, int f, int a, int st, bool ml_oth, int &lo, int &hi) {
  const int ng = d.ng;
  if (kind == FLXS || kind == FLXC) {
    if (send && kind == FLXC) { lo = d.cs; hi = d.ce; } else { lo = d.s; hi = d.e; }
    if (o == 0) {
      hi += st;
      if (!send && kind == FLXC && d.act) { if (f == 1) lo += d.cnx; else hi -= d.cnx; }
      return;
    }
    const int edge = o > 0 ? hi + 1 : lo;
    lo = hi = edge;
    return;
  }
  if (send) {
    if (kind == COAR) { lo = d.cs; hi = d.ce; } else { lo = d.s; hi = d.e; }
    if (o == 0) {
      hi += a;
      if (kind == FINE && d.act) { if (f == 1) lo += d.cnx - ng; else hi -= d.cnx - ng; }
    } else if (o > 0) {
      lo = hi - ng + 1;
      if (kind == FINE) hi += a; else if (a && ml_oth) hi += 1;
    } else {
      hi = lo + ng - 1;
      if (kind == FINE) hi += a;
      else { lo += a; hi += a; if (a && ml_oth) lo -= 1; }
    }
    return;
  }
  const int n = kind == PROL ? ng/2 : ng;
  if (kind == COAR || kind == PROL) { lo = d.cs; hi = d.ce; } else { lo = d.s; hi = d.e; }
  if (o == 0) {
    hi += a;
    if (d.act) {
      if (kind == COAR || kind == PROL) { if (f == 0) hi += n; else lo -= n; }
      else if (kind == FINE) { if (f == 1) lo += d.cnx; else hi -= d.cnx; }
    }
  } else if (o > 0) {
    lo = hi + 1 + a; hi = hi + n + a;
    if (kind == COAR) lo -= a;
    else if ((kind == SAME || kind == FINE) && a && ml_oth) lo -= 1;
  } else {
    hi = lo - 1; lo = lo - n;
    if (kind == COAR) hi += a;
    else if ((kind == SAME || kind == FINE) && a && ml_oth) hi += 1;
  }
}
}  // namespace

MeshBoundaryValuesSMR::MeshBoundaryValuesSMR(MeshBlockPack *pp, int nvar_) : pmy_pack(pp), nvar(nvar_) {
  Mesh *pm = pp->pmesh;
  MeshBlock *pmb = pp->pmb;
  const RegionIndcs &in = pm->mb_indcs;
  const int ndim = pm->three_d ? 3 : (pm->multi_d ? 2 : 1);
  const int nmb = pp->nmb_thispack;
  const bool ml = true;
  nnghbr = ndim == 3 ? 56 : (ndim == 2 ? 24 : 8);
  const int ng = in.ng;
  const int nx[3] = {in.nx1, in.nx2, in.nx3};
  Dir D[3];
  for (int d = 0; d < 3; ++d) {
    const bool act = d == 0 || nx[d] > 1;
    const int s = act ? ng : 0, cnx = act ? nx[d]/2 : 1;
    D[d] = Dir{s, act ? ng + nx[d] - 1 : 0, s, act ? ng + cnx - 1 : 0, cnx, ng, act};
  }
  std::vector<int> cc(2*6*56*3*6, 0), fc(2*6*56*3*6, 0), ndat(2*56*2*5, 0), slot_ox(56*3, 0);
  auto at = [](std::vector<int> &t, int sr, int kind, int n, int v) { return &t[((((size_t)sr*6 + kind)*56 + n)*3 + v)*6]; };
  // InitializeBuffers, bvals.cpp:322-439: (slot, ox1, ox2, ox3, f1, f2)
  struct Slot { int n, o[3], f1, f2; };
  std::vector<Slot> slots;
  const int nfx = 2, nfy = ndim > 1 ? 2 : 1, nfz = ndim > 2 ? 2 : 1;
  for (int n = -1; n <= 1; n += 2) for (int fz = 0; fz < nfz; ++fz) for (int fy = 0; fy < nfy; ++fy)
    slots.push_back({NeighborIndex(n, 0, 0, fy, fz), {n, 0, 0}, fy, fz});
  if (ndim > 1) {
    for (int m = -1; m <= 1; m += 2) for (int fz = 0; fz < nfz; ++fz) for (int fx = 0; fx < nfx; ++fx)
      slots.push_back({NeighborIndex(0, m, 0, fx, fz), {0, m, 0}, fx, fz});
    for (int m = -1; m <= 1; m += 2) for (int n = -1; n <= 1; n += 2) for (int fz = 0; fz < nfz; ++fz)
      slots.push_back({NeighborIndex(n, m, 0, fz, 0), {n, m, 0}, fz, 0});
  }
  if (ndim > 2) {
    for (int l = -1; l <= 1; l += 2) for (int fy = 0; fy < nfy; ++fy) for (int fx = 0; fx < nfx; ++fx)
      slots.push_back({NeighborIndex(0, 0, l, fx, fy), {0, 0, l}, fx, fy});
    for (int l = -1; l <= 1; l += 2) for (int n = -1; n <= 1; n += 2) for (int fy = 0; fy < nfy; ++fy)
      slots.push_back({NeighborIndex(n, 0, l, fy, 0), {n, 0, l}, fy, 0});
    for (int l = -1; l <= 1; l += 2) for (int m = -1; m <= 1; m += 2) for (int fx = 0; fx < nfx; ++fx)
      slots.push_back({NeighborIndex(0, m, l, fx, 0), {0, m, l}, fx, 0});
    for (int l = -1; l <= 1; l += 2) for (int m = -1; m <= 1; m += 2) for (int n = -1; n <= 1; n += 2)
      slots.push_back({NeighborIndex(n, m, l, 0, 0), {n, m, l}, 0, 0});
  }
  for (const Slot &sl : slots) {
    const int n = sl.n;
    for (int d = 0; d < 3; ++d) slot_ox[3*n + d] = sl.o[d];
    const int fl[3] = {sl.f1, sl.o[0] != 0 ? sl.f1 : sl.f2, (sl.o[0] != 0 && sl.o[1] != 0) ? sl.f1 : sl.f2};
    for (int isfc = 0; isfc < 2; ++isfc) {
      std::vector<int> &tab = isfc ? fc : cc;
      for (int sr = 0; sr < 2; ++sr) {
        for (int kind = 0; kind < 6; ++kind) {
          if (kind == SAME && (sl.f1 || sl.f2)) continue;
          if (kind == PROL && sr == 0) continue;
          if (kind == FLXS && !isfc) continue;
          for (int v = 0; v < (isfc ? 3 : 1); ++v) {
            int *b = at(tab, sr, kind, n, v);
            for (int d = 0; d < 3; ++d) {
              const int a = (isfc && v == d) ? 1 : 0, st = (isfc && v != d) ? 1 : 0;
              bool oth = false;
              for (int q = 0; q < 3; ++q) if (q != d && sl.o[q] != 0) oth = true;
              Interval(static_cast<Kind>(kind), sr == 0, D[d], sl.o[d], fl[d], a, st, isfc && ml && oth, b[2*d], b[2*d + 1]);
            }
          }
        }
        const int kinds5[5] = {SAME, COAR, FINE, FLXS, FLXC};
        for (int q = 0; q < 5; ++q) {
          const int kind = kinds5[q];
          if ((kind == SAME && (sl.f1 || sl.f2)) || (kind == FLXS && !isfc)) continue;
          int mx = 0;
          for (int v = 0; v < (isfc ? 3 : 1); ++v) {
            const int *b = at(tab, sr, kind, n, v);
            mx = std::max(mx, (b[1] - b[0] + 1)*(b[3] - b[2] + 1)*(b[5] - b[4] + 1));
          }
          ndat[((static_cast<size_t>(isfc)*56 + n)*2 + sr)*5 + q] = mx;
        }
      }
    }
  }
  // edge owners after Sum(same) / Zero(finer) / Sum(finer): flux_correct_fc.cpp:445-790
  std::vector<int> nflx(static_cast<size_t>(nmb)*48, 1);
  static const int fe[6][5] = {{0, 16, 20, 32, 36}, {4, 18, 22, 34, 38}, {8, 16, 18, 40, 44},
                               {12, 20, 22, 42, 46}, {24, 32, 34, 40, 42}, {28, 36, 38, 44, 46}};
  auto face_edges = [&](int n) -> const int * { for (auto &r : fe) if (r[0] == n) return r; return nullptr; };
  const std::vector<int> &ngh = pmb->nghbr_smr;
  for (int m = 0; m < nmb; ++m) {
    const int mylev = pmb->mb_lev[m];
    int *nf = &nflx[static_cast<size_t>(m)*48];
    auto add = [&](bool finer) {
      for (int n = 0; n < std::min(nnghbr, 48); ++n) {
        const int g = ngh[(static_cast<size_t>(m)*56 + n)*3], l = ngh[(static_cast<size_t>(m)*56 + n)*3 + 1];
        if (g < 0 || l < mylev || (l > mylev) != finer) continue;
        if (const int *r = face_edges(n)) { for (int q = 1; q < 5; ++q) nf[r[q]] += 1; }
        else if ((n >= 16 && n < 24) || (n >= 32 && n < 48)) nf[n] += 1;
      }
    };
    add(false);
    for (int n = 0; n < std::min(nnghbr, 48); ++n) {
      const int g = ngh[(static_cast<size_t>(m)*56 + n)*3], l = ngh[(static_cast<size_t>(m)*56 + n)*3 + 1];
      if (g < 0 || l <= mylev) continue;
      if (const int *r = face_edges(n)) { for (int q = 1; q < 5; ++q) nf[r[q]] = 0; }
      else if ((n >= 16 && n < 24) || (n >= 32 && n < 48)) nf[n] = 0;
    }
    add(true);
  }
  // receive buffers: slot after slot, nvar*max(ndat) doubles per block
  std::vector<long long> layout(4*56*2, 0);
  size_t sizes[4];
  const int cls_fc[4] = {0, 0, 1, 1}, cls_nv[4] = {nvar, nvar, 3, 3};
  for (int cls = 0; cls < 4; ++cls) {
    long long off = 0;
    for (int n = 0; n < 56; ++n) {
      int mx = 0;
      for (int sr = 0; sr < 2; ++sr) {
        const int *q = &ndat[((static_cast<size_t>(cls_fc[cls])*56 + n)*2 + sr)*5];
        if (cls == 0 || cls == 2) mx = std::max(mx, std::max(q[0], std::max(q[1], q[2])));
        else if (cls == 1) mx = std::max(mx, q[4]);
        else mx = std::max(mx, std::max(q[3], q[4]));
      }
      const long long stride = static_cast<long long>(cls_nv[cls])*mx;
      layout[(static_cast<size_t>(cls)*56 + n)*2] = off;
      layout[(static_cast<size_t>(cls)*56 + n)*2 + 1] = stride;
      off += stride*nmb;
    }
    sizes[cls] = static_cast<size_t>(std::max<long long>(off, 1));
  }
  // ---- ranks (bvals_smr.py _plan_ranks): per class one buffer [segments of neighbours in this pack,
  // by (slot, block)] [segments received, rank after rank] [segments sent, rank after rank]; inside a
  // message the segments are ordered by (receiver gid, receiver slot), each of the size of the
  // receiver's slot
  {
    const int gids = pp->gids;
    struct Rem { int m, n, gid, lev, dest, rank; };
    std::vector<Rem> remote;
    for (int m = 0; m < nmb; ++m)
      for (int n = 0; n < 56; ++n) {
        const int g = pmb->nghbr_smr_gid[static_cast<size_t>(m)*56 + n];
        if (g < 0 || pm->rank_eachmb[g] == pm->my_rank) continue;
        const int *q = &ngh[(static_cast<size_t>(m)*56 + n)*3];
        remote.push_back({m, n, g, q[1], q[2], pm->rank_eachmb[g]});
      }
    for (const Rem &r : remote) if (std::find(peers.begin(), peers.end(), r.rank) == peers.end()) peers.push_back(r.rank);
    std::sort(peers.begin(), peers.end());
    auto carries = [](int cls, int lev_s, int lev_r, int dn) {
      if (cls == 0 || cls == 2) return true;
      if (cls == 1) return lev_s > lev_r && (dn < 16 || (dn >= 24 && dn < 32));
      return lev_s >= lev_r && dn < 48;
    };
    auto lay = [&](int cls, int n, int q) { return layout[(static_cast<size_t>(cls)*56 + n)*2 + q]; };
    std::vector<long long> soff(static_cast<size_t>(4)*nmb*56, 0), roff(static_cast<size_t>(4)*nmb*56, 0);
    for (int cls = 0; cls < 4; ++cls) {
      for (int m = 0; m < nmb; ++m)
        for (int n = 0; n < 56; ++n) {
          const int *q = &ngh[(static_cast<size_t>(m)*56 + n)*3];
          if (q[0] < 0) continue;
          if (q[0] < nmb) soff[(static_cast<size_t>(cls)*nmb + m)*56 + n] = lay(cls, q[2], 0) + q[0]*lay(cls, q[2], 1);
          roff[(static_cast<size_t>(cls)*nmb + m)*56 + n] = lay(cls, n, 0) + m*lay(cls, n, 1);
        }
      long long off = lay(cls, 55, 0) + lay(cls, 55, 1)*nmb;      // end of the in-pack segments
      for (int r : peers) {                       // what this rank receives from rank r
        const long long start = off;
        std::vector<std::array<int, 2>> items;
        for (const Rem &x : remote)
          if (x.rank == r && lay(cls, x.n, 1) > 0 && carries(cls, x.lev, pmb->mb_lev[x.m], x.n)) items.push_back({gids + x.m, x.n});
        std::sort(items.begin(), items.end());
        for (const auto &it : items) {
          roff[(static_cast<size_t>(cls)*nmb + (it[0] - gids))*56 + it[1]] = off;
          off += lay(cls, it[1], 1);
        }
        recv_slices[cls][r] = {start, off};
      }
      for (int r : peers) {                       // what it sends to rank r
        const long long start = off;
        std::vector<std::array<int, 4>> items;
        for (const Rem &x : remote)
          if (x.rank == r && lay(cls, x.dest, 1) > 0 && carries(cls, pmb->mb_lev[x.m], x.lev, x.dest))
            items.push_back({x.gid, x.dest, x.m, x.n});
        std::sort(items.begin(), items.end());
        for (const auto &it : items) {
          soff[(static_cast<size_t>(cls)*nmb + it[2])*56 + it[3]] = off;
          off += lay(cls, it[1], 1);
        }
        send_slices[cls][r] = {start, off};
      }
      sizes[cls] = static_cast<size_t>(std::max<long long>(off, 1));
    }
    if (!peers.empty()) {
      d_soff.Realloc(soff.size()); d_roff.Realloc(roff.size());
      HIPCHK(hipMemcpy(d_soff.p, soff.data(), sizeof(long long)*soff.size(), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(d_roff.p, roff.data(), sizeof(long long)*roff.size(), hipMemcpyHostToDevice));
    }
  }
  auto up_i = [](DvceArray<int> &d, const std::vector<int> &h) {
    d.Realloc(h.size());
    HIPCHK(hipMemcpy(d.p, h.data(), sizeof(int)*h.size(), hipMemcpyHostToDevice));
  };
  up_i(d_nghbr, ngh); up_i(d_lev, pmb->mb_lev); up_i(d_cc, cc); up_i(d_fc, fc); up_i(d_ndat, ndat);
  up_i(d_ox, slot_ox); up_i(d_nflx, nflx);
  d_layout.Realloc(layout.size());
  HIPCHK(hipMemcpy(d_layout.p, layout.data(), sizeof(long long)*layout.size(), hipMemcpyHostToDevice));
  for (int cls = 0; cls < 4; ++cls) {
    buf[cls].Realloc(sizes[cls]);
    HIPCHK(hipMemset(buf[cls].p, 0, sizeof(Real)*sizes[cls]));
  }
  // same-level neighbours in this pack: one direct gather (the uniform-mesh kernel with a 27-direction table) instead
  // of pack -> buffer -> unpack; akmi_smr::direct_same makes the SMR kernels skip those slots (AKMI_SMR_DIRECT=0: off)
  {
    std::vector<int> same(static_cast<size_t>(nmb)*27, -1);
    for (int m = 0; m < nmb; ++m)
      for (int o3 = -1; o3 <= 1; ++o3) for (int o2 = -1; o2 <= 1; ++o2) for (int o1 = -1; o1 <= 1; ++o1) {
        if ((o1 == 0 && o2 == 0 && o3 == 0) || (ndim < 3 && o3) || (ndim < 2 && o2)) continue;
        const int *q = &ngh[(static_cast<size_t>(m)*56 + NeighborIndex(o1, o2, o3, 0, 0))*3];
        if (q[0] >= 0 && q[0] < nmb && q[1] == pmb->mb_lev[m]) same[static_cast<size_t>(m)*27 + (o3 + 1)*9 + (o2 + 1)*3 + (o1 + 1)] = q[0];
      }
    up_i(d_same, same);
    std::vector<unsigned char> needs(nmb, 0);
    for (int m = 0; m < nmb; ++m)
      for (int n = 0; n < 56; ++n) {
        const int *q = &ngh[(static_cast<size_t>(m)*56 + n)*3];
        if (q[0] >= 0 && q[1] < pmb->mb_lev[m]) needs[m] = 1;
      }
    d_needs.Realloc(needs.size());
    HIPCHK(hipMemcpy(d_needs.p, needs.data(), needs.size(), hipMemcpyHostToDevice));
    smr_c.needs_coarse = d_needs.p;
    const char *e = std::getenv("AKMI_SMR_DIRECT");
    smr_c.direct_same = (e && std::atoi(e) == 0) ? 0 : 1;
  }
  smr_c.nnghbr = nnghbr; smr_c.multilevel = 1;
  smr_c.soff = peers.empty() ? nullptr : d_soff.p;   // one rank: layout[] addresses the buffers
  smr_c.roff = peers.empty() ? nullptr : d_roff.p;
  smr_c.nghbr = d_nghbr.p; smr_c.mblev = d_lev.p; smr_c.cc_tab = d_cc.p; smr_c.fc_tab = d_fc.p;
  smr_c.ndat = d_ndat.p; smr_c.slot_ox = d_ox.p; smr_c.layout = d_layout.p;
}

// Work lists (akmi_smr::lists): the SMR kernels are launched over the (block, slot) pairs that can have work instead
// of all nmb*56; built once by the library from the tables.  AKMI_SMR_LISTS=0: A/B switch.
void MeshBoundaryValuesSMR::BuildLists(const akmi_pack *pk, hipStream_t st) {
  const char *e = std::getenv("AKMI_SMR_LISTS");
  if (e && std::atoi(e) == 0) return;
  d_lists.Realloc(static_cast<size_t>(2)*pk->nmb*56*AKMI_SMR_NLISTS);
  int cnt[AKMI_SMR_NLISTS];
  if (akmi_smr_build_lists(pk, &smr_c, d_lists.p, cnt, st) != AKMI_COMPLETE) AKMI_FATAL(std::string(akmi_last_error()));
  smr_c.lists = d_lists.p;
  for (int l = 0; l < AKMI_SMR_NLISTS; ++l) smr_c.list_cnt[l] = cnt[l];
}

// The copy lists are an optimisation of the pack / unpack kernels, which stay in the library: a list that cannot be built
// (index space of 2^31 elements and more, no memory for the set-up scratch, a pair the walk cannot classify) is reported
// once and the exchange falls back to the kernels.  AKMI_SMR_FC_MAP / AKMI_SMR_CC_MAP = 0: never build the list (A/B
// switch); = 1: the list is REQUIRED (a failure is fatal -- what the tests of the lists themselves ask for).
static int map_mode(const char *name) {            // -1: default (try, fall back), 0: off, 1: required
  const char *e = std::getenv(name);
  return e ? (std::atoi(e) != 0 ? 1 : 0) : -1;
}
static bool map_failed(const char *what, int mode) {
  const std::string msg = std::string(what) + ": " + akmi_last_error();
  if (mode == 1) AKMI_FATAL(msg);
  std::fprintf(stderr, "### WARNING %s -- falling back to the pack / unpack kernels\n", msg.c_str());
  return true;
}

void MeshBoundaryValuesSMR::BuildFcMaps(const akmi_pack *pk, hipStream_t st) {
  const int mode = map_mode("AKMI_SMR_FC_MAP");
  if (mode == 0) return;
  const long long nb = static_cast<long long>(buf[2].n);
  long long lo = nb;
  for (int r : peers) lo = std::min(lo, send_slices[2].at(r).first);
  for (int which = 0; which < 2; ++which) {
    long long tail = 0;
    const long long n = akmi_smr_fc_map(pk, &smr_c, buf[2].p, nb, lo, nb, which, nullptr, 0, &tail, st);
    if (n < 0) { map_failed("akmi_smr_fc_map", mode); return; }
    d_fc_map[which].Realloc(static_cast<size_t>(2*std::max<long long>(n, 1)));
    if (akmi_smr_fc_map(pk, &smr_c, buf[2].p, nb, lo, nb, which, d_fc_map[which].p, n, &tail, st) != n) {
      map_failed("akmi_smr_fc_map", mode); return;
    }
    fc_np[which] = n; fc_tail[which] = tail;
  }
  fc_map_on = true;
}

void MeshBoundaryValuesSMR::BuildCcMap(const akmi_pack *pk, hipStream_t st) {
  const int mode = map_mode("AKMI_SMR_CC_MAP");
  if (mode == 0 || !peers.empty()) return;
  const long long nb = static_cast<long long>(buf[0].n);
  long long tail = 0;
  const long long n = akmi_smr_cc_map(pk, &smr_c, nvar, d_same.p, buf[0].p, nb, nullptr, 0, &tail, st);
  if (n < 0) { map_failed("akmi_smr_cc_map", mode); return; }
  d_cc_map.Realloc(static_cast<size_t>(2*std::max<long long>(n, 1)));
  if (akmi_smr_cc_map(pk, &smr_c, nvar, d_same.p, buf[0].p, nb, d_cc_map.p, n, &tail, st) != n) {
    map_failed("akmi_smr_cc_map", mode); return;
  }
  cc_np = n; cc_tail = tail; cc_map_on = true;
}

MeshBoundaryValuesSMR::~MeshBoundaryValuesSMR() {
  d_lists.Free(); d_fc_map[0].Free(); d_fc_map[1].Free(); d_cc_map.Free();
  d_nghbr.Free(); d_lev.Free(); d_cc.Free(); d_fc.Free(); d_ndat.Free(); d_ox.Free(); d_nflx.Free(); d_same.Free(); d_needs.Free();
  d_layout.Free(); d_soff.Free(); d_roff.Free();
  for (auto &b : buf) b.Free();
}

// the messages of one class: one per peer rank (bvals.cpp:134-310)
void MeshBoundaryValuesSMR::Post(int cls, hipStream_t st) {
  if (peers.empty()) return;
  std::vector<Comm::Msg> m;
  for (int r : peers) {
    const auto s = send_slices[cls].at(r), v = recv_slices[cls].at(r);
    m.push_back({r, buf[cls].p + s.first, s.second - s.first, buf[cls].p + v.first, v.second - v.first});
  }
  Comm::World().Post(m, st, cls);
}
void MeshBoundaryValuesSMR::Wait(int cls, hipStream_t st) {
  if (!peers.empty()) Comm::World().Wait(st, cls);
}

}  // namespace host
}  // namespace akmi
