// Host-side construction of a sector operator: basis maps, one-spin hopping tables (CSR in the
// reference's row order + compact ELL for the device), separable diagonal.
//
// Semantics follow the reference's stored-sparse path (never its buggy "direct" path,
// SURVEY.md 0.6):
//   basis            ED_SETUP.f90:720-775        orbital p <-> bit p-1, ascending integers
//   fermionic sign   ED_SETUP.f90:807-833        (-1)^(# occupied orbitals below pos)
//   one-spin hops    ED_HAMILTONIAN/sparse/H_up.f90:1-89, H_dw.f90:1-89
//   diagonal         ED_HAMILTONIAN/sparse/H_local.f90:1-102
//   DimDw split      ED_HAMILTONIAN.f90:93-105
// The construction itself is our own: the one-body part is first reduced to a list of directed
// orbital hops (a <- b, amplitude), then applied to every basis state with bit arithmetic.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <thread>

#include "hxv_internal.hpp"

namespace hxv {

namespace {

int64_t binom(int n, int k) {
  if (k < 0 || k > n) return 0;
  k = std::min(k, n - k);
  int64_t r = 1;
  for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
  return r;
}

// ascending list of ns-bit integers with popcount n (Gosper's hack)
std::vector<uint32_t> make_map(int ns, int n) {
  std::vector<uint32_t> map;
  map.reserve((size_t)binom(ns, n));
  if (n == 0) {
    map.push_back(0);
    return map;
  }
  uint64_t x = (1ull << n) - 1, lim = 1ull << ns;
  while (x < lim) {
    map.push_back((uint32_t)x);
    uint64_t c = x & (~x + 1), r = x + c;
    x = (((r ^ x) >> 2) / c) | r;
  }
  return map;
}

struct Hop {
  int a, b;  // a <- b (0-based orbitals)
  cplx t;
};

// Fortran-order accessors of the model arrays (0-based arguments)
struct ModelView {
  const hxv_model& m;
  int L, S, O, B;
  explicit ModelView(const hxv_model& mm) : m(mm), L(mm.nlat), S(mm.nspin), O(mm.norb), B(mm.nbath) {}
  size_t i6(int il, int jl, int is, int js, int io, int jo) const {
    return (size_t)il + (size_t)L * (jl + (size_t)L * (is + (size_t)S * (js + (size_t)S * (io + (size_t)O * jo))));
  }
  cplx hloc(int il, int jl, int s, int io, int jo) const {
    size_t k = i6(il, jl, s, s, io, jo);
    return cplx(m.imphloc[2 * k], m.imphloc[2 * k + 1]);
  }
  cplx hbath(int il, int jl, int s, int io, int jo, int ib) const {
    size_t k = i6(il, jl, s, s, io, jo) + (size_t)L * L * S * S * O * O * ib;
    return cplx(m.hbath[2 * k], m.hbath[2 * k + 1]);
  }
  double v(int il, int s, int io, int ib) const { return m.vbath[(size_t)il + (size_t)L * (s + (size_t)S * (io + (size_t)O * ib))]; }
  int imp(int il, int io) const { return io + il * O; }                         // ED_SETUP.f90:563-568
  int bath(int il, int io, int ib) const { return L * O + imp(il, io) + ib * L * O; }  // ED_SETUP.f90:367-375
};

// One-spin hop list in the reference's loop order (H_up.f90:8-28, :31-56, :60-87) and the
// orbital energies that enter the diagonal (H_local.f90:22-28, :85-93).
std::string one_body(const ModelView& mv, int spin, std::vector<Hop>& hops, std::vector<double>& eps) {
  int ns = mv.L * mv.O * (mv.B + 1);
  eps.assign(ns, 0.0);
  hops.clear();
  for (int il = 0; il < mv.L; ++il)
    for (int jl = 0; jl < mv.L; ++jl)
      for (int io = 0; io < mv.O; ++io)
        for (int jo = 0; jo < mv.O; ++jo) {
          cplx t = mv.hloc(il, jl, spin, io, jo);
          int a = mv.imp(il, io), b = mv.imp(jl, jo);
          if (a == b) {
            if (t.imag() != 0.0) return "impHloc has a complex diagonal element: not Hermitian";
            eps[a] = t.real();
          } else if (t != cplx(0, 0)) {
            hops.push_back({a, b, t});
          }
        }
  for (int ib = 0; ib < mv.B; ++ib)
    for (int il = 0; il < mv.L; ++il)
      for (int jl = 0; jl < mv.L; ++jl)
        for (int io = 0; io < mv.O; ++io)
          for (int jo = 0; jo < mv.O; ++jo) {
            cplx t = mv.hbath(il, jl, spin, io, jo, ib);
            int a = mv.bath(il, io, ib), b = mv.bath(jl, jo, ib);
            if (a == b)
              eps[a] = t.real();  // bath_diag = DREAL(...), ED_HAMILTONIAN_SPARSE_HxV.f90:71
            else if (t != cplx(0, 0))
              hops.push_back({a, b, t});
          }
  for (int il = 0; il < mv.L; ++il)
    for (int io = 0; io < mv.O; ++io)
      for (int ib = 0; ib < mv.B; ++ib) {
        double V = mv.v(il, spin, io, ib);
        if (V == 0.0) continue;
        int i = mv.imp(il, io), a = mv.bath(il, io, ib);
        hops.push_back({a, i, cplx(V, 0)});  // imp occupied, bath empty -> bath
        hops.push_back({i, a, cplx(V, 0)});  // bath occupied, imp empty -> imp
      }
  return "";
}

inline int parity_below(uint32_t m, int pos) { return __builtin_popcount(m & ((1u << pos) - 1u)) & 1; }

// H_sigma(i,j) for all basis states: rows collect (j, value) in ascending j, i.e. the
// reference's row-list order (it loops over the source state j outermost).  Entries are generated source-major and brought to
// rows by a stable counting sort; two hops that land on the same (row, column) -- only possible when the hop LIST holds the same
// ordered orbital pair twice -- are summed in list order where the first of them stands (ED_SPARSE_MATRIX.f90:267-273).
void apply_hops(const std::vector<uint32_t>& map, const std::vector<Hop>& hops_in, int ns, SpinOp& op) {
  const int dim = (int)map.size();
  op.dim = dim;
  // same (a <- b) twice in the list: one hop with the summed amplitude at the first one's place (sign flips are exact, so
  // +-t1 +-t2 summed per element is bit for bit +-(t1 + t2))
  std::vector<Hop> hops;
  hops.reserve(hops_in.size());
  for (const Hop& h : hops_in) {
    bool dup = false;
    for (Hop& g : hops)
      if (g.a == h.a && g.b == h.b) {
        g.t += h.t;
        dup = true;
        break;
      }
    if (!dup) hops.push_back(h);
  }
  // index of a configuration: a table over all ns-bit integers where that is small, else binary search in the sorted map
  std::vector<int32_t> index_of;
  if (ns <= 20) {
    index_of.assign((size_t)1 << ns, -1);
    for (int j = 0; j < dim; ++j) index_of[map[j]] = j;
  }
  struct Trip {
    int32_t i, j;
    cplx v;
  };
  std::vector<Trip> trips;
  trips.reserve((size_t)dim * 16);
  std::vector<int64_t> count(dim + 1, 0);
  for (int j = 0; j < dim; ++j) {
    const uint32_t m = map[j];
    for (const Hop& h : hops) {
      if (!((m >> h.b) & 1u) || ((m >> h.a) & 1u)) continue;
      const uint32_t m1 = m & ~(1u << h.b);
      const int sgn = parity_below(m, h.b) ^ parity_below(m1, h.a);
      const uint32_t m2 = m1 | (1u << h.a);
      const int i = index_of.empty() ? (int)(std::lower_bound(map.begin(), map.end(), m2) - map.begin()) : index_of[m2];
      trips.push_back({i, j, sgn ? -h.t : h.t});
      ++count[i + 1];
    }
  }
  op.rowptr.assign(dim + 1, 0);
  for (int i = 0; i < dim; ++i) op.rowptr[i + 1] = op.rowptr[i] + count[i + 1];
  op.cols.resize(trips.size());
  op.vals.resize(trips.size());
  std::vector<int64_t> fill(op.rowptr.begin(), op.rowptr.end() - 1);
  for (const Trip& t : trips) {
    const int64_t p = fill[t.i]++;
    op.cols[p] = t.j;
    op.vals[p] = t.v;
  }
}

// ---- device row order (SectorHost::up_perm) ----------------------------------------------------------------------------------------
// The orbital -> bit assignment: the high orbitals of pass A's default prefix blocks stay where the reference has them; the LOW orbitals are
// sorted by the number of hops that tie them to a high orbital, least tied first (stable: ties keep the reference's order).  A low orbital
// at bit L-1-d hops into a high one from 2^d runs of consecutive rows per block pair (d = 0: one run, a half block), so the tied orbitals
// want the top low bits.  C3 (2x2 + 3 replicas, L = 12): high = replica 3, tied = the four cluster sites -> bits 8..11, replicas 1, 2 -> 0..7.
// C4 (BHZ): the reference's numbering is already sorted this way -> no row order.
// Three steps, so that the one-spin matrix between device rows is built BESIDE the two reference matrices (a cold open of an Ns=16 sector stays
// near 20 ms): plan_row_order decides the order and relabels the basis (before the matrices are built), build_row_order_matrix is the third
// worker of the matrix build, finish_row_order carries the per-row tables over once they exist.
void plan_row_order(SectorHost& s, const std::vector<Hop>& hops_up, int ns, bool nd_tables_ok) {
  // (test hooks: HXV_ROW_ORDER_MIN_DIMUP lowers the size below which the reference's order is kept [2048: one block or two, nothing to gain];
  //  HXV_ROW_ORDER_BITS chooses the order for that many block bits instead of the default plan's -- the suite sets the handle's "tile_bits_up"
  //  to the same value -- so that small sectors exercise the row order against the oracle)
  const char* emin = std::getenv("HXV_ROW_ORDER_MIN_DIMUP");
  const char* ebits = std::getenv("HXV_ROW_ORDER_BITS");
  if (!row_order_enabled() || s.dimup < (emin ? std::atoi(emin) : 2048) || ns > 24) return;
  if (!nd_tables_ok) return;  // (the table-free spH0nd kernel searches the sorted reference basis)
  std::vector<cplx> amps;
  for (const Hop& h : hops_up) {
    const cplx a = (h.t.real() < 0 || (h.t.real() == 0 && h.t.imag() < 0)) ? -h.t : h.t;
    if (std::find(amps.begin(), amps.end(), a) == amps.end()) amps.push_back(a);
  }
  const int L = ebits ? std::atoi(ebits) : default_lowbits_up(ns, s.nup, (int)amps.size());
  if (L <= 0 || L >= ns) return;  // one block: nothing leaves it
  std::vector<int> deg(ns, 0);
  for (const Hop& h : hops_up) {
    if (h.a >= L && h.b < L) deg[h.b]++;
    if (h.b >= L && h.a < L) deg[h.a]++;
  }
  std::vector<int> low(L);
  for (int o = 0; o < L; ++o) low[o] = o;
  std::stable_sort(low.begin(), low.end(), [&](int x, int y) { return deg[x] < deg[y]; });
  std::vector<int32_t> pos(ns);
  bool ident = true;
  for (int b = 0; b < L; ++b) {
    pos[low[b]] = b;
    ident = ident && low[b] == b;
  }
  for (int o = L; o < ns; ++o) pos[o] = o;
  if (ident) return;
  const int dim = s.dimup;
  // device configurations: relabel every reference configuration, sort
  std::vector<uint32_t> key_of(dim);
  for (int i = 0; i < dim; ++i) {
    uint32_t m = s.map_up[i], x = 0;
    while (m) {
      const int o = __builtin_ctz(m);
      m &= m - 1;
      x |= 1u << pos[o];
    }
    key_of[i] = x;
  }
  std::vector<int32_t> iperm(dim);
  for (int i = 0; i < dim; ++i) iperm[i] = i;
  std::sort(iperm.begin(), iperm.end(), [&](int32_t a, int32_t b) { return key_of[a] < key_of[b]; });
  s.up_pos = pos;
  s.up_iperm = iperm;
  s.up_perm.assign(dim, 0);
  s.key_up.resize(dim);
  s.map_up_dev.resize(dim);
  s.up_sign.assign(dim, 0);
  // sign of a basis state: parity of the pairs of occupied orbitals whose order the relabelling reverses
  std::vector<std::pair<int, int>> inv;
  for (int a = 0; a < ns; ++a)
    for (int b = a + 1; b < ns; ++b)
      if (pos[a] > pos[b]) inv.push_back({a, b});
  for (int d = 0; d < dim; ++d) {
    const int i = iperm[d];
    s.up_perm[i] = d;
    s.key_up[d] = key_of[i];
    s.map_up_dev[d] = s.map_up[i];
    const uint32_t m = s.map_up[i];
    int par = 0;
    for (const auto& pr : inv) par ^= (int)((m >> pr.first) & (m >> pr.second) & 1u);
    s.up_sign[d] = (uint8_t)par;
  }
}

// H_up between device rows = H written with the relabelled orbitals on the sorted device configurations (its ELL tables are built with the
// other two matrices' in the next phase of the build)
void build_row_order_matrix(SectorHost& s, const std::vector<Hop>& hops_up, int ns) {
  std::vector<Hop> hd(hops_up);
  for (Hop& h : hd) {
    h.a = s.up_pos[h.a];
    h.b = s.up_pos[h.b];
  }
  apply_hops(s.key_up, hd, ns, s.up_dev);
}

// the per-row tables of the diagonal and of the spH0nd block by device row (they are built in the reference's order first)
void finish_row_order(SectorHost& s) {
  const int dim = s.dimup;
  s.a_up_dev.resize(dim);
  for (int d = 0; d < dim; ++d) s.a_up_dev[d] = s.a_up[s.up_iperm[d]];
  // the spH0nd move tables: rows and targets relabelled, signs carried over
  if (!s.nd_up.empty()) {
    const size_t nq = s.nd_up.size() / (size_t)dim;
    s.nd_up_dev.assign(s.nd_up.size(), ND_INVALID);
    for (size_t q = 0; q < nq; ++q)
      for (int i = 0; i < dim; ++i) {
        const uint32_t w = s.nd_up[q * dim + i];
        if (w == ND_INVALID) continue;
        const int j = (int)(w & 0x7FFFFFFFu), di = s.up_perm[i], dj = s.up_perm[j];
        const uint32_t sg = (w >> 31) ^ s.up_sign[di] ^ s.up_sign[dj];
        s.nd_up_dev[q * dim + di] = (uint32_t)dj | (sg << 31);
      }
  }
}

}  // namespace

bool row_order_enabled() {
  const char* e = std::getenv("HXV_ROW_ORDER");
  return !(e && e[0] == '0');
}
// what the sector-image cache must know of the environment: an image holds the tables of ONE device row order
std::string row_order_env_key() {
  std::string k = row_order_enabled() ? "R1" : "R0";
  for (const char* n : {"HXV_ROW_ORDER_MIN_DIMUP", "HXV_ROW_ORDER_BITS"}) {
    const char* v = std::getenv(n);
    k += "|";
    if (v) k += v;
  }
  return k;
}

void dw_split(int dimdw, int rank, int nranks, int& qdw, int& dw0) {
  // ED_HAMILTONIAN.f90:93-105
  int q = dimdw / nranks, rem = dimdw % nranks;
  qdw = q + (rank < rem ? 1 : 0);
  dw0 = rank * q + std::min(rank, rem);
}

// Padded all-gather layout (include/hxv.h, hxv_apply_device): every rank contributes cmax =
// ceil(DimDw/nranks) column slots, so rank r's slab starts at slot r*cmax; ranks that own one
// column less (ED_HAMILTONIAN.f90:93-98) leave their last slot unused.
void make_vcol(SectorHost& s) {
  s.pitch = (s.dimup + 7) & ~7;
  s.cmax = (s.dimdw + s.nranks - 1) / s.nranks;
  s.vcol.resize(s.dimdw);
  for (int r = 0; r < s.nranks; ++r) {
    int q, c0;
    dw_split(s.dimdw, r, s.nranks, q, c0);
    for (int c = 0; c < q; ++c) s.vcol[c0 + c] = (uint32_t)(r * s.cmax + c);
  }
}

namespace {
int g_exchange_default = -1;
}
int default_exchange() {
  if (g_exchange_default >= 0) return g_exchange_default;
  const char* e = std::getenv("HXV_EXCHANGE");
  if (e && std::string(e) == "alltoall") return 2;
  return (e && std::string(e) == "halo") ? 1 : 0;
}
void set_default_exchange(int mode) { g_exchange_default = mode; }

// The dw-only row panel [nrows x DimDw] that goes with an open sector (all-to-all exchange, ED_HAMILTONIAN_SPARSE_HxV.f90:272-296: the
// dw hops act on the transposed vector): same one-spin matrix H_dw, no basis of its own, no H_up, no diagonal.
std::string make_panel_host(const SectorHost& m, int nrows, SectorHost& p) {
  if (nrows < 1 || nrows > m.dimup) return "panel rows outside 1..DimUp";
  p = SectorHost();
  p.ns = m.ns;
  p.nup = m.nup;
  p.ndw = m.ndw;
  p.dimdw = m.dimdw;
  p.map_dw = m.map_dw;
  p.panel_rows = nrows;
  p.dimup = nrows;
  p.dim = (int64_t)nrows * m.dimdw;
  p.rank = 0;
  p.nranks = 1;
  dw_split(p.dimdw, 0, 1, p.qdw, p.dw0);
  p.ishift = 0;
  make_vcol(p);
  p.map_up.assign(nrows, 0u);
  p.up = SpinOp();
  p.up.dim = nrows;
  p.up.rowptr.assign(nrows + 1, 0);
  std::string e = build_ell(p.up);
  if (!e.empty()) return e;
  p.dw = m.dw;
  p.separable_diag = true;
  p.a_up.assign(nrows, 0.0);
  p.a_dw.assign(m.dimdw, 0.0);
  p.cross = m.cross;
  p.nd = NonLocalParams();
  return "";
}

// Halo layout: which columns of the other ranks do the rows of H_dw owned by each rank reference?  Every rank knows the
// whole one-spin matrix, so it derives its own receive list and what every other rank expects from it without talking.
void make_halo(SectorHost& s, const std::vector<int64_t>* more_ptr, const std::vector<int32_t>* more_cols) {
  // (more_ptr / more_cols: further columns each column's rows reference besides those of H_dw -- the dw moves of the spH0nd block)
  const int P = s.nranks;
  std::vector<int> owner(s.dimdw), first(P + 1, 0);
  for (int r = 0; r < P; ++r) {
    int q, c0;
    dw_split(s.dimdw, r, P, q, c0);
    first[r] = c0;
    for (int c = 0; c < q; ++c) owner[c0 + c] = r;
  }
  first[P] = s.dimdw;
  auto needed_by = [&](int r) {
    std::vector<char> mark(s.dimdw, 0);
    for (int c = first[r]; c < first[r + 1]; ++c) {
      for (int64_t p = s.dw.rowptr[c]; p < s.dw.rowptr[c + 1]; ++p)
        if (owner[s.dw.cols[p]] != r) mark[s.dw.cols[p]] = 1;
      if (more_ptr)
        for (int64_t p = (*more_ptr)[c]; p < (*more_ptr)[c + 1]; ++p)
          if (owner[(*more_cols)[p]] != r) mark[(*more_cols)[p]] = 1;
    }
    std::vector<int32_t> out;
    for (int c = 0; c < s.dimdw; ++c)
      if (mark[c]) out.push_back(c);
    return out;
  };
  s.halo_cols = needed_by(s.rank);
  s.halo_ptr.assign(P + 1, 0);
  for (int32_t c : s.halo_cols) s.halo_ptr[owner[c] + 1]++;
  for (int r = 0; r < P; ++r) s.halo_ptr[r + 1] += s.halo_ptr[r];
  s.send_cols.clear();
  s.send_ptr.assign(P + 1, 0);
  for (int r = 0; r < P; ++r) {
    if (r != s.rank)
      for (int32_t c : needed_by(r))
        if (owner[c] == s.rank) s.send_cols.push_back(c - s.dw0);
    s.send_ptr[r + 1] = (int32_t)s.send_cols.size();
  }
  // columns nobody here references keep slot 0: pass B's tile loads touch every column of a block that holds a local
  // column, but only referenced columns are ever gathered from the tile
  s.vcol.assign(s.dimdw, 0u);
  for (int c = 0; c < s.qdw; ++c) s.vcol[s.dw0 + c] = (uint32_t)c;
  for (size_t k = 0; k < s.halo_cols.size(); ++k) s.vcol[s.halo_cols[k]] = (uint32_t)(s.qdw + k);
  s.exchange = 1;
}

std::vector<uint32_t> translate_ell_src(const std::vector<uint32_t>& ell, const std::vector<uint32_t>& vcol) {
  std::vector<uint32_t> out(ell);
  for (auto& e : out)
    if (e != ELL_EMPTY) e = (e & ~ELL_SRC_MASK) | vcol[e & ELL_SRC_MASK];
  return out;
}

std::string build_ell(SpinOp& op) {
  if (op.dim > (1 << ELL_SRC_BITS)) return "spin-sector dimension exceeds 2^20 (ELL source index)";
  op.coef.clear();
  op.real_vals = true;
  int K = 0;
  for (int i = 0; i < op.dim; ++i) K = std::max<int>(K, (int)(op.rowptr[i + 1] - op.rowptr[i]));
  op.K = K;
  op.ell.assign((size_t)std::max(K, 1) * op.dim, ELL_EMPTY);
  // distinct |amplitudes| in order of first appearance (a handful: the last one found is tried first)
  int last = -1;
  for (int i = 0; i < op.dim; ++i) {
    int k = 0;
    for (int64_t p = op.rowptr[i]; p < op.rowptr[i + 1]; ++p, ++k) {
      cplx c = op.vals[p];
      uint32_t sign = 0;
      if (c.real() < 0 || (c.real() == 0 && c.imag() < 0)) {
        c = -c;
        sign = 1;
      }
      if (c.imag() != 0.0) op.real_vals = false;
      int id = -1;
      if (last >= 0 && op.coef[last] == c)
        id = last;
      else
        for (int q = 0; q < (int)op.coef.size(); ++q)
          if (op.coef[q] == c) {
            id = q;
            break;
          }
      if (id < 0) {
        id = (int)op.coef.size();
        if (id >= MAX_COEF) return "more than 1023 distinct hopping amplitudes";
        op.coef.push_back(c);
      }
      last = id;
      op.ell[(size_t)k * op.dim + i] = (uint32_t)op.cols[p] | ((uint32_t)id << ELL_SRC_BITS) | (sign << 31);
    }
  }
  if (op.coef.empty()) op.coef.push_back(cplx(0, 0));
  return "";
}

std::string build_sector_from_model(const hxv_model& m, int nup, int ndw, int rank, int nranks, SectorHost& s, int panel_rows) {
  if (m.nlat < 1 || m.norb < 1 || m.nspin < 1 || m.nspin > 2 || m.nbath < 0) return "bad Nlat/Norb/Nspin/Nbath";
  if (m.norb > 5) return "Norb > 5 (Uloc has 5 entries)";
  if (m.nlat > 16) return "Nlat > 16";
  if (!m.imphloc) return "imphloc is NULL";
  if (m.nbath > 0 && (!m.hbath || !m.vbath)) return "hbath/vbath is NULL with Nbath>0";
  int ns = m.nlat * m.norb * (m.nbath + 1);
  if (ns > 24) return "Ns > 24";
  if (nup < 0 || nup > ns || ndw < 0 || ndw > ns) return "nup/ndw outside [0,Ns]";
  if (nranks < 1 || rank < 0 || rank >= nranks) return "bad rank/nranks";
  ModelView mv(m);
  s.ns = ns; s.nup = nup; s.ndw = ndw;
  s.map_up = make_map(ns, nup);
  s.map_dw = make_map(ns, ndw);
  s.dimup = (int)s.map_up.size();
  s.dimdw = (int)s.map_dw.size();
  s.dim = (int64_t)s.dimup * s.dimdw;
  if (nranks > s.dimdw) return "nranks > DimDw: shrink the communicator first (ED_HAMILTONIAN.f90:63-89)";
  s.rank = rank; s.nranks = nranks;
  dw_split(s.dimdw, rank, nranks, s.qdw, s.dw0);
  s.ishift = (int64_t)s.dw0 * s.dimup;
  make_vcol(s);

  std::vector<Hop> hops_up, hops_dw;
  std::vector<double> eps_up, eps_dw;
  bool twin_spins = false;  // H_up and H_dw are the same matrix (Nspin = 1 or equal spin blocks, nup = ndw): built once
  std::string e = one_body(mv, 0, hops_up, eps_up);          // spin index 1       (H_up.f90:14)
  if (!e.empty()) return e;
  e = one_body(mv, m.nspin - 1, hops_dw, eps_dw);            // spin index Nspin   (H_dw.f90:14)
  if (!e.empty()) return e;
  if (panel_rows > 0) {
    // dw-only row panel [panel_rows x DimDw] of the vector (all-to-all exchange): the up index is just a row
    // count here -- no basis, no H_up, no diagonal
    if (panel_rows > s.dimup) return "panel_rows > DimUp";
    s.panel_rows = panel_rows;
    s.dimup = panel_rows;
    s.dim = (int64_t)panel_rows * s.dimdw;
    s.map_up.assign(panel_rows, 0u);
    s.ishift = 0;
    make_vcol(s);
    s.up = SpinOp();
    s.up.dim = panel_rows;
    s.up.rowptr.assign(panel_rows + 1, 0);
  } else {
  }
  {
    // the two one-spin matrices are independent: one host thread each (equal spins and fillings: one build, one copy)
    auto same_hops = [&]() {
      if (hops_up.size() != hops_dw.size()) return false;
      for (size_t q = 0; q < hops_up.size(); ++q)
        if (hops_up[q].a != hops_dw[q].a || hops_up[q].b != hops_dw[q].b || hops_up[q].t != hops_dw[q].t) return false;
      return true;
    };
    const bool twin = panel_rows == 0 && nup == ndw && same_hops();
    // the device row order is decided before the matrices are built, so that H_up between device rows is the third worker of this build
    if (panel_rows == 0) {
      const bool nd_on = m.norb > 1 && (m.jx != 0.0 || m.jp != 0.0);
      const bool nd_tables = (int64_t)m.nlat * m.norb * m.norb * (int64_t)std::max(s.dimup, s.dimdw) <= (int64_t)32 << 20;
      plan_row_order(s, hops_up, ns, !nd_on || nd_tables);
    }
    GuardedThread th, th_dev;   // (join on every path; their exceptions come back as error strings)
    if (panel_rows == 0 && !twin) th.run([&] { apply_hops(s.map_up, hops_up, ns, s.up); });
    if (s.row_order()) th_dev.run([&] { build_row_order_matrix(s, hops_up, ns); });
    try {
      apply_hops(s.map_dw, hops_dw, ns, s.dw);
    } catch (const std::exception& ex) {
      th.join();
      th_dev.join();
      return std::string("building H_dw: ") + ex.what();
    }
    th.join();
    th_dev.join();
    if (!th.err.empty()) return "building H_up: " + th.err;
    if (!th_dev.err.empty()) return "building H_up between device rows: " + th_dev.err;
    twin_spins = twin;
  }
  // (the spH0nd block reaches columns that H_dw does not: it keeps the all-gather layout)
  if (nranks > 1 && panel_rows == 0 && default_exchange() == 1) {
    if (m.norb > 1 && (m.jx != 0.0 || m.jp != 0.0)) {
      // the spH0nd block (sparse/H_non_local.f90:23-98) moves a dw electron between two orbitals of a site: its partner columns join the halo
      std::vector<int64_t> xp(s.dimdw + 1, 0);
      std::vector<int32_t> xc;
      for (int c = 0; c < s.dimdw; ++c) {
        const uint32_t m0 = s.map_dw[c];
        for (int il = 0; il < m.nlat; ++il)
          for (int x = 0; x < m.norb; ++x)
            for (int y = 0; y < m.norb; ++y) {
              if (x == y) continue;
              const int a = mv.imp(il, x), b = mv.imp(il, y);
              if (!((m0 >> a) & 1u) || ((m0 >> b) & 1u)) continue;
              const uint32_t m2 = (m0 & ~(1u << a)) | (1u << b);
              xc.push_back((int32_t)(std::lower_bound(s.map_dw.begin(), s.map_dw.end(), m2) - s.map_dw.begin()));
            }
        xp[c + 1] = (int64_t)xc.size();
      }
      make_halo(s, &xp, &xc);
    } else {
      make_halo(s);
    }
  }
  // (exchange 2 = the reference's two transposes: the all-gather layout stays, only the product's exchange differs)
  // (its row panels need a row for every rank: tiny sectors keep the all-gather)
  if (nranks > 1 && nranks <= s.dimup && panel_rows == 0 && default_exchange() == 2 && !(m.norb > 1 && (m.jx != 0.0 || m.jp != 0.0))) s.exchange = 2;
  {
    std::string e_up, e_dev;
    GuardedThread th, th_dev;
    if (!twin_spins) th.run([&] { e_up = build_ell(s.up); });
    if (s.row_order()) th_dev.run([&] { e_dev = build_ell(s.up_dev); });
    try {
      e = build_ell(s.dw);
    } catch (const std::exception& ex) {
      e = std::string("building the H_dw tables: ") + ex.what();
    }
    th.join();
    th_dev.join();
    if (!th.err.empty()) return "building the H_up tables: " + th.err;
    if (!th_dev.err.empty()) return "building the tables of H_up between device rows: " + th_dev.err;
    if (!e_dev.empty()) return "H_up in device row order: " + e_dev;
    if (!e_up.empty()) return e_up;
    if (!e.empty()) return e;
    if (twin_spins) s.up = s.dw;
  }

  // ---- diagonal: D(iup,idw) = a_up[iup] + a_dw[idw] + cross(mup,mdw)   (H_local.f90:21-93)
  const int L = m.nlat, O = m.norb, nimp = L * O;
  const double Ust = m.ust, Upp = m.ust - m.jh;
  double cst = 0.0;
  if (m.hfmode) {
    for (int il = 0; il < L; ++il)
      for (int io = 0; io < O; ++io) cst += 0.25 * m.uloc[io];
    if (O > 1)
      for (int il = 0; il < L; ++il)
        for (int io = 0; io < O; ++io)
          for (int jo = io + 1; jo < O; ++jo) cst += 0.25 * Ust + 0.25 * Upp;
  }
  auto one_spin_diag = [&](uint32_t mm, const std::vector<double>& eps) {
    double d = 0.0;
    for (int p = 0; p < ns; ++p)
      if ((mm >> p) & 1u) d += eps[p] - (p < nimp ? m.xmu : 0.0);
    for (int il = 0; il < L; ++il) {
      for (int io = 0; io < O; ++io) {
        double ni = (mm >> mv.imp(il, io)) & 1u;
        if (m.hfmode) d -= 0.5 * m.uloc[io] * ni;
        for (int jo = io + 1; jo < O; ++jo) {
          double nj = (mm >> mv.imp(il, jo)) & 1u;
          d += Upp * ni * nj;
          if (m.hfmode) d -= 0.5 * (Ust + Upp) * (ni + nj);
        }
      }
    }
    return d;
  };
  s.separable_diag = true;
  s.a_up.resize(s.dimup);
  s.a_dw.resize(s.dimdw);
  for (int i = 0; i < s.dimup; ++i) s.a_up[i] = panel_rows > 0 ? 0.0 : one_spin_diag(s.map_up[i], eps_up) + cst;
  for (int i = 0; i < s.dimdw; ++i) s.a_dw[i] = one_spin_diag(s.map_dw[i], eps_dw);
  s.nd = NonLocalParams();
  s.nd.active = (O > 1 && (m.jx != 0.0 || m.jp != 0.0)) ? 1 : 0;  // Jhflag, ED_SETUP.f90:200-201
  s.nd.nlat = L;
  s.nd.norb = O;
  s.nd.jx = m.jx;
  s.nd.jp = m.jp;
  s.nd_up.clear();
  s.nd_dw.clear();
  if (s.nd.active && panel_rows == 0 && (int64_t)L * O * O * (int64_t)std::max(s.dimup, s.dimdw) <= (int64_t)32 << 20) {
    auto moves = [&](const std::vector<uint32_t>& map, const std::vector<uint32_t>* slot, std::vector<uint32_t>& tab) {
      const int dim = (int)map.size();
      tab.assign((size_t)L * O * O * dim, ND_INVALID);
      for (int il = 0; il < L; ++il)
        for (int x = 0; x < O; ++x)
          for (int y = 0; y < O; ++y) {
            if (x == y) continue;
            const int a = mv.imp(il, x), b = mv.imp(il, y);  // c^+_b c_a
            uint32_t* row = &tab[(size_t)((il * O + x) * O + y) * dim];
            for (int i = 0; i < dim; ++i) {
              const uint32_t m0 = map[i];
              if (!((m0 >> a) & 1u) || ((m0 >> b) & 1u)) continue;
              const uint32_t m1 = m0 & ~(1u << a), m2 = m1 | (1u << b);
              const int sg = parity_below(m0, a) ^ parity_below(m1, b);
              const int j = (int)(std::lower_bound(map.begin(), map.end(), m2) - map.begin());
              row[i] = (slot ? (*slot)[j] : (uint32_t)j) | ((uint32_t)sg << 31);
            }
          }
    };
    moves(s.map_up, nullptr, s.nd_up);
    moves(s.map_dw, &s.vcol, s.nd_dw);
  }
  s.cross = CrossParams();
  s.cross.norb = O;
  s.cross.nlat = L;
  s.cross.ust = (O > 1) ? Ust : 0.0;
  for (int io = 0; io < O; ++io) {
    s.cross.uloc[io] = m.uloc[io];
    for (int il = 0; il < L; ++il) s.cross.orbmask[io] |= 1u << mv.imp(il, io);
  }
  for (int il = 0; il < L; ++il)
    for (int io = 0; io < O; ++io) s.cross.sitemask[il] |= 1u << mv.imp(il, io);
  if (s.row_order()) finish_row_order(s);
  return "";
}

double host_diag_element(const SectorHost& s, int iup, int idw) {
  if (!s.separable_diag) return s.diag_stored[(size_t)iup + (size_t)(idw - s.dw0) * s.dimup];
  uint32_t mu = s.map_up[iup], md = s.map_dw[idw], both = mu & md;
  double d = s.a_up[iup] + s.a_dw[idw];
  for (int io = 0; io < s.cross.norb; ++io) d += s.cross.uloc[io] * __builtin_popcount(both & s.cross.orbmask[io]);
  if (s.cross.norb > 1)
    for (int il = 0; il < s.cross.nlat; ++il) {
      uint32_t sm = s.cross.sitemask[il];
      d += s.cross.ust * (__builtin_popcount(mu & sm) * __builtin_popcount(md & sm) - __builtin_popcount(both & sm));
    }
  return d;
}

std::string build_sector_from_csr(int dimup, int dimdw, const int64_t* up_rp, const int32_t* up_cols, const double* up_vals,
                                  const int64_t* dw_rp, const int32_t* dw_cols, const double* dw_vals, const double* diag, int rank,
                                  int nranks, SectorHost& s) {
  if (dimup < 1 || dimdw < 1) return "bad DimUp/DimDw";
  if (!up_rp || !dw_rp || !diag) return "NULL CSR/diag pointer";
  if (nranks < 1 || rank < 0 || rank >= nranks || nranks > dimdw) return "bad rank/nranks";
  s.ns = 0; s.nup = s.ndw = -1;
  s.dimup = dimup; s.dimdw = dimdw;
  s.dim = (int64_t)dimup * dimdw;
  s.rank = rank; s.nranks = nranks;
  dw_split(dimdw, rank, nranks, s.qdw, s.dw0);
  s.ishift = (int64_t)s.dw0 * dimup;
  make_vcol(s);
  auto load = [](int dim, const int64_t* rp, const int32_t* cols, const double* vals, SpinOp& op) -> std::string {
    op.dim = dim;
    op.rowptr.assign(rp, rp + dim + 1);
    if (op.rowptr[0] != 0) return "rowptr[0] != 0";
    int64_t nnz = op.rowptr[dim];
    op.cols.resize((size_t)nnz);
    op.vals.resize((size_t)nnz);
    for (int i = 0; i < dim; ++i)
      if (op.rowptr[i + 1] < op.rowptr[i]) return "rowptr not monotone";
    for (int64_t p = 0; p < nnz; ++p) {
      if (cols[p] < 1 || cols[p] > dim) return "column index outside [1,dim]";
      op.cols[p] = cols[p] - 1;
      op.vals[p] = cplx(vals[2 * p], vals[2 * p + 1]);
    }
    return build_ell(op);
  };
  std::string e = load(dimup, up_rp, up_cols, up_vals, s.up);
  if (!e.empty()) return "H_up: " + e;
  e = load(dimdw, dw_rp, dw_cols, dw_vals, s.dw);
  if (!e.empty()) return "H_dw: " + e;
  s.separable_diag = false;
  size_t nloc = (size_t)s.qdw * dimup;
  s.diag_stored.resize(nloc);
  for (size_t i = 0; i < nloc; ++i) {
    if (diag[2 * i + 1] != 0.0) return "complex diagonal element: not Hermitian";
    s.diag_stored[i] = diag[2 * i];
  }
  // (the exchange of a split sector is chosen like for sectors opened from a model; a stored spH0nd block -- hxv_set_nonlocal_csr --
  //  needs the whole gathered vector and is refused on the other two)
  if (nranks > 1 && default_exchange() == 1) make_halo(s);
  if (nranks > 1 && nranks <= dimup && default_exchange() == 2) s.exchange = 2;
  return "";
}

}  // namespace hxv
