// Host-side construction of the owner-computes block plan (see saa_plan.h).  Pure C++, no HIP.
#include "saa_plan.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <thread>

namespace saa {
namespace {

// ------------------------------------------------------------------------------------------------
// node blocks: recursive coordinate bisection of the node cloud
// ------------------------------------------------------------------------------------------------
// The order in which a bisection along axis `ax` sees the nodes.  Plain: by coordinate, ties by node id (on a lattice numbered
// x fastest a cut inside a plane of nodes then takes a strip of it).  With `cell` (per node, three integers: the coordinates in
// units of the mesh size, rounded): by LAYER of the mesh size along the axis, inside a layer by the cells of the other two
// axes (lower axis first - the order of the ids on a lattice numbered z fastest, like the synthetic beams), then by coordinate and id.  A lattice whose nodes are
// displaced by less than half a cell and numbered at random is then cut like the lattice itself - along planes of nodes and
// strips of them - instead of through the cloud of displaced coordinates (ragged block faces: 9 % more halo nodes, 1.2 % more
// element copies); a mesh without any lattice is cut along layers of the mesh size, a coherent strip of the last layer going
// to either side.
struct AxisOrder {
  const double *xyz;
  const int32_t *cell;  // null: plain order
  bool operator()(int ax, int32_t a, int32_t b) const {
    if (cell) {
      const int32_t *ca = cell + 3 * static_cast<int64_t>(a), *cb = cell + 3 * static_cast<int64_t>(b);
      if (ca[ax] != cb[ax]) return ca[ax] < cb[ax];
      for (int o = 0; o < 3; ++o)
        if (o != ax && ca[o] != cb[o]) return ca[o] < cb[o];
    }
    const double va = xyz[3 * static_cast<int64_t>(a) + ax], vb = xyz[3 * static_cast<int64_t>(b) + ax];
    return va < vb || (va == vb && a < b);
  }
};

struct Rcb {
  const double *xyz;
  std::vector<int32_t> &order;          // node ids being permuted in place
  std::vector<int32_t> &block_start;    // filled leaf by leaf, in order
  const int64_t *weight = nullptr;      // per node (caller's id): work it brings to its block; null = 1
  const int32_t *cell = nullptr;        // per node: coordinates in mesh sizes, rounded (AxisOrder); null = plain order
  void split(int64_t lo, int64_t hi, int32_t nblk) {
    if (nblk <= 1) {
      // inside a block: lexicographic by coordinates.  On structured regions that makes the block-local index a
      // lattice index, so that translated copies of an element pair differ by one constant in all their vertex slots
      // - what the clash-free packing of reorder_for_lds is built on - whatever numbering the caller uses.
      const double *c = xyz;
      std::sort(order.begin() + lo, order.begin() + hi, [c](int32_t a, int32_t b) {
        const double *pa = c + 3 * static_cast<int64_t>(a), *pb = c + 3 * static_cast<int64_t>(b);
        if (pa[0] != pb[0]) return pa[0] < pb[0];
        if (pa[1] != pb[1]) return pa[1] < pb[1];
        if (pa[2] != pb[2]) return pa[2] < pb[2];
        return a < b;
      });
      block_start.push_back(static_cast<int32_t>(lo));
      return;
    }
    const int32_t left_blk = nblk / 2;
    const int64_t n = hi - lo;
    int64_t k = (n * left_blk + nblk / 2) / nblk;
    k = std::max<int64_t>(1, std::min<int64_t>(n - 1, k));
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int64_t i = lo; i < hi; ++i) {
      const double *p = xyz + 3 * static_cast<int64_t>(order[i]);
      for (int a = 0; a < 3; ++a) {
        mn[a] = std::min(mn[a], p[a]);
        mx[a] = std::max(mx[a], p[a]);
      }
    }
    int ax = 0;
    for (int a = 1; a < 3; ++a)
      if (mx[a] - mn[a] > mx[ax] - mn[ax]) ax = a;
    const AxisOrder ord{xyz, cell};
    auto before = [ord, ax](int32_t a, int32_t b) { return ord(ax, a, b); };
    if (weight) {  // cut where the left part holds its share of the WORK, not of the nodes
      std::sort(order.begin() + lo, order.begin() + hi, before);
      int64_t total = 0;
      for (int64_t i = lo; i < hi; ++i) total += weight[order[i]];
      const int64_t want = (total * left_blk + nblk / 2) / nblk;
      int64_t acc = 0, kk = 0;
      while (kk < n - 1 && acc + weight[order[lo + kk]] / 2 < want) acc += weight[order[lo + kk++]];
      k = std::max<int64_t>(left_blk, std::min<int64_t>(n - (nblk - left_blk), std::max<int64_t>(1, kk)));
    } else {
      std::nth_element(order.begin() + lo, order.begin() + lo + k, order.begin() + hi, before);
    }
    split(lo, lo + k, left_blk);
    split(lo + k, hi, nblk - left_blk);
  }
};

int32_t choose_block_count(int32_t n_nodes, int32_t block_nodes) {
  int64_t nb = (static_cast<int64_t>(n_nodes) + block_nodes - 1) / block_nodes;
  // MI355X has 256 CUs: once there is more than about a chip-full of blocks, make the count a
  // multiple of 256 so that every CU gets the same number of (equal-sized) blocks.
  if (nb > 192) nb = (nb + 255) / 256 * 256;
  nb = std::max<int64_t>(1, std::min<int64_t>(nb, n_nodes));
  return static_cast<int32_t>(nb);
}

// Automatic block size (measured on MI355X, pair kernel).  Larger blocks duplicate fewer border elements and
// stage fewer halo records per owned node (element copies 1.25x at ~740 owned nodes against 1.41x at ~380).
//   * up to ~360k nodes every CU gets exactly ONE block and all 256 run as a single wave of 1024-thread workgroups (1M
//     tets: 13.0 us/step; 512 blocks of 372: 16.3).  Up to ~215k nodes (840 per block on average, +14 % in the fullest)
//     the resident kernel's LDS image holds the block (~165 B per owned node of 160 KB): a 209k-node Delaunay mesh
//     runs resident at 9.0 us/step, a 215k-node one at 9.8 - round 3 stopped at 760 per block and sent them to 512 blocks
//     of 512 threads outside the resident kernel (15.5 / 15.7).  Beyond that the one-launch-per-step kernel takes the same
//     256 blocks (its image is 48 B per local + 24 B per owned node) and still beats the smaller blocks by 12 % (243k
//     nodes: 15.5 against 17.8 us, 255k: 16.6 against 18.7; tools/capacity_point.py);
//   * larger partitions take ~720-node blocks in several rounds of 512-thread workgroups, which overlap each
//     other's memory and LDS phases (8.2M tets: 84.6 us/step; 377-node blocks 91.5; 942-node blocks 118).
int32_t auto_block_nodes(int32_t n_nodes) {
  constexpr int32_t kCUs = 256, kBig = 1400;
  if (n_nodes <= kDefaultBlockNodes) return n_nodes;  // tiny mesh: one block
  if (n_nodes <= kCUs * kBig) return std::max<int32_t>(96, (n_nodes + kCUs - 1) / kCUs);
  return kLargeMeshBlockNodes;
}

// ------------------------------------------------------------------------------------------------
// work items: pairs of face-adjacent elements
// ------------------------------------------------------------------------------------------------
// The 12 even permutations of 4 vertices: re-ordering a tet by one of them keeps its orientation (the
// sign of detJ the reference keeps, Mat_construction.py:93) and therefore its nodal forces.
constexpr int kEvenPerms[12][4] = {{0, 1, 2, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {1, 0, 3, 2}, {1, 2, 0, 3}, {1, 3, 2, 0},
                                   {2, 0, 1, 3}, {2, 1, 3, 0}, {2, 3, 0, 1}, {3, 0, 2, 1}, {3, 1, 0, 2}, {3, 2, 1, 0}};

// Item vertex order (a, p, q, r, b): tets A = (a; p,q,r) and B = (b; p,r,q).  Relabellings that keep
// both orientations: rotations of (p,q,r) and the exchange A<->B (a<->b together with q<->r).
constexpr int kPairSyms[6][5] = {{0, 1, 2, 3, 4}, {0, 2, 3, 1, 4}, {0, 3, 1, 2, 4},
                                 {4, 1, 3, 2, 0}, {4, 3, 2, 1, 0}, {4, 2, 1, 3, 0}};

// Greedy matching of the block's elements into face-sharing pairs with compatible orientation.
// in: loc = 4 block-local node ids per element.  out: items (8 uint16 each); returns their number.
// xl / h (optional): coordinates of the block-local nodes and the mesh size, for the shape classes below.
int32_t build_items(const std::vector<uint16_t> &loc, int32_t n_elem, int32_t n_owned, std::vector<uint16_t> &items,
                    const double *xl = nullptr, double h = 0.0, bool augment = false) {
  struct Face {
    uint64_t key;
    int32_t elem;
    int32_t omit;
  };
  std::vector<Face> faces(4 * static_cast<size_t>(n_elem));
  for (int32_t e = 0; e < n_elem; ++e)
    for (int k = 0; k < 4; ++k) {
      uint16_t t[3];
      int m = 0;
      for (int a = 0; a < 4; ++a)
        if (a != k) t[m++] = loc[4 * static_cast<size_t>(e) + a];
      std::sort(t, t + 3);
      faces[4 * static_cast<size_t>(e) + k] = {
          (static_cast<uint64_t>(t[0]) << 32) | (static_cast<uint64_t>(t[1]) << 16) | t[2], e, k};
    }
  std::sort(faces.begin(), faces.end(),
            [](const Face &x, const Face &y) { return x.key < y.key || (x.key == y.key && x.elem < y.elem); });
  // nb[e][k] = element across the face opposite vertex k, nbk = the vertex IT omits; -1: none in this block
  std::vector<int32_t> nb(4 * static_cast<size_t>(n_elem), -1), nbk(4 * static_cast<size_t>(n_elem), 0);
  for (size_t i = 0; i + 1 < faces.size(); ++i) {
    if (faces[i].key != faces[i + 1].key) continue;
    if (i + 2 < faces.size() && faces[i + 2].key == faces[i].key) continue;  // non-manifold face: leave alone
    if (i > 0 && faces[i - 1].key == faces[i].key) continue;
    nb[4 * static_cast<size_t>(faces[i].elem) + faces[i].omit] = faces[i + 1].elem;
    nbk[4 * static_cast<size_t>(faces[i].elem) + faces[i].omit] = faces[i + 1].omit;
    nb[4 * static_cast<size_t>(faces[i + 1].elem) + faces[i + 1].omit] = faces[i].elem;
    nbk[4 * static_cast<size_t>(faces[i + 1].elem) + faces[i + 1].omit] = faces[i].omit;
  }
  auto apex_first = [&](int32_t e, int k, uint16_t out[4]) {  // even permutation with vertex k in front
    for (const auto &pm : kEvenPerms)
      if (pm[0] == k) {
        for (int a = 0; a < 4; ++a) out[a] = loc[4 * static_cast<size_t>(e) + pm[a]];
        return;
      }
  };
  // B = (b; x,y,z) pairs with A = (a; p,q,r) iff (x,y,z) is a rotation of (p,r,q)
  auto compatible = [](const uint16_t A[4], const uint16_t B[4]) {
    for (int t = 0; t < 3; ++t)
      if (B[1 + t] == A[1] && B[1 + (t + 1) % 3] == A[3] && B[1 + (t + 2) % 3] == A[2]) return true;
    return false;
  };
  // an element is interior if all its nodes are owned; pairs never mix the two classes (interior items run before
  // the halo records are in LDS)
  auto interior = [&](int32_t e) {
    const uint16_t *c = &loc[4 * static_cast<size_t>(e)];
    return c[0] < n_owned && c[1] < n_owned && c[2] < n_owned && c[3] < n_owned;
  };
  std::vector<char> used(n_elem, 0);
  auto free_degree = [&](int32_t e) {
    int d = 0;
    for (int k = 0; k < 4; ++k) {
      const int32_t f = nb[4 * static_cast<size_t>(e) + k];
      d += (f >= 0 && !used[f]);
    }
    return d;
  };
  // eligible partners of e across face k (free, same class, compatible orientation)
  auto partner = [&](int32_t e, int k) -> int32_t {
    const int32_t f = nb[4 * static_cast<size_t>(e) + k];
    if (f < 0 || used[f] || interior(f) != interior(e)) return -1;
    uint16_t A[4], B[4];
    apex_first(e, k, A);
    apex_first(f, nbk[4 * static_cast<size_t>(e) + k], B);
    return compatible(A, B) ? f : -1;
  };
  items.clear();
  items.reserve(8 * static_cast<size_t>(n_elem));
  int32_t n_items = 0;
  auto emit_pair = [&](int32_t e, int k) {
    const int32_t f = nb[4 * static_cast<size_t>(e) + k];
    uint16_t A[4], B[4];
    apex_first(e, k, A);
    apex_first(f, nbk[4 * static_cast<size_t>(e) + k], B);
    const uint16_t it[8] = {A[0], A[1], A[2], A[3], B[0], 1, 0, 0};
    items.insert(items.end(), it, it + 8);
    ++n_items;
  };
  auto emit_single = [&](int32_t e) {
    uint16_t it[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int a = 0; a < 4; ++a) it[a] = loc[4 * static_cast<size_t>(e) + a];
    it[4] = it[1];  // dummy second apex: a valid LDS index that is read but never accumulated
    items.insert(items.end(), it, it + 8);
    ++n_items;
  };
  // Greedy matching in a SPATIAL order of the elements (xl given), else in the order of the caller's list: an element
  // takes the free face neighbour that has the fewest other free neighbours left, ties broken by the pair's shape.
  //   * order: cell by cell of a grid of the mesh size h (cells lexicographic; an element belongs to the cell of the
  //     lowest corner of its bounding box), inside a cell by where the element's centroid sits in it - first the ORDER of
  //     its three offsets from the cell corner (six patterns: the six tets of a Kuhn cube), then the offsets in quarters
  //     of a cell - so that the tets of every cube come up in the same order and meet the same situation;
  //   * shape of a candidate pair: the vector between the two centroids in quarters of h and the vector between the two
  //     apices in units of h, made independent of which element is named first - coarse enough to be the same for nodes
  //     moved off their lattice positions by a fifth of a cell (in halves of h the keys of a jittered lattice were noise,
  //     and ties broken by noise pair equal cubes differently: 48 % of the items in pattern classes instead of 59 %).
  // In the order of the caller's list (what this loop used before) a mesh numbered at random left 5 % of its elements
  // single - 7 % more work items than the same mesh numbered cell by cell - and paired equal cells differently, so that
  // the pattern classes of the LDS packing could not form.  Measured alternatives on the 1M-tet beams (pairs found / items
  // in clash-free halves by construction, structured | jittered and shuffled): list order 99.9 % / 58 % | 94.8 % / 31 %;
  // this order 99.8 % / - | 99.2 % / 59 %; fewest-options-first (Karp-Sipser) with shape ties 99.8 % / 43 % | 99.5 % /
  // 37 %: best matching, but its fronts run inwards from the block faces and pair equal cells differently; shape by
  // shape, most frequent first: a third of the elements single.
  std::vector<uint64_t> ekey(4 * static_cast<size_t>(n_elem), 0), ekey_min;
  if (xl != nullptr && h > 0.0) {
    auto centroid = [&](int32_t e, double c[3]) {
      for (int j = 0; j < 3; ++j) {
        c[j] = 0.0;
        for (int a = 0; a < 4; ++a) c[j] += 0.25 * xl[3 * static_cast<size_t>(loc[4 * static_cast<size_t>(e) + a]) + j];
      }
    };
    for (int32_t e = 0; e < n_elem; ++e)
      for (int k = 0; k < 4; ++k) {
        const int32_t f = nb[4 * static_cast<size_t>(e) + k];
        if (f < 0 || f < e) continue;
        const int kf = nbk[4 * static_cast<size_t>(e) + k];
        double ce[3], cf[3];
        centroid(e, ce);
        centroid(f, cf);
        const double *ae = xl + 3 * static_cast<size_t>(loc[4 * static_cast<size_t>(e) + k]);
        const double *af = xl + 3 * static_cast<size_t>(loc[4 * static_cast<size_t>(f) + kf]);
        int q[6];
        for (int j = 0; j < 3; ++j) {
          q[j] = static_cast<int>(std::lround(4.0 * (cf[j] - ce[j]) / h));
          q[3 + j] = static_cast<int>(std::lround((af[j] - ae[j]) / h));
        }
        int sign = 0;
        for (int j = 0; j < 6 && sign == 0; ++j) sign = q[j] > 0 ? 1 : (q[j] < 0 ? -1 : 0);
        uint64_t key = 0;
        for (int j = 0; j < 6; ++j) key = (key << 8) | static_cast<uint64_t>(((sign < 0 ? -q[j] : q[j]) + 128) & 255);
        ekey[4 * static_cast<size_t>(e) + k] = key;
        ekey[4 * static_cast<size_t>(f) + kf] = key;
      }
  }
  std::vector<int32_t> order(n_elem);
  std::iota(order.begin(), order.end(), 0);
  std::vector<uint64_t> okey;  // cell of every element (spatial variant only)
  if (xl != nullptr && h > 0.0) {
    okey.resize(n_elem);
    double lo[3] = {1e300, 1e300, 1e300};
    for (int32_t e = 0; e < n_elem; ++e)
      for (int a = 0; a < 4; ++a)
        for (int j = 0; j < 3; ++j) lo[j] = std::min(lo[j], xl[3 * static_cast<size_t>(loc[4 * static_cast<size_t>(e) + a]) + j]);
    for (int32_t e = 0; e < n_elem; ++e) {
      // cell of the element: the cell of the lowest corner of its bounding box, with a quarter cell of slack for nodes
      // that sit a little off their lattice position
      uint64_t cell = 0;
      for (int j = 2; j >= 0; --j) {
        double mn = 1e300;
        for (int a = 0; a < 4; ++a) mn = std::min(mn, xl[3 * static_cast<size_t>(loc[4 * static_cast<size_t>(e) + a]) + j]);
        cell = (cell << 16) | static_cast<uint64_t>(static_cast<int>(std::floor((mn - lo[j]) / h + 0.25)) & 0xffff);
      }
      // inside a cell: by where the element's centroid sits in it (eighths of a cell), so that the tets of every cube
      // come in the same order
      // (first by the ORDER of the centroid's three offsets from the cell corner - which is largest, which smallest: six
      // patterns, one per tet of a Kuhn cube, and the same for a node moved by a fifth of a cell - then by the offsets in
      // quarters of a cell)
      double off[3];
      for (int j = 0; j < 3; ++j) {
        double c = 0.0;
        for (int a = 0; a < 4; ++a) c += 0.25 * xl[3 * static_cast<size_t>(loc[4 * static_cast<size_t>(e) + a]) + j];
        off[j] = (c - (lo[j] + h * static_cast<double>((cell >> (16 * j)) & 0xffff))) / h;
      }
      uint64_t best = static_cast<uint64_t>((off[0] > off[1]) + 2 * (off[1] > off[2]) + 4 * (off[0] > off[2]));
      for (int j = 2; j >= 0; --j)
        best = (best << 8) | static_cast<uint64_t>(static_cast<int>(std::floor(4.0 * off[j] + 0.5)) & 0xff);
      okey[e] = cell;
      ekey_min.push_back(best);
    }
    std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
      if (okey[x] != okey[y]) return okey[x] < okey[y];
      if (ekey_min[x] != ekey_min[y]) return ekey_min[x] < ekey_min[y];
      return x < y;
    });
  }
  // the matching first (mate / face of every element), the items afterwards
  std::vector<int32_t> mate(n_elem, -1);
  std::vector<int8_t> mface(n_elem, -1);
  auto join = [&](int32_t e, int k) {
    const int32_t f = nb[4 * static_cast<size_t>(e) + k];
    mate[e] = f;
    mface[e] = static_cast<int8_t>(k);
    mate[f] = e;
    mface[f] = static_cast<int8_t>(nbk[4 * static_cast<size_t>(e) + k]);
  };
  int32_t n_single = 0;
  for (int32_t e : order) {
    if (used[e]) continue;
    int best_k = -1, best_deg = 99;
    uint64_t best_key = ~0ull;
    for (int k = 0; k < 4; ++k) {
      const int32_t f = partner(e, k);
      if (f < 0) continue;
      used[e] = 1;  // (f's options without e; the list-order variant counts every free neighbour, as it always has)
      const int df = free_degree(f);
      used[e] = 0;
      const uint64_t kk = ekey[4 * static_cast<size_t>(e) + k];
      if (df < best_deg || (df == best_deg && kk < best_key)) {
        best_deg = df;
        best_key = kk;
        best_k = k;
      }
    }
    if (best_k >= 0) {
      join(e, best_k);
      used[e] = used[nb[4 * static_cast<size_t>(e) + best_k]] = 1;
    } else {
      used[e] = 1;
      ++n_single;
    }
  }
  // Unstructured meshes: the greedy pass leaves 5-10 % of the elements single (a Delaunay mesh of random points: 10 %),
  // and a single costs a lane as much as a pair.  Augmenting paths - single s, its neighbour f matched with g, g's other
  // neighbour t single: (s,f) (g,t) instead of (f,g), and longer chains of such exchanges - pick nearly all of them up
  // (99.6 % of the element copies paired); here an all-owned element may also pair with one that has halo nodes (the item
  // then waits for the halo records like its second element would have).  Only on request (`augment`: the caller asks when more than 1.5 % of a
  // block's elements stayed single): on lattices (99.8 % paired) the pairing stays exactly what the pattern classes of
  // the LDS packing were tuned on.
  if (augment && n_single > 0) {
    auto eligible = [&](int32_t e, int k) -> int32_t {  // partner(e, k) without the `used` and the class test
      const int32_t f = nb[4 * static_cast<size_t>(e) + k];
      if (f < 0) return -1;
      uint16_t A[4], B[4];
      apex_first(e, k, A);
      apex_first(f, nbk[4 * static_cast<size_t>(e) + k], B);
      return compatible(A, B) ? f : -1;
    };
    // augmenting paths by depth-first search (the bipartite scheme on a general graph: an element enters a path at most
    // once, so odd cycles are not contracted and a few paths are missed, but every path found is a valid one); paths of
    // up to kDepth exchanges, shortest first
    std::vector<int32_t> seen(n_elem, -1);
    int32_t stamp = 0;
    constexpr int kDepth = 10;
    // u is free (or has just lost its partner): find it a partner, displacing others along the way
    auto grow = [&](auto &&self, int32_t u, int depth) -> bool {
      for (int k = 0; k < 4; ++k) {  // a free neighbour first
        const int32_t v = eligible(u, k);
        if (v < 0 || seen[v] == stamp || mate[v] >= 0) continue;
        seen[v] = stamp;
        join(u, k);
        return true;
      }
      if (depth == 0) return false;
      for (int k = 0; k < 4; ++k) {
        const int32_t v = eligible(u, k);
        if (v < 0 || seen[v] == stamp) continue;
        const int32_t w = mate[v];
        if (w < 0 || seen[w] == stamp) continue;
        seen[v] = seen[w] = stamp;
        mate[w] = -1;  // w gives v up if it finds another partner
        if (self(self, w, depth - 1)) {
          join(u, k);
          return true;
        }
        mate[w] = v;   // (mface[w] is untouched: the pair stands as it was)
      }
      return false;
    };
    for (int depth = 1; depth <= kDepth; ++depth) {
      int32_t gained = 0;
      for (int32_t s0 : order) {
        if (mate[s0] >= 0) continue;
        ++stamp;
        seen[s0] = stamp;
        gained += grow(grow, s0, depth);
      }
      if (gained == 0 && depth >= 3) break;
    }
  }
  std::vector<char> out_done(n_elem, 0);
  for (int32_t e : order) {
    if (out_done[e]) continue;
    out_done[e] = 1;
    if (mate[e] >= 0) {
      out_done[mate[e]] = 1;
      emit_pair(e, mface[e]);
    } else {
      emit_single(e);
    }
  }
  return n_items;
}

// ------------------------------------------------------------------------------------------------
// LDS packing of the items of one block part
// ------------------------------------------------------------------------------------------------
// ds_read_b128 lane groups of a 32-lane half (MI355X_MICROARCH.md, LDS): {0-3,12-15,20-27}, {4-11,16-19,28-31}
constexpr int kGroupLanes[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};

struct PackStats {
  double read_mult = 0.0, atomic_mult = 0.0;  // sums of worst multiplicities
  int64_t read_cnt = 0, atomic_cnt = 0;
  int64_t by_construction = 0;  // items placed in clash-free halves by the pattern-class stage
  void add(const PackStats &o) {
    read_mult += o.read_mult;
    atomic_mult += o.atomic_mult;
    read_cnt += o.read_cnt;
    atomic_cnt += o.atomic_cnt;
    by_construction += o.by_construction;
  }
};

// Re-order (and re-label) the items of one block part for the LDS traffic of the element phase (LDS image
// and bank rules: saa_plan.h).  Measured on gfx950 (tools/lds_microbench.hip): ds_add_f64 costs 7 cycles per
// wave-instruction conflict-free, 22 for random nodes, 60 when 6 lanes hit one address (the natural order
// of the 6 tets around a cube diagonal); ds_read_b128 6 conflict-free, 11 random.
// Packing, one half-wave (32 item slots) at a time: scan the not yet placed items and take those for which
// one orientation-preserving relabelling (6 for a pair, 12 for a single tet) and one of the half's two
// ds_read_b128 lane groups leaves every vertex on a free bank: node mod 16 free in the group (reads), node
// mod 32 free in the half for owned vertices (atomics).  When the scan window runs dry the placement that
// raises the worst multiplicities least is taken.
// `pad` (0..1): the element phase is LDS-bound, an idle lane is nearly free while a bank clash costs every
// lane of the half a whole extra LDS pass; so a half is closed with NULL items (flag 2, skipped by the
// kernel) once no clash-free item is found, as long as the list grows by at most the fraction `pad`.
// Output: `out` (8 uint16 per item slot, nulls included); returns the number of item slots.
int32_t reorder_for_lds(const uint16_t *items, int32_t n_items, int32_t n_owned, double pad,
                        std::vector<uint16_t> &out, PackStats &st) {
  constexpr int kHalf = 32, kSlots = 5;
  constexpr int kWindow = 768;  // pool items examined per half before clashes / padding are accepted
  out.clear();
  if (n_items <= 0) return 0;
  std::vector<int32_t> pool(n_items);
  for (int32_t e = 0; e < n_items; ++e) pool[e] = e;
  int32_t pad_budget = static_cast<int32_t>(pad * n_items);
  std::vector<uint16_t> &scratch = out;
  auto n_syms = [](const uint16_t *it) { return it[5] ? 6 : 12; };
  auto relabel = [](const uint16_t *it, int q, uint16_t out[5]) {
    if (it[5]) {
      for (int a = 0; a < kSlots; ++a) out[a] = it[kPairSyms[q][a]];
    } else {
      for (int a = 0; a < 4; ++a) out[a] = it[kEvenPerms[q][a]];
      out[4] = out[1];  // dummy: p's record again (same address = broadcast), never accumulated
    }
  };
  int32_t done = 0, remaining = n_items;
  // ---- stage 1: conflict-free halves by construction ------------------------------------------------------------
  // Two items whose vertex slots differ by the same amount in every slot ("translates": what a structured region of a
  // mesh is made of) clash in all slots or in none.  So: bring every item into the relabelling with the smallest
  // difference pattern (slot k minus slot 0, mod 32), sort by pattern, and whenever a pattern class holds items with
  // all 32 residues of slot 0 mod 32, those 32 items form a half-wave without a single bank clash - residues 0..15 go
  // to the lanes of the first ds_read_b128 group, 16..31 to the second (distinct mod 16 inside a group for the reads,
  // distinct mod 32 over the half for the atomics, in every slot at once).  The greedy scan below cannot find these:
  // its last lanes of a group need one specific residue in each of five slots.  Unstructured regions have no classes
  // worth the name and fall through to stage 2 unchanged.
  {
    struct Cls {
      uint32_t key;   // 4 x 5-bit differences + pair flag
      uint8_t res, q;
      int32_t item;
    };
    std::vector<Cls> cls(n_items);
    for (int32_t e = 0; e < n_items; ++e) {
      const uint16_t *it = items + 8 * static_cast<size_t>(e);
      uint32_t best = 0xffffffffu;
      uint8_t best_q = 0, best_res = 0;
      for (int q = 0; q < n_syms(it); ++q) {
        uint16_t v[5];
        relabel(it, q, v);
        uint32_t key = it[5] ? 1u : 0u;
        for (int a = 1; a < 5; ++a) key = (key << 5) | ((v[a] - v[0]) & 31u);
        if (key < best) {
          best = key;
          best_q = static_cast<uint8_t>(q);
          best_res = static_cast<uint8_t>(v[0] & 31u);
        }
      }
      cls[e] = {best, best_res, best_q, e};
    }
    std::sort(cls.begin(), cls.end(), [](const Cls &x, const Cls &y) {
      if (x.key != y.key) return x.key < y.key;
      if (x.res != y.res) return x.res < y.res;
      return x.item < y.item;
    });
    std::vector<char> taken(n_items, 0);
    for (size_t lo = 0; lo < cls.size();) {
      size_t hi = lo;
      while (hi < cls.size() && cls[hi].key == cls[lo].key) ++hi;
      // per residue: the run [first[r], first[r+1]) inside [lo, hi)
      size_t first[33];
      {
        size_t p = lo;
        for (int r = 0; r < 32; ++r) {
          while (p < hi && cls[p].res < r) ++p;
          first[r] = p;
        }
        first[32] = hi;
      }
      size_t halves = hi - lo;
      for (int r = 0; r < 32; ++r) halves = std::min(halves, first[r + 1] - first[r]);
      for (size_t h = 0; h < halves; ++h) {
        scratch.resize(8 * static_cast<size_t>(done + kHalf), 0);
        for (int r = 0; r < 32; ++r) {
          const Cls &c = cls[first[r] + h];
          const uint16_t *it = items + 8 * static_cast<size_t>(c.item);
          uint16_t v[5];
          relabel(it, c.q, v);
          const int lane = kGroupLanes[r >> 4][r & 15];
          uint16_t *dst = &scratch[8 * static_cast<size_t>(done + lane)];
          for (int a = 0; a < kSlots; ++a) dst[a] = v[a];
          dst[5] = it[5];
          taken[c.item] = 1;
        }
        const int real = (cls[lo].key >> 20) ? 5 : 4;  // pair flag sits above the four 5-bit fields
        st.read_mult += 2.0 * real;
        st.read_cnt += 2 * real;
        st.atomic_mult += real;
        st.atomic_cnt += real;
        done += kHalf;
        remaining -= kHalf;
        st.by_construction += kHalf;
      }
      lo = hi;
    }
    // second tier, on what the classes have left: 16 items of one class with all residues mod 16 make a clash-free
    // ds_read_b128 group; two such groups (of whatever classes) share a half-wave, where only their atomics may still
    // meet (mod 32 across the half) - counted honestly below
    struct Grp {
      int32_t item[16];
      uint8_t q[16];
    };
    std::vector<Grp> groups;
    for (size_t lo = 0; lo < cls.size();) {
      size_t hi = lo;
      while (hi < cls.size() && cls[hi].key == cls[lo].key) ++hi;
      std::vector<int32_t> by_res[16];
      for (size_t i = lo; i < hi; ++i)
        if (!taken[cls[i].item]) by_res[cls[i].res & 15].push_back(static_cast<int32_t>(i));
      size_t n_grp = hi - lo;
      for (auto &b : by_res) n_grp = std::min(n_grp, b.size());
      for (size_t h = 0; h < n_grp; ++h) {
        Grp gp;
        for (int r = 0; r < 16; ++r) {
          const Cls &c = cls[by_res[r][h]];
          gp.item[r] = c.item;
          gp.q[r] = c.q;
        }
        groups.push_back(gp);
      }
      lo = hi;
    }
    for (size_t g2 = 0; g2 + 1 < groups.size(); g2 += 2) {
      scratch.resize(8 * static_cast<size_t>(done + kHalf), 0);
      uint8_t cnt_at[kSlots][32] = {};
      int max_at[kSlots] = {}, real_max = 4;
      for (int g = 0; g < 2; ++g)
        for (int r = 0; r < 16; ++r) {
          const int32_t e = groups[g2 + g].item[r];
          const uint16_t *it = items + 8 * static_cast<size_t>(e);
          uint16_t v[5];
          relabel(it, groups[g2 + g].q[r], v);
          uint16_t *dst = &scratch[8 * static_cast<size_t>(done + kGroupLanes[g][r])];
          const int real = it[5] ? 5 : 4;
          real_max = std::max(real_max, real);
          for (int a = 0; a < kSlots; ++a) {
            dst[a] = v[a];
            if (a < real && v[a] < n_owned) max_at[a] = std::max<int>(max_at[a], ++cnt_at[a][v[a] & 31]);
          }
          dst[5] = it[5];
          taken[e] = 1;
        }
      st.read_mult += 2.0 * real_max;
      st.read_cnt += 2 * real_max;
      for (int a = 0; a < real_max; ++a)
        if (max_at[a] > 0) {
          st.atomic_mult += max_at[a];
          ++st.atomic_cnt;
        }
      done += kHalf;
      remaining -= kHalf;
      st.by_construction += kHalf;
    }
    pool.erase(std::remove_if(pool.begin(), pool.end(), [&](int32_t e) { return taken[e] != 0; }), pool.end());
  }
  // ---- stage 2: greedy packing of what is left ----------------------------------------------------------------
  while (remaining > 0) {
    const int32_t cap = kHalf;
    scratch.resize(8 * static_cast<size_t>(done + kHalf), 0);
    for (int32_t l = 0; l < kHalf; ++l) scratch[8 * static_cast<size_t>(done + l) + 5] = 2;  // null until filled
    int32_t free_lane[2][16], n_free[2] = {0, 0};
    for (int g = 0; g < 2; ++g)
      for (int j = 0; j < 16; ++j)
        if (kGroupLanes[g][j] < cap) free_lane[g][n_free[g]++] = kGroupLanes[g][j];
    int used[2] = {0, 0};
    uint8_t cnt_rd[2][kSlots][16] = {}, cnt_at[kSlots][32] = {};
    int max_rd[2][kSlots] = {}, max_at[kSlots] = {};
    uint32_t taken_rd[2][kSlots] = {}, taken_at[kSlots] = {};
    int32_t placed = 0;
    auto put = [&](int32_t pool_pos, int q, int g) {
      const uint16_t *it = items + 8 * static_cast<size_t>(pool[pool_pos]);
      uint16_t v[5];
      relabel(it, q, v);
      const int32_t lane = free_lane[g][used[g]++];
      uint16_t *dst = &scratch[8 * static_cast<size_t>(done + lane)];
      const int real = it[5] ? 5 : 4;
      for (int a = 0; a < kSlots; ++a) {
        dst[a] = v[a];
        if (a >= real) continue;
        taken_rd[g][a] |= 1u << (v[a] & 15);
        max_rd[g][a] = std::max<int>(max_rd[g][a], ++cnt_rd[g][a][v[a] & 15]);
        if (v[a] < n_owned) {
          taken_at[a] |= 1u << (v[a] & 31);
          max_at[a] = std::max<int>(max_at[a], ++cnt_at[a][v[a] & 31]);
        }
      }
      dst[5] = it[5];
      ++placed;
      --remaining;
      pool[pool_pos] = -1;
    };
    const int32_t lim = std::min<int32_t>(static_cast<int32_t>(pool.size()), kWindow);
    // pass 1: clash-free placements
    for (int32_t p = 0; p < lim && placed < cap; ++p) {
      const uint16_t *it = items + 8 * static_cast<size_t>(pool[p]);
      const int real = it[5] ? 5 : 4;
      bool ok = false;
      for (int q = 0; q < n_syms(it) && !ok; ++q) {
        uint16_t v[5];
        relabel(it, q, v);
        uint32_t at = 0;
        for (int a = 0; a < real; ++a)
          if (v[a] < n_owned) at |= taken_at[a] >> (v[a] & 31);
        if (at & 1u) continue;
        for (int g = 0; g < 2 && !ok; ++g) {
          if (used[g] >= n_free[g]) continue;
          uint32_t rd = 0;
          for (int a = 0; a < real; ++a) rd |= taken_rd[g][a] >> (v[a] & 15);
          if (rd & 1u) continue;
          put(p, q, g);
          ok = true;
        }
      }
    }
    // pass 2: fill the rest where it raises the worst multiplicities least (an LDS instruction costs its
    // WORST bank multiplicity; reads 3 x ds_read_b128 per vertex ~4 cycles a level, atomics 3 x ds_add_f64 ~7)
    while (placed < cap && remaining > 0) {
      if (pad_budget > 0) {  // leave the remaining lanes of this half idle instead of clashing
        pad_budget -= cap - placed;
        break;
      }
      int32_t best_p = -1, best_q = 0, best_g = 0, best_k = 1 << 30;
      for (int32_t p = 0; p < lim && best_k > 0; ++p) {
        if (pool[p] < 0) continue;
        const uint16_t *it = items + 8 * static_cast<size_t>(pool[p]);
        const int real = it[5] ? 5 : 4;
        for (int q = 0; q < n_syms(it) && best_k > 0; ++q) {
          uint16_t v[5];
          relabel(it, q, v);
          int k_at = 0;
          for (int a = 0; a < real; ++a)
            if (v[a] < n_owned && cnt_at[a][v[a] & 31] + 1 > max_at[a]) k_at += 21;
          for (int g = 0; g < 2; ++g) {
            if (used[g] >= n_free[g]) continue;
            int k = k_at;
            for (int a = 0; a < real; ++a)
              if (cnt_rd[g][a][v[a] & 15] + 1 > max_rd[g][a]) k += 12;
            if (k < best_k) {
              best_k = k;
              best_p = p;
              best_q = q;
              best_g = g;
            }
          }
        }
      }
      put(best_p, best_q, best_g);
    }
    pool.erase(std::remove(pool.begin(), pool.begin() + lim, -1), pool.begin() + lim);
    for (int a = 0; a < kSlots; ++a) {
      for (int g = 0; g < 2; ++g)
        if (max_rd[g][a] > 0) {
          st.read_mult += max_rd[g][a];
          ++st.read_cnt;
        }
      if (max_at[a] > 0) {
        st.atomic_mult += max_at[a];
        ++st.atomic_cnt;
      }
    }
    done += cap;
  }
  // trim trailing null slots of the last half
  while (done > 0 && scratch[8 * static_cast<size_t>(done - 1) + 5] == 2) --done;
  scratch.resize(8 * static_cast<size_t>(done));
  return done;
}


// pi[l] = rank of owned node l of a block when its nodes are sorted lexicographically with the axes in the q-th order
void block_axis_order(const double *xyz, const int32_t *new_to_old, int32_t n_owned, int q, std::vector<uint16_t> &pi) {
  static const int kPerm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  const int k0 = kPerm[q][0], k1 = kPerm[q][1], k2 = kPerm[q][2];
  std::vector<int32_t> ord(n_owned);
  std::iota(ord.begin(), ord.end(), 0);
  std::sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) {
    const double *pa = xyz + 3 * static_cast<int64_t>(new_to_old[a]), *pb = xyz + 3 * static_cast<int64_t>(new_to_old[b]);
    if (pa[k0] != pb[k0]) return pa[k0] < pb[k0];
    if (pa[k1] != pb[k1]) return pa[k1] < pb[k1];
    if (pa[k2] != pb[k2]) return pa[k2] < pb[k2];
    return a < b;
  });
  pi.resize(n_owned);
  for (int32_t r = 0; r < n_owned; ++r) pi[ord[r]] = static_cast<uint16_t>(r);
}
// Pseudo-lattice numbering of a block's nodes for meshes whose nodes do not sit on a lattice: the coordinates are
// quantised with the mesh size h (cube root of six mean element volumes: the edge of the cube a Kuhn tet came from, the
// typical node spacing of any other mesh), and a node with quantised position (i, j, k) gets a local index whose residue
// mod 32 is that of its lexicographic lattice index i + L_i * (j + L_j * k) (L: the block's extent in cells along the
// axis; which axis runs fastest is the variant) - as far as the 32 residue classes have room (each holds n/32 indices);
// what does not fit takes the indices left over.  Two items that are translates of each other by whole cells then differ
// by one constant in the residues of all their vertices, which is all the pattern classes of reorder_for_lds need: they
// never look at the indices themselves.  Halo nodes are placed the same way (their local index is n_owned + position in
// the halo list), so that boundary items fall into classes too.  On a jittered lattice this restores the classes of the
// undisturbed one; on a genuinely unstructured mesh it yields them wherever the mesh is locally regular.
struct LatticeColours {
  double lo[3], h;  // lo: the lattice plane the lowest owned node rounds to
  int ext[3], axis[3];
  // residue class (mod 32) of a position
  int operator()(const double *x) const {
    int q[3];
    for (int k = 0; k < 3; ++k) q[k] = static_cast<int>(std::floor((x[k] - lo[k]) / h + 0.5));
    return (q[axis[0]] + ext[axis[0]] * (q[axis[1]] + ext[axis[1]] * q[axis[2]])) & 31;
  }
};
LatticeColours lattice_colours(const double *xyz, const int32_t *new_to_old, int32_t n_owned, double h, int variant) {
  static const int kPerm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  LatticeColours lc;
  lc.h = h;
  // Where the lattice planes lie along each axis: the PHASE of the node coordinates modulo h, as the circular mean
  // of x/h - the mean over all nodes of the block, so that nodes moved off their lattice position by a fifth of a cell
  // in either direction leave it where it was; taking the lowest node as the origin would shift every rounding
  // boundary by that node's own displacement and put a tenth of the nodes into the wrong cell (each of them sits in
  // seventeen items, which then fall out of their classes).  A mesh without any lattice has no phase; any value does.
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, sn[3] = {0, 0, 0}, cs[3] = {0, 0, 0};
  const double two_pi = 6.283185307179586;
  for (int32_t l = 0; l < n_owned; ++l)
    for (int k = 0; k < 3; ++k) {
      const double v = xyz[3 * static_cast<int64_t>(new_to_old[l]) + k];
      lo[k] = std::min(lo[k], v);
      hi[k] = std::max(hi[k], v);
      sn[k] += std::sin(two_pi * v / h);
      cs[k] += std::cos(two_pi * v / h);
    }
  for (int k = 0; k < 3; ++k) {
    const double phase = h * std::atan2(sn[k], cs[k]) / two_pi;                // planes at phase + m h
    lc.lo[k] = phase + h * std::floor((lo[k] - phase) / h + 0.5);             // the plane the lowest node rounds to
    lc.ext[k] = static_cast<int>(std::floor((hi[k] - lc.lo[k]) / h + 0.5)) + 1;
    lc.axis[k] = kPerm[variant % 6][k];
  }
  return lc;
}
// indices first..first+n-1 handed out so that index % 32 == colour wherever the class has room; pi[l] = index - first
void assign_by_colour(const std::vector<int> &colour, int32_t first, std::vector<uint16_t> &pi) {
  const int32_t n = static_cast<int32_t>(colour.size());
  std::vector<int32_t> next(32);
  for (int r = 0; r < 32; ++r) next[r] = first + ((r - first) % 32 + 32) % 32;  // smallest index >= first with residue r
  pi.assign(n, 0xffff);
  std::vector<int32_t> spill;
  for (int32_t l = 0; l < n; ++l) {
    int32_t &i = next[colour[l]];
    if (i < first + n) {
      pi[l] = static_cast<uint16_t>(i - first);
      i += 32;
    } else {
      spill.push_back(l);
    }
  }
  size_t j = 0;
  for (int r = 0; r < 32 && j < spill.size(); ++r)
    for (int32_t i = next[r]; i < first + n && j < spill.size(); i += 32) pi[spill[j++]] = static_cast<uint16_t>(i - first);
}
void block_lattice_order(const double *xyz, const int32_t *new_to_old, int32_t n_owned, double h, int variant,
                         std::vector<uint16_t> &pi) {
  const LatticeColours lc = lattice_colours(xyz, new_to_old, n_owned, h, variant);
  std::vector<int> colour(n_owned);
  for (int32_t l = 0; l < n_owned; ++l) colour[l] = lc(xyz + 3 * static_cast<int64_t>(new_to_old[l]));
  assign_by_colour(colour, 0, pi);
}
// halo vertex slots of `n` items renamed through pih (old -> new position in the halo list)
void relabel_halo(uint16_t *items, int32_t n, int32_t n_owned, const std::vector<uint16_t> &pih) {
  for (int32_t i = 0; i < n; ++i) {
    uint16_t *it = items + 8 * static_cast<size_t>(i);
    for (int a = 0; a < 5; ++a)
      if (it[a] >= n_owned) it[a] = static_cast<uint16_t>(n_owned + pih[it[a] - n_owned]);
  }
}
// owned vertex slots of `n` items renamed through pi (halo slots keep their numbers)
void relabel_owned(uint16_t *items, int32_t n, int32_t n_owned, const std::vector<uint16_t> &pi) {
  for (int32_t i = 0; i < n; ++i) {
    uint16_t *it = items + 8 * static_cast<size_t>(i);
    for (int a = 0; a < (it[5] ? 5 : 4); ++a)
      if (it[a] < n_owned) it[a] = pi[it[a]];
  }
}

// ------------------------------------------------------------------------------------------------
// Joint numbering and packing for blocks without lattice structure
// ------------------------------------------------------------------------------------------------
// Worst bank multiplicities of a packed list as the kernels will meet them (indices as they stand): per half-wave and
// vertex slot the reads of either 16-lane group (index mod 16) and the atomics of the half (owned indices mod 32).
void measure_pack(const uint16_t *list, int32_t n_slots, int32_t n_owned, PackStats &st) {
  for (int32_t base = 0; base < n_slots; base += 32) {
    uint8_t cnt_rd[2][5][16] = {}, cnt_at[5][32] = {};
    int max_rd[2][5] = {}, max_at[5] = {};
    for (int g = 0; g < 2; ++g)
      for (int j = 0; j < 16; ++j) {
        const int32_t sl = base + kGroupLanes[g][j];
        if (sl >= n_slots) continue;
        const uint16_t *it = list + 8 * static_cast<size_t>(sl);
        if (it[5] == 2) continue;
        const int real = it[5] ? 5 : 4;
        for (int a = 0; a < real; ++a) {
          max_rd[g][a] = std::max<int>(max_rd[g][a], ++cnt_rd[g][a][it[a] & 15]);
          if (it[a] < n_owned) max_at[a] = std::max<int>(max_at[a], ++cnt_at[a][it[a] & 31]);
        }
      }
    for (int a = 0; a < 5; ++a) {
      for (int g = 0; g < 2; ++g)
        if (max_rd[g][a] > 0) {
          st.read_mult += max_rd[g][a];
          ++st.read_cnt;
        }
      if (max_at[a] > 0) {
        st.atomic_mult += max_at[a];
        ++st.atomic_cnt;
      }
    }
  }
}

// A mesh without lattice structure has no pattern classes, and with ANY numbering fixed beforehand the greedy packing of
// reorder_for_lds runs into the same wall: the last lanes of a 16-lane group need one specific bank residue in each of five
// vertex slots, and among a few thousand items with effectively random residues there is none (14 lanes fill clash-free,
// the 15th in two cases of five, the 16th never: read conflict factor 1.7 on a Delaunay mesh whatever the numbering).  What
// can still be chosen when those lanes are filled is the residue of a node that has not appeared yet.  So here the
// numbering is decided WHILE the groups are formed: a node's residue (local index mod 32) is fixed the first time an item
// that names it is placed, as whatever the group it joins has free; items all of whose nodes are fixed go first, items
// with free nodes fill the lanes that nothing else fits.  Lists `n_list` are packed one after the other with one shared
// numbering (the first round, then the rest).  Out: per owned node and per halo position the residue chosen (assign_by_colour
// turns them into indices), and the packed lists in the OLD indices (the caller renames them).
struct JointState {
  int32_t n_owned = 0, n_halo = 0;
  std::vector<int8_t> res;        // per local node: residue 0..31, -1 = free
  int32_t room_owned[32], room_halo[32];
};

int32_t joint_pack_list(const uint16_t *items, int32_t n_items, JointState &js, std::vector<uint16_t> &out, double pad = 0.0) {
  constexpr int kHalf = 32, kSlots = 5, kWindow = 1024;
  out.clear();
  if (n_items <= 0) return 0;
  const int32_t n_owned = js.n_owned;
  std::vector<int32_t> pool(n_items);
  std::iota(pool.begin(), pool.end(), 0);
  auto n_syms = [](const uint16_t *it) { return it[5] ? 6 : 12; };
  auto relabel = [](const uint16_t *it, int q, uint16_t v[5]) {
    if (it[5]) {
      for (int a = 0; a < kSlots; ++a) v[a] = it[kPairSyms[q][a]];
    } else {
      for (int a = 0; a < 4; ++a) v[a] = it[kEvenPerms[q][a]];
      v[4] = v[1];
    }
  };
  int32_t done = 0, remaining = n_items;
  int32_t pad_budget = static_cast<int32_t>(pad * n_items);  // idle lanes allowed instead of clashes (reorder_for_lds: `pad`)
  std::vector<int32_t> bucket[6];
  while (remaining > 0) {
    out.resize(8 * static_cast<size_t>(done + kHalf), 0);
    for (int32_t l = 0; l < kHalf; ++l) out[8 * static_cast<size_t>(done + l) + 5] = 2;
    int used[2] = {0, 0};
    uint32_t taken_rd[2][kSlots] = {}, taken_at[kSlots] = {};
    uint8_t cnt_rd[2][kSlots][16] = {}, cnt_at[kSlots][32] = {};
    int max_rd[2][kSlots] = {}, max_at[kSlots] = {};
    int32_t placed = 0;
    // a residue for the free node `v` in slot a of group g: room left in its class, bank free for the group's reads and,
    // if owned, for the half's atomics; the class with most room left (keeps the classes level).  -1: none.
    auto pick = [&](uint16_t v, int g, int a) -> int {
      const bool owned = v < n_owned;
      const int32_t *room = owned ? js.room_owned : js.room_halo;
      // residues whose bank is free for the group's reads (both halves of the 32: c and c + 16 share a read bank) and, for
      // an owned node, for the half's atomics
      uint32_t ok = ~(taken_rd[g][a] | (taken_rd[g][a] << 16));
      if (owned) ok &= ~taken_at[a];
      int best = -1, best_room = 0;
      while (ok) {
        const int c = __builtin_ctz(ok);
        ok &= ok - 1;
        if (room[c] > best_room) {
          best = c;
          best_room = room[c];
        }
      }
      return best;
    };
    auto put = [&](int32_t pool_pos, int q, int g, const int *chosen) {  // chosen[a]: residue for a free node in slot a, or -1
      const uint16_t *it = items + 8 * static_cast<size_t>(pool[pool_pos]);
      uint16_t v[5];
      relabel(it, q, v);
      const int32_t lane = kGroupLanes[g][used[g]++];
      uint16_t *dst = &out[8 * static_cast<size_t>(done + lane)];
      const int real = it[5] ? 5 : 4;
      for (int a = 0; a < kSlots; ++a) {
        dst[a] = v[a];
        if (a >= real) continue;
        if (js.res[v[a]] < 0) {
          int c = chosen ? chosen[a] : -1;
          if (c < 0) {  // forced placement: the class with most room, clash or not
            const int32_t *room = v[a] < n_owned ? js.room_owned : js.room_halo;
            c = 0;
            for (int k = 1; k < 32; ++k)
              if (room[k] > room[c]) c = k;
          }
          js.res[v[a]] = static_cast<int8_t>(c);
          --(v[a] < n_owned ? js.room_owned : js.room_halo)[c];
        }
        const int c = js.res[v[a]];
        taken_rd[g][a] |= 1u << (c & 15);
        max_rd[g][a] = std::max<int>(max_rd[g][a], ++cnt_rd[g][a][c & 15]);
        if (v[a] < n_owned) {
          taken_at[a] |= 1u << c;
          max_at[a] = std::max<int>(max_at[a], ++cnt_at[a][c]);
        }
      }
      dst[5] = it[5];
      ++placed;
      --remaining;
      pool[pool_pos] = -1;
    };
    const int32_t lim = std::min<int32_t>(static_cast<int32_t>(pool.size()), kWindow);
    // clash-free placements: first the items whose nodes are all fixed (sweep 0), then those with one free node, two, ...
    // (the window sorted into those classes once per half, as the numbering stands at its start)
    for (auto &b : bucket) b.clear();
    for (int32_t p = 0; p < lim; ++p) {
      const uint16_t *it = items + 8 * static_cast<size_t>(pool[p]);
      int n_free = 0;
      for (int a = 0; a < (it[5] ? 5 : 4); ++a) n_free += js.res[it[a]] < 0;
      bucket[n_free].push_back(p);
    }
    for (int sweep = 0; sweep <= 5 && placed < kHalf; ++sweep)
      for (size_t bi = 0; bi < bucket[sweep].size() && placed < kHalf; ++bi) {
        const int32_t p = bucket[sweep][bi];
        if (pool[p] < 0) continue;
        const uint16_t *it = items + 8 * static_cast<size_t>(pool[p]);
        const int real = it[5] ? 5 : 4;
        bool ok = false;
        for (int q = 0; q < n_syms(it) && !ok; ++q) {
          uint16_t v[5];
          relabel(it, q, v);
          for (int g = 0; g < 2 && !ok; ++g) {
            if (used[g] >= 16) continue;
            int chosen[5] = {-1, -1, -1, -1, -1};
            bool fits = true;
            for (int a = 0; a < real && fits; ++a) {
              const int c = js.res[v[a]];
              if (c >= 0) {
                if ((taken_rd[g][a] >> (c & 15)) & 1u) fits = false;
                if (v[a] < n_owned && ((taken_at[a] >> c) & 1u)) fits = false;
              } else {
                chosen[a] = pick(v[a], g, a);
                if (chosen[a] < 0) fits = false;
              }
            }
            if (!fits) continue;
            put(p, q, g, chosen);
            ok = true;
          }
        }
      }
    // the rest where it raises the worst multiplicities least (free nodes cost nothing: they take a free bank if any)
    while (placed < kHalf && remaining > 0) {
      if (pad_budget > 0 && placed > 0) {  // leave the remaining lanes of this half idle instead of clashing
        pad_budget -= kHalf - placed;
        break;
      }
      int32_t best_p = -1, best_q = 0, best_g = 0, best_k = 1 << 30;
      for (int32_t p = 0; p < lim && best_k > 0; ++p) {
        if (pool[p] < 0) continue;
        const uint16_t *it = items + 8 * static_cast<size_t>(pool[p]);
        const int real = it[5] ? 5 : 4;
        for (int q = 0; q < n_syms(it) && best_k > 0; ++q) {
          uint16_t v[5];
          relabel(it, q, v);
          for (int g = 0; g < 2; ++g) {
            if (used[g] >= 16) continue;
            int k = 0;
            for (int a = 0; a < real; ++a) {
              const int c = js.res[v[a]];
              if (c < 0) continue;
              if (cnt_rd[g][a][c & 15] + 1 > max_rd[g][a]) k += 12;
              if (v[a] < n_owned && cnt_at[a][c] + 1 > max_at[a]) k += 21;
            }
            if (k < best_k) {
              best_k = k;
              best_p = p;
              best_q = q;
              best_g = g;
            }
          }
        }
      }
      // free nodes of the forced item: a free bank where there is one
      const uint16_t *it = items + 8 * static_cast<size_t>(pool[best_p]);
      uint16_t v[5];
      relabel(it, best_q, v);
      int chosen[5] = {-1, -1, -1, -1, -1};
      for (int a = 0; a < (it[5] ? 5 : 4); ++a)
        if (js.res[v[a]] < 0) chosen[a] = pick(v[a], best_g, a);
      put(best_p, best_q, best_g, chosen);
    }
    pool.erase(std::remove(pool.begin(), pool.begin() + lim, -1), pool.begin() + lim);
    done += kHalf;
  }
  while (done > 0 && out[8 * static_cast<size_t>(done - 1) + 5] == 2) --done;
  out.resize(8 * static_cast<size_t>(done));
  return done;
}

// ------------------------------------------------------------------------------------------------
// repair_margin: item slots the chunk repair leaves free per block for single elements and idle slots of the packing (what
// it balances are element copies; items = copies / 2 + half the single elements + idle slots).  overflow (out): by how many
// items the fullest block exceeds the chunk count the repair aimed at (0: none, or no repair) - build_plan tries once more
// with a larger margin then.
bool build_once(int32_t n_nodes, int32_t n_elems, const double *xyz, const int32_t *tets,
                int32_t block_nodes, Plan &plan, std::string &err, bool &too_big, const int32_t *extra_work,
                int32_t repair_margin, int32_t *overflow) {
  too_big = false;
  if (overflow) *overflow = 0;
  int32_t chunk_capacity = 0;
  bool second_list_pays = false;  // decided with the chunk repair below
  plan = Plan();
  plan.n_nodes = n_nodes;
  plan.n_elems = n_elems;
  const int32_t nb = choose_block_count(n_nodes, block_nodes);

  plan.new_to_old.resize(n_nodes);
  std::iota(plan.new_to_old.begin(), plan.new_to_old.end(), 0);
  std::vector<int32_t> block_start;
  block_start.reserve(nb + 1);
  Rcb rcb{xyz, plan.new_to_old, block_start};
  // coordinates in units of the mesh size (cube root of six mean element volumes), rounded: AxisOrder
  std::vector<int32_t> cell;
  {
    const char *snap_env = diag_env("SAA_PLAN_SNAP_CUTS");
    if (!(snap_env && snap_env[0] == '0') && n_elems > 0 && nb > 1) {
      double vol = 0.0, lo[3] = {1e300, 1e300, 1e300};
      for (int32_t e = 0; e < n_elems; ++e) {
        const double *x0 = xyz + 3 * static_cast<int64_t>(tets[4 * static_cast<int64_t>(e)]),
                     *x1 = xyz + 3 * static_cast<int64_t>(tets[4 * static_cast<int64_t>(e) + 1]),
                     *x2 = xyz + 3 * static_cast<int64_t>(tets[4 * static_cast<int64_t>(e) + 2]),
                     *x3 = xyz + 3 * static_cast<int64_t>(tets[4 * static_cast<int64_t>(e) + 3]);
        const double a[3] = {x1[0] - x0[0], x1[1] - x0[1], x1[2] - x0[2]}, b2[3] = {x2[0] - x0[0], x2[1] - x0[1], x2[2] - x0[2]},
                     c[3] = {x3[0] - x0[0], x3[1] - x0[1], x3[2] - x0[2]};
        vol += std::fabs(a[0] * (b2[1] * c[2] - b2[2] * c[1]) - a[1] * (b2[0] * c[2] - b2[2] * c[0]) + a[2] * (b2[0] * c[1] - b2[1] * c[0]));
      }
      const double h = std::cbrt(vol / n_elems);  // |detJ| = 6 V: the edge of the cube six such tets fill
      if (h > 0.0 && std::isfinite(h)) {
        for (int32_t i = 0; i < n_nodes; ++i)
          for (int a = 0; a < 3; ++a) lo[a] = std::min(lo[a], xyz[3 * static_cast<int64_t>(i) + a]);
        cell.resize(3 * static_cast<size_t>(n_nodes));
        double dev = 0.0;  // largest distance of a node from its cell's lattice point, in mesh sizes
        for (int32_t i = 0; i < n_nodes; ++i)
          for (int a = 0; a < 3; ++a) {
            const double u = (xyz[3 * static_cast<int64_t>(i) + a] - lo[a]) / h, r = std::floor(u + 0.5);
            cell[3 * static_cast<size_t>(i) + a] = static_cast<int32_t>(r);
            dev = std::max(dev, std::fabs(u - r));
          }
        // an exact lattice keeps the plain order (ties by node id: whatever strips the caller's numbering gives - measured
        // 0.8 % better than strips by cell on the slabs of the 8-GPU partition, whose local numbering is first-touch)
        if (dev > 1e-6) rcb.cell = cell.data();
      }
    }
  }
  const AxisOrder axis_order{xyz, rcb.cell};
  // A block's work is its element copies, not its nodes: blocks in the bulk (more elements per node, every face
  // shared with a neighbour) must get fewer nodes than blocks at the surface, or the slowest block - which paces
  // all the others, directly in the resident kernel - carries ~10 % more items than the mean.  Weighted bisection:
  // a node weighs the elements around it, then the weights of every block are rescaled by its measured
  // copies / mean and the bisection is repeated (a few rounds; it only moves the cuts).
  std::vector<int64_t> weight;
  const char *wenv = diag_env("SAA_PLAN_WEIGHTED");
  if (nb > 1 && !(wenv && wenv[0] == '0')) {
    weight.assign(n_nodes, 1024);
    for (int64_t i = 0; i < 4 * static_cast<int64_t>(n_elems); ++i) weight[tets[i]] += 1024;
    if (extra_work)
      for (int32_t i = 0; i < n_nodes; ++i) weight[i] += 4096ll * extra_work[i];  // a copy weighs ~4 node-valences
    rcb.weight = weight.data();
    std::vector<int32_t> owner(n_nodes), copies;
    // (four rounds: more do not help - on a lattice the cuts move by whole layers of nodes, and the copies per block stay
    // within 2 % of their mean whatever the weights; measured with 10 and 20 rounds)
    for (int round = 0; round < 4; ++round) {
      std::iota(plan.new_to_old.begin(), plan.new_to_old.end(), 0);
      block_start.clear();
      rcb.split(0, n_nodes, nb);
      const int32_t nblk = static_cast<int32_t>(block_start.size());
      for (int32_t b = 0; b < nblk; ++b) {
        const int32_t end = b + 1 < nblk ? block_start[b + 1] : n_nodes;
        for (int32_t i = block_start[b]; i < end; ++i) owner[plan.new_to_old[i]] = b;
      }
      copies.assign(nblk, 0);
      for (int32_t e = 0; e < n_elems; ++e) {
        int32_t bs[4];
        int cnt = 0;
        for (int a = 0; a < 4; ++a) {
          const int32_t b = owner[tets[4 * static_cast<int64_t>(e) + a]];
          bool seen = false;
          for (int j = 0; j < cnt; ++j) seen |= (bs[j] == b);
          if (!seen) bs[cnt++] = b;
        }
        for (int j = 0; j < cnt; ++j) ++copies[bs[j]];
      }
      if (extra_work)
        for (int32_t i = 0; i < n_nodes; ++i) copies[owner[i]] += extra_work[i];
      double mean = 0;
      int32_t mx = 0;
      for (int32_t c : copies) {
        mean += c;
        mx = std::max(mx, c);
      }
      mean /= nblk;
      if (round == 3 || mx <= 1.015 * mean) break;
      for (int32_t i = 0; i < n_nodes; ++i)
        weight[i] = std::max<int64_t>(1, static_cast<int64_t>(weight[i] * (copies[owner[i]] / mean)));
    }
    // ---- chunk repair (one block per CU, 1024-thread workgroups: see first_round_cap below) ---------------------------
    // A block's items run as 1024 in the first round + 64-item chunks in the second phase, dealt to 16 waves on four SIMDs:
    // a block one chunk over the others gives one SIMD 11 chunks instead of 10, and the blocks wait for each other.  The
    // re-weighted bisection leaves the copies within +-2 % of their mean, a handful of blocks a few items over the chunk
    // count the mean fits.  Those hand nodes to the sibling leaf of their last bisection - the cut between the two moves,
    // a strip of the plane it runs through changes sides - as long as the sibling stays inside the budget itself.
    const char *rep_env = diag_env("SAA_PLAN_CHUNK_REPAIR");
    const int32_t nblk = static_cast<int32_t>(block_start.size());
    if (!(rep_env && rep_env[0] == '0') && nblk <= 256 && nblk >= 2 && (nblk & (nblk - 1)) == 0) {
      auto leaf_end = [&](int32_t b) { return b + 1 < nblk ? block_start[b + 1] : n_nodes; };
      for (int32_t b = 0; b < nblk; ++b)
        for (int32_t i = block_start[b]; i < leaf_end(b); ++i) owner[plan.new_to_old[i]] = b;
      auto count_copies = [&](std::vector<int32_t> &out) {
        out.assign(nblk, 0);
        for (int32_t e = 0; e < n_elems; ++e) {
          int32_t bs[4];
          int cnt = 0;
          for (int a = 0; a < 4; ++a) {
            const int32_t bb = owner[tets[4 * static_cast<int64_t>(e) + a]];
            bool seen = false;
            for (int j = 0; j < cnt; ++j) seen |= (bs[j] == bb);
            if (!seen) bs[cnt++] = bb;
          }
          for (int j = 0; j < cnt; ++j) ++out[bs[j]];
        }
        if (extra_work)
          for (int32_t i = 0; i < n_nodes; ++i) out[owner[i]] += extra_work[i];
      };
      count_copies(copies);
      double mean = 0;
      int32_t mx = 0;
      for (int32_t c : copies) {
        mean += c;
        mx = std::max(mx, c);
      }
      mean /= nblk;
      // items ~ copies / 2 + a few single elements and idle slots; the chunk count the mean fits with 0.4 % to spare
      const double mean_items = 0.5 * mean + 6.0;
      const int32_t chunks = static_cast<int32_t>(std::ceil((1.004 * mean_items - 1024.0) / 64.0));
      // 16 chunks of the first round + `chunks` on four SIMDs: only when that is a multiple of four does one chunk more
      // (the second rounding of two separate lists, a block a few items over) cost a SIMD a whole further chunk.  41 or 42
      // chunks put 11 on the busiest SIMD either way - the slabs of the 8-GPU partition: the one-list layout and the repair
      // then only cost their somewhat worse packing (+0.3 ... 0.9 % measured), so they stay off.
      second_list_pays = mx >= 4096 && (16 + std::max(chunks, 0)) % 4 == 0;
      if (second_list_pays) {  // (mx >= 4096: what pick_threads answers with 1024 threads)
        const int32_t budget = 2 * (1024 + 64 * std::max(chunks, 0) - repair_margin);
        chunk_capacity = 1024 + 64 * std::max(chunks, 0);
        // node -> elements
        std::vector<int64_t> adj_off(static_cast<size_t>(n_nodes) + 1, 0);
        for (int64_t i = 0; i < 4 * static_cast<int64_t>(n_elems); ++i) ++adj_off[tets[i] + 1];
        for (int32_t i = 0; i < n_nodes; ++i) adj_off[i + 1] += adj_off[i];
        std::vector<int32_t> adj(adj_off[n_nodes]);
        {
          std::vector<int64_t> cur(adj_off.begin(), adj_off.end() - 1);
          for (int32_t e = 0; e < n_elems; ++e)
            for (int a = 0; a < 4; ++a) adj[cur[tets[4 * static_cast<int64_t>(e) + a]]++] = e;
        }
        std::vector<char> mark(n_elems, 0);
        std::vector<int32_t> touched;
        int32_t repaired = 0, failed = 0;
        for (int32_t b = 0; b + 1 < nblk; b += 2) {
          if (copies[b] <= budget && copies[b + 1] <= budget) continue;
          const int32_t heavy = copies[b] >= copies[b + 1] ? b : b + 1, light = heavy ^ 1;
          if (copies[light] >= budget) {
            ++failed;
            continue;
          }
          const int32_t s0 = block_start[b], s2 = leaf_end(b + 1);
          int32_t k = block_start[b + 1] - s0;
          // the order of the bisection that made the two leaves: along the longest axis of the pair, ties by node id
          double mn[3] = {1e300, 1e300, 1e300}, mxx[3] = {-1e300, -1e300, -1e300};
          for (int32_t i = s0; i < s2; ++i)
            for (int a = 0; a < 3; ++a) {
              const double v = xyz[3 * static_cast<int64_t>(plan.new_to_old[i]) + a];
              mn[a] = std::min(mn[a], v);
              mxx[a] = std::max(mxx[a], v);
            }
          int ax = 0;
          for (int a = 1; a < 3; ++a)
            if (mxx[a] - mn[a] > mxx[ax] - mn[ax]) ax = a;
          std::sort(plan.new_to_old.begin() + s0, plan.new_to_old.begin() + s2,
                    [&](int32_t p, int32_t q) { return axis_order(ax, p, q); });
          // elements touching the pair
          touched.clear();
          for (int32_t i = s0; i < s2; ++i)
            for (int64_t j = adj_off[plan.new_to_old[i]]; j < adj_off[plan.new_to_old[i] + 1]; ++j)
              if (!mark[adj[j]]) {
                mark[adj[j]] = 1;
                touched.push_back(adj[j]);
              }
          for (int32_t e : touched) mark[e] = 0;
          const int32_t other = copies[b] + copies[b + 1];  // (for the extra work of nodes that change sides)
          (void)other;
          auto pair_copies = [&](int32_t kk, int32_t &cl, int32_t &cr) {
            for (int32_t i = s0; i < s2; ++i) owner[plan.new_to_old[i]] = i - s0 < kk ? b : b + 1;
            cl = cr = 0;
            for (int32_t e : touched) {
              bool l = false, r = false;
              for (int a = 0; a < 4; ++a) {
                const int32_t o = owner[tets[4 * static_cast<int64_t>(e) + a]];
                l |= o == b;
                r |= o == b + 1;
              }
              cl += l;
              cr += r;
            }
            if (extra_work)
              for (int32_t i = s0; i < s2; ++i) (i - s0 < kk ? cl : cr) += extra_work[plan.new_to_old[i]];
          };
          const int32_t dir = heavy == b ? -1 : +1, k0 = k, limit = std::max<int32_t>(8, (heavy == b ? k : (s2 - s0 - k)) / 8);
          int32_t cl = 0, cr = 0, best_k = k;
          bool ok = false;
          for (int32_t moved = 2; moved <= limit; moved += 2) {
            const int32_t kk = k0 + dir * moved;
            if (kk < 1 || kk > s2 - s0 - 1) break;
            pair_copies(kk, cl, cr);
            const int32_t ch = heavy == b ? cl : cr, cli = heavy == b ? cr : cl;
            if (cli > budget) break;
            best_k = kk;
            if (ch <= budget) {
              ok = true;
              break;
            }
          }
          k = ok ? best_k : k0;
          pair_copies(k, cl, cr);
          copies[b] = cl;
          copies[b + 1] = cr;
          ok ? ++repaired : ++failed;
          if (diag_env("SAA_PLAN_DEBUG"))
            fprintf(stderr, "plan:   pair %d/%d: %s, cut moved by %d nodes, copies now %d / %d (last trial %d / %d)\n", b, b + 1,
                    ok ? "repaired" : "not repaired", std::abs(k - k0), copies[b], copies[b + 1], cl, cr);
          // the two leaves in their block-local order again
          auto lex = [&](int32_t p, int32_t q) {
            const double *pa = xyz + 3 * static_cast<int64_t>(p), *pb = xyz + 3 * static_cast<int64_t>(q);
            if (pa[0] != pb[0]) return pa[0] < pb[0];
            if (pa[1] != pb[1]) return pa[1] < pb[1];
            if (pa[2] != pb[2]) return pa[2] < pb[2];
            return p < q;
          };
          std::sort(plan.new_to_old.begin() + s0, plan.new_to_old.begin() + s0 + k, lex);
          std::sort(plan.new_to_old.begin() + s0 + k, plan.new_to_old.begin() + s2, lex);
          block_start[b + 1] = s0 + k;
        }
        // Pairs that cannot settle it between themselves (a strip that changes sides brings its new owner more copies
        // than it takes from the old one): the cut ONE level up moves instead, between the pair and the pair next to it,
        // and both pairs are bisected again by weight.
        int32_t repaired4 = 0;
        if (nblk >= 4) {
          auto axis_of = [&](int32_t lo, int32_t hi) {
            double mn[3] = {1e300, 1e300, 1e300}, mxx[3] = {-1e300, -1e300, -1e300};
            for (int32_t i = lo; i < hi; ++i)
              for (int a = 0; a < 3; ++a) {
                const double v = xyz[3 * static_cast<int64_t>(plan.new_to_old[i]) + a];
                mn[a] = std::min(mn[a], v);
                mxx[a] = std::max(mxx[a], v);
              }
            int ax = 0;
            for (int a = 1; a < 3; ++a)
              if (mxx[a] - mn[a] > mxx[ax] - mn[ax]) ax = a;
            return ax;
          };
          auto sort_axis = [&](int32_t lo, int32_t hi, int ax) {
            std::sort(plan.new_to_old.begin() + lo, plan.new_to_old.begin() + hi,
                      [&](int32_t p, int32_t q) { return axis_order(ax, p, q); });
          };
          auto weighted_mid = [&](int32_t lo, int32_t hi) {  // Rcb::split for two leaves
            int64_t total = 0;
            for (int32_t i = lo; i < hi; ++i) total += weight[plan.new_to_old[i]];
            const int64_t want = (total + 1) / 2;
            int64_t acc = 0;
            int32_t kk = 0;
            const int32_t n = hi - lo;
            while (kk < n - 1 && acc + weight[plan.new_to_old[lo + kk]] / 2 < want) acc += weight[plan.new_to_old[lo + kk++]];
            return std::max(1, std::min(n - 1, kk));
          };
          auto lex_sort = [&](int32_t lo, int32_t hi) {
            std::sort(plan.new_to_old.begin() + lo, plan.new_to_old.begin() + hi, [&](int32_t p, int32_t q) {
              const double *pa = xyz + 3 * static_cast<int64_t>(p), *pb = xyz + 3 * static_cast<int64_t>(q);
              if (pa[0] != pb[0]) return pa[0] < pb[0];
              if (pa[1] != pb[1]) return pa[1] < pb[1];
              if (pa[2] != pb[2]) return pa[2] < pb[2];
              return p < q;
            });
          };
          for (int32_t b = 0; b + 3 < nblk; b += 4) {
            bool over = false;
            for (int j = 0; j < 4; ++j) over |= copies[b + j] > budget;
            if (!over) continue;
            if (diag_env("SAA_PLAN_DEBUG"))
              fprintf(stderr, "plan:   blocks %d..%d before: %d %d %d %d\n", b, b + 3, copies[b], copies[b + 1], copies[b + 2], copies[b + 3]);
            const int32_t s0 = block_start[b], s4 = leaf_end(b + 3), m0 = block_start[b + 2] - s0;
            const int32_t heavy_left = (copies[b] + copies[b + 1] >= copies[b + 2] + copies[b + 3]) ? 1 : 0;
            std::vector<int32_t> saved(plan.new_to_old.begin() + s0, plan.new_to_old.begin() + s4);
            const int32_t saved_starts[3] = {block_start[b + 1], block_start[b + 2], block_start[b + 3]};
            touched.clear();
            for (int32_t i = s0; i < s4; ++i)
              for (int64_t j = adj_off[plan.new_to_old[i]]; j < adj_off[plan.new_to_old[i] + 1]; ++j)
                if (!mark[adj[j]]) {
                  mark[adj[j]] = 1;
                  touched.push_back(adj[j]);
                }
            for (int32_t e : touched) mark[e] = 0;
            const int ax = axis_of(s0, s4);
            bool ok = false;
            int32_t c4[4] = {0, 0, 0, 0}, st[3] = {0, 0, 0};
            const int32_t limit = std::max<int32_t>(16, (heavy_left ? m0 : (s4 - s0 - m0)) / 6);
            for (int32_t moved = 4; moved <= limit && !ok; moved += 4) {
              const int32_t m = m0 + (heavy_left ? -moved : moved);
              if (m < 2 || m > s4 - s0 - 2) break;
              sort_axis(s0, s4, ax);
              const int32_t lo2[2] = {s0, s0 + m}, hi2[2] = {s0 + m, s4};
              st[1] = s0 + m;
              for (int h = 0; h < 2; ++h) {
                sort_axis(lo2[h], hi2[h], axis_of(lo2[h], hi2[h]));
                st[2 * h] = lo2[h] + weighted_mid(lo2[h], hi2[h]);
              }
              auto count4 = [&]() {
                const int32_t bounds[5] = {s0, st[0], st[1], st[2], s4};
                for (int j = 0; j < 4; ++j)
                  for (int32_t i = bounds[j]; i < bounds[j + 1]; ++i) owner[plan.new_to_old[i]] = b + j;
                for (int j = 0; j < 4; ++j) c4[j] = 0;
                for (int32_t e : touched) {
                  bool in[4] = {false, false, false, false};
                  for (int a = 0; a < 4; ++a) {
                    const int32_t o = owner[tets[4 * static_cast<int64_t>(e) + a]] - b;
                    if (o >= 0 && o < 4) in[o] = true;
                  }
                  for (int j = 0; j < 4; ++j) c4[j] += in[j];
                }
                if (extra_work)
                  for (int32_t i = s0; i < s4; ++i) c4[owner[plan.new_to_old[i]] - b] += extra_work[plan.new_to_old[i]];
              };
              // (the weights only approximate the copies: the cut inside each pair then follows the copies themselves)
              for (int it = 0; it < 40; ++it) {
                count4();
                bool changed = false;
                for (int h = 0; h < 2; ++h) {
                  const int32_t d = c4[2 * h] - c4[2 * h + 1];
                  if (std::abs(d) <= 12) continue;
                  const int32_t nk = st[2 * h] + (d > 0 ? -2 : 2);
                  if (nk <= lo2[h] + 1 || nk >= hi2[h] - 1) continue;
                  st[2 * h] = nk;
                  changed = true;
                }
                if (!changed) break;
              }
              ok = c4[0] <= budget && c4[1] <= budget && c4[2] <= budget && c4[3] <= budget;
            }
            if (ok) {
              const int32_t bounds[5] = {s0, st[0], st[1], st[2], s4};
              for (int j = 0; j < 4; ++j) {
                lex_sort(bounds[j], bounds[j + 1]);
                copies[b + j] = c4[j];
              }
              block_start[b + 1] = st[0];
              block_start[b + 2] = st[1];
              block_start[b + 3] = st[2];
              ++repaired4;
            } else {  // as it was
              std::copy(saved.begin(), saved.end(), plan.new_to_old.begin() + s0);
              block_start[b + 1] = saved_starts[0];
              block_start[b + 2] = saved_starts[1];
              block_start[b + 3] = saved_starts[2];
              for (int j = 0; j < 4; ++j)
                for (int32_t i = (j == 0 ? s0 : block_start[b + j]); i < (j == 3 ? s4 : block_start[b + j + 1]); ++i)
                  owner[plan.new_to_old[i]] = b + j;
            }
            if (diag_env("SAA_PLAN_DEBUG"))
              fprintf(stderr, "plan:   blocks %d..%d: %s one level up (last trial %d %d %d %d)\n", b, b + 3, ok ? "repaired" : "not repaired",
                      c4[0], c4[1], c4[2], c4[3]);
          }
        }
        if (diag_env("SAA_PLAN_DEBUG"))
          fprintf(stderr, "plan: chunk repair: budget %d copies per block (%d second-phase chunks), %d pairs repaired, %d not, %d groups of "
                          "four one level up\n", budget, chunks, repaired, failed, repaired4);
      }
    }
  } else {
    rcb.split(0, n_nodes, nb);
  }
  block_start.push_back(n_nodes);
  const int32_t n_blocks = static_cast<int32_t>(block_start.size()) - 1;

  plan.old_to_new.resize(n_nodes);
  std::vector<int32_t> node_block(n_nodes);
  for (int32_t b = 0; b < n_blocks; ++b)
    for (int32_t i = block_start[b]; i < block_start[b + 1]; ++i) {
      plan.old_to_new[plan.new_to_old[i]] = i;
      node_block[i] = b;
    }

  // element copies per block: an element belongs to every block owning one of its nodes
  std::vector<int64_t> off(n_blocks + 1, 0);
  auto blocks_of = [&](int32_t e, int32_t out[4]) {
    int cnt = 0;
    for (int a = 0; a < 4; ++a) {
      const int32_t b = node_block[plan.old_to_new[tets[4 * static_cast<int64_t>(e) + a]]];
      bool seen = false;
      for (int j = 0; j < cnt; ++j) seen |= (out[j] == b);
      if (!seen) out[cnt++] = b;
    }
    return cnt;
  };
  for (int32_t e = 0; e < n_elems; ++e) {
    int32_t bs[4];
    const int cnt = blocks_of(e, bs);
    for (int j = 0; j < cnt; ++j) ++off[bs[j] + 1];
  }
  for (int32_t b = 0; b < n_blocks; ++b) off[b + 1] += off[b];
  plan.n_elem_copies = off[n_blocks];
  if (plan.n_elem_copies > INT32_MAX) {
    err = "partition too large: more than 2^31 element copies";
    return false;
  }
  std::vector<int32_t> elem_of(plan.n_elem_copies);
  {
    std::vector<int64_t> cur(off.begin(), off.end() - 1);
    for (int32_t e = 0; e < n_elems; ++e) {
      int32_t bs[4];
      const int cnt = blocks_of(e, bs);
      for (int j = 0; j < cnt; ++j) elem_of[cur[bs[j]]++] = e;
    }
  }

  // ---- per block (serial, cheap): halo list and block-local connectivity ---------------------------
  plan.blocks.resize(n_blocks);
  std::vector<uint16_t> loc(4 * static_cast<size_t>(plan.n_elem_copies));
  std::vector<int32_t> tmp;
  for (int32_t b = 0; b < n_blocks; ++b) {
    BlockDesc &d = plan.blocks[b];
    d.node_start = block_start[b];
    d.n_owned = block_start[b + 1] - block_start[b];
    d.halo_off = static_cast<int32_t>(plan.halo_ids.size());
    const int32_t lo = d.node_start, hi = d.node_start + d.n_owned;
    tmp.clear();
    for (int64_t c = off[b]; c < off[b + 1]; ++c)
      for (int a = 0; a < 4; ++a) {
        const int32_t g = plan.old_to_new[tets[4 * static_cast<int64_t>(elem_of[c]) + a]];
        if (g < lo || g >= hi) tmp.push_back(g);
      }
    std::sort(tmp.begin(), tmp.end());
    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
    d.n_halo = static_cast<int32_t>(tmp.size());
    if (d.n_owned + d.n_halo > kMaxLocalNodes) {
      too_big = true;
      return false;
    }
    for (int64_t c = off[b]; c < off[b + 1]; ++c)
      for (int a = 0; a < 4; ++a) {
        const int32_t g = plan.old_to_new[tets[4 * static_cast<int64_t>(elem_of[c]) + a]];
        int32_t l;
        if (g >= lo && g < hi)
          l = g - lo;
        else
          l = d.n_owned + static_cast<int32_t>(std::lower_bound(tmp.begin(), tmp.end(), g) - tmp.begin());
        loc[4 * static_cast<size_t>(c) + a] = static_cast<uint16_t>(l);
      }
    plan.halo_ids.insert(plan.halo_ids.end(), tmp.begin(), tmp.end());
    plan.max_owned = std::max(plan.max_owned, d.n_owned);
    plan.max_local = std::max(plan.max_local, d.n_owned + d.n_halo);
    plan.n_halo_total += d.n_halo;
  }

  // ---- per block (threads): pair the elements, split interior / boundary items, pack for the LDS ----
  std::vector<std::vector<uint16_t>> block_items(n_blocks);
  std::vector<int32_t> n_interior(n_blocks, 0), n_items(n_blocks, 0), n_paired(n_blocks, 0);
  // Idle-lane padding is OFF by default: measured on MI355X (1M tets) 14.6 us/step without, 15.4 / 16.9 /
  // 17.7 us with 10 / 30 / 50 % padding - the extra sweeps cost more than the bank clashes they remove.
  const char *pad_env = diag_env("SAA_PLAN_PAD");
  const double pad = pad_env ? atof(pad_env) : 0.0;
  const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  const unsigned n_thr = static_cast<unsigned>(std::min<int64_t>(hw, std::max<int32_t>(1, n_blocks / 8)));
  std::vector<PackStats> stats(n_thr);
  std::vector<std::vector<uint16_t>> block_perm(n_blocks);  // non-empty: old -> new local index of the block's owned nodes
  std::vector<std::vector<uint16_t>> block_halo_perm(n_blocks);  // non-empty: old -> new position in the block's halo list
  const char *shp_env = diag_env("SAA_PLAN_SHAPE_PAIRS");
  const bool shape_pairs = !(shp_env && shp_env[0] == '0');
  const char *force_env = diag_env("SAA_PLAN_FORCE_TRIALS");  // (experiments: try the other numberings on every block)
  const bool force_trials = force_env && force_env[0] == '1';
  const char *lat_env = diag_env("SAA_PLAN_LATTICE_ORDERS");
  const bool lattice_orders = !(lat_env && lat_env[0] == '0');
  std::atomic<int32_t> renumbered{0};
  std::atomic<int32_t> q_hist[13] = {};
  const char *alt_env = diag_env("SAA_PLAN_FIXED_AXES");
  const bool alt_axes = !(alt_env && alt_env[0] == '1');
  // One block per CU and 1024-thread workgroups (api: pick_threads): the resident kernel runs a block's first 1024
  // interior items before the halo arrives and EVERYTHING else as one second list - further interior items together with
  // the boundary items, packed together.  As two lists (interior remainder, boundary) each was rounded up to whole 64-item
  // chunks: a block's ~2530 items then are 41 chunks for 74 of the 256 blocks of the 1M-tet beam instead of 40, one of the
  // four SIMDs gets 11 chunks instead of 10, and since every block waits for its neighbours those blocks pace all.
  int64_t max_copies = 0;
  for (int32_t b = 0; b < n_blocks; ++b) max_copies = std::max<int64_t>(max_copies, off[b + 1] - off[b]);
  const char *cap_env = diag_env("SAA_PLAN_FIRST_ROUND_CAP");
  const bool cap_on = !(cap_env && cap_env[0] == '0');
  const int32_t first_round_cap = (cap_on && second_list_pays && n_blocks <= 256 && max_copies >= 4096) ? 1024 : INT32_MAX;
  std::atomic<int32_t> next{0};
  auto work = [&](unsigned t) {
    std::vector<uint16_t> items, items_s, part_a, part_b, part_q, loc_b, trial, pi;
    std::vector<double> xl;
    std::vector<char> interior;
    for (int32_t b = next.fetch_add(1); b < n_blocks; b = next.fetch_add(1)) {
      const BlockDesc &d = plan.blocks[b];
      const int32_t ne = static_cast<int32_t>(off[b + 1] - off[b]);
      loc_b.assign(loc.begin() + 4 * off[b], loc.begin() + 4 * off[b + 1]);
      // coordinates of the block-local nodes and the block's mesh size (cube root of six mean element volumes)
      xl.resize(3 * static_cast<size_t>(d.n_owned + d.n_halo));
      for (int32_t l = 0; l < d.n_owned + d.n_halo; ++l) {
        const int32_t g = l < d.n_owned ? d.node_start + l : plan.halo_ids[d.halo_off + (l - d.n_owned)];
        for (int k = 0; k < 3; ++k) xl[3 * static_cast<size_t>(l) + k] = xyz[3 * static_cast<int64_t>(plan.new_to_old[g]) + k];
      }
      double h_mesh = 0.0;
      {
        double vol = 0.0;
        for (int32_t e = 0; e < ne; ++e) {
          const double *x0 = &xl[3 * static_cast<size_t>(loc_b[4 * e])], *x1 = &xl[3 * static_cast<size_t>(loc_b[4 * e + 1])],
                       *x2 = &xl[3 * static_cast<size_t>(loc_b[4 * e + 2])], *x3 = &xl[3 * static_cast<size_t>(loc_b[4 * e + 3])];
          const double a[3] = {x1[0] - x0[0], x1[1] - x0[1], x1[2] - x0[2]}, bb[3] = {x2[0] - x0[0], x2[1] - x0[1], x2[2] - x0[2]},
                       c[3] = {x3[0] - x0[0], x3[1] - x0[1], x3[2] - x0[2]};
          vol += std::fabs(a[0] * (bb[1] * c[2] - bb[2] * c[1]) - a[1] * (bb[0] * c[2] - bb[2] * c[0]) + a[2] * (bb[0] * c[1] - bb[1] * c[0]));
        }
        h_mesh = ne > 0 ? std::cbrt(vol / ne) : 0.0;  // |detJ| = 6 V
      }
      // pairing in the order of the caller's element list, and in the spatial order; the latter when it leaves at least
      // 1 % fewer items (a mesh numbered cell by cell pairs best as it comes: 99.9 % of the elements of the structured
      // beams, against 97.8 % in the spatial order)
      int32_t ni = build_items(loc_b, ne, d.n_owned, items);
      if (shape_pairs && h_mesh > 0.0) {
        const int32_t ni_s = build_items(loc_b, ne, d.n_owned, items_s, xl.data(), h_mesh);
        bool spatial = false;
        if (100 * static_cast<int64_t>(ni_s) < 99 * static_cast<int64_t>(ni)) {
          items.swap(items_s);
          ni = ni_s;
          spatial = true;
        }
        // an unstructured mesh (more than 1.5 % of the block's elements single either way; lattices: 0.2 %): the same
        // matching improved by augmenting paths
        if (2000 * static_cast<int64_t>(ni) > 1015 * static_cast<int64_t>(ne)) {
          const int32_t ni_a = spatial ? build_items(loc_b, ne, d.n_owned, items_s, xl.data(), h_mesh, true)
                                       : build_items(loc_b, ne, d.n_owned, items_s, nullptr, 0.0, true);
          if (ni_a < ni) {
            items.swap(items_s);
            ni = ni_a;
          }
        }
      }
      // interior items (every real vertex owned) first: they can run before the halo records arrive
      std::vector<uint16_t> &out = block_items[b];
      out.resize(8 * static_cast<size_t>(ni));
      int32_t n_in = 0, paired = 0;
      interior.assign(ni, 0);
      for (int32_t i = 0; i < ni; ++i) {
        const uint16_t *it = &items[8 * static_cast<size_t>(i)];
        bool in = true;
        for (int a = 0; a < (it[5] ? 5 : 4); ++a) in &= it[a] < d.n_owned;
        interior[i] = in;
        n_in += in;
        paired += it[5];
      }
      if (n_in > first_round_cap) {  // interior items beyond the first round join the second list
        int32_t kept = 0;
        for (int32_t i = 0; i < ni; ++i)
          if (interior[i] && ++kept > first_round_cap) interior[i] = 0;
        n_in = first_round_cap;
      }
      int32_t wi = 0, wb = n_in;
      for (int32_t i = 0; i < ni; ++i) {
        const int32_t dst = interior[i] ? wi++ : wb++;
        std::copy(&items[8 * static_cast<size_t>(i)], &items[8 * static_cast<size_t>(i)] + 8,
                  &out[8 * static_cast<size_t>(dst)]);
      }
      // One numbering of the block's nodes, packed: q = 0 the plan order (lexicographic by exact coordinates, x fastest),
      // 1..5 the other lexicographic orders of the owned nodes, 6..11 pseudo-lattice numberings of owned AND halo nodes
      // (block_lattice_order).  cost = extra LDS passes, weighted as in reorder_for_lds (12 per read level, 21 per atomic).
      struct Trial {
        std::vector<uint16_t> pa, pb, pio, pih;
        PackStats si, sb;
        int32_t mi = 0, mb = 0;
        double cost = 0.0;
      };
      auto evaluate = [&](int q, Trial &tr) {
        trial.assign(out.begin(), out.end());
        tr.pio.clear();
        tr.pih.clear();
        if (q >= 1 && q < 6) block_axis_order(xyz, plan.new_to_old.data() + d.node_start, d.n_owned, q, tr.pio);
        if (q >= 6) {
          block_lattice_order(xyz, plan.new_to_old.data() + d.node_start, d.n_owned, h_mesh, q - 6, tr.pio);
          if (d.n_halo > 0) {  // halo list in the same pseudo-lattice: the boundary items form classes too
            const LatticeColours lc = lattice_colours(xyz, plan.new_to_old.data() + d.node_start, d.n_owned, h_mesh, q - 6);
            std::vector<int> colour(d.n_halo);
            for (int32_t hh = 0; hh < d.n_halo; ++hh)
              colour[hh] = lc(xyz + 3 * static_cast<int64_t>(plan.new_to_old[plan.halo_ids[d.halo_off + hh]]));
            assign_by_colour(colour, d.n_owned, tr.pih);
            relabel_halo(trial.data() + 8 * static_cast<size_t>(n_in), ni - n_in, d.n_owned, tr.pih);
          }
        }
        if (!tr.pio.empty()) relabel_owned(trial.data(), ni, d.n_owned, tr.pio);
        tr.si = PackStats();
        tr.sb = PackStats();
        tr.mi = reorder_for_lds(trial.data(), n_in, d.n_owned, pad, tr.pa, tr.si);
        tr.mb = reorder_for_lds(trial.data() + 8 * static_cast<size_t>(n_in), ni - n_in, d.n_owned, pad, tr.pb, tr.sb);
        tr.cost = 12.0 * (tr.si.read_mult + tr.sb.read_mult - tr.si.read_cnt - tr.sb.read_cnt) +
                  21.0 * (tr.si.atomic_mult + tr.sb.atomic_mult - tr.si.atomic_cnt - tr.sb.atomic_cnt);
      };
      // Which axis runs fastest inside the block decides how well its items pack: with L layers along the (outer, middle,
      // inner) axes the lattice index is inner + L_in * (middle + L_mid * outer), and the translates of an element pair
      // reach all 32 bank residues only if those strides are not all multiples of 8 - a 12 x 8 x 8-node box numbered
      // (x, y, z) packs a quarter of its interior items clash-free, numbered (y, z, x) three quarters; ragged layers at the
      // block faces and the residues of the halo slots decide the rest.  So every numbering is tried where the plan order
      // leaves something to gain, and the cheapest becomes the block's (it has to beat the plan order by 3 %).
      Trial best, cand;
      evaluate(0, best);
      int best_q = 0;
      const int64_t constr0 = best.si.by_construction + best.sb.by_construction;
      if (alt_axes && ni >= 256 && (force_trials || 10 * constr0 < 7 * static_cast<int64_t>(best.mi + best.mb))) {
        const double cost0 = best.cost;
        // (with the pseudo-lattice numberings available the five other exact orders are not tried: on the 1M-tet beams they
        // won in 8 of 512 blocks, and every trial packs the whole block)
        const bool lattice = lattice_orders && h_mesh > 0.0;
        for (int q = lattice ? 6 : 1; q < (lattice ? 12 : 6); ++q) {
          evaluate(q, cand);
          if (cand.cost < 0.97 * cost0 && cand.cost < best.cost) {
            std::swap(best, cand);
            best_q = q;
          }
        }
      }
      // no lattice structure to speak of (less than 30 % of the items in classes under the best numbering so far): the
      // numbering decided while the groups are formed (joint_pack_list)
      if (alt_axes && ni >= 256 &&
          10 * (best.si.by_construction + best.sb.by_construction) < 3 * static_cast<int64_t>(best.mi + best.mb)) {
        const double cost0 = best.cost;
        JointState js;
        js.n_owned = d.n_owned;
        js.n_halo = d.n_halo;
        js.res.assign(static_cast<size_t>(d.n_owned + d.n_halo), -1);
        for (int c = 0; c < 32; ++c) {
          js.room_owned[c] = c < d.n_owned ? (d.n_owned - c + 31) / 32 : 0;
          js.room_halo[c] = 0;
        }
        for (int32_t i = d.n_owned; i < d.n_owned + d.n_halo; ++i) ++js.room_halo[i & 31];
        cand.pio.clear();
        cand.pih.clear();
        cand.mi = joint_pack_list(out.data(), n_in, js, cand.pa, pad);
        cand.mb = joint_pack_list(out.data() + 8 * static_cast<size_t>(n_in), ni - n_in, js, cand.pb, pad);
        auto colours = [&](int32_t first, int32_t n, int32_t *room, std::vector<int> &col) {
          col.resize(n);
          for (int32_t l = 0; l < n; ++l) {
            int c = js.res[first + l];
            if (c < 0) {  // never named by an item: wherever there is room
              c = 0;
              for (int k = 1; k < 32; ++k)
                if (room[k] > room[c]) c = k;
              --room[c];
            }
            col[l] = c;
          }
        };
        std::vector<int> col;
        colours(0, d.n_owned, js.room_owned, col);
        assign_by_colour(col, 0, cand.pio);
        if (d.n_halo > 0) {
          colours(d.n_owned, d.n_halo, js.room_halo, col);
          assign_by_colour(col, d.n_owned, cand.pih);
        }
        for (std::vector<uint16_t> *lst : {&cand.pa, &cand.pb}) {
          const int32_t m = static_cast<int32_t>(lst->size() / 8);
          // (halo first: both renamings keep a vertex on its side of n_owned)
          if (!cand.pih.empty()) relabel_halo(lst->data(), m, d.n_owned, cand.pih);
          relabel_owned(lst->data(), m, d.n_owned, cand.pio);
        }
        cand.si = PackStats();
        cand.sb = PackStats();
        measure_pack(cand.pa.data(), cand.mi, d.n_owned, cand.si);
        measure_pack(cand.pb.data(), cand.mb, d.n_owned, cand.sb);
        cand.cost = 12.0 * (cand.si.read_mult + cand.sb.read_mult - cand.si.read_cnt - cand.sb.read_cnt) +
                    21.0 * (cand.si.atomic_mult + cand.sb.atomic_mult - cand.si.atomic_cnt - cand.sb.atomic_cnt);
        if (cand.cost < 0.97 * cost0) {
          std::swap(best, cand);
          best_q = 12;
        }
      }
      if (!best.pio.empty()) block_perm[b] = best.pio;
      if (!best.pih.empty()) block_halo_perm[b] = best.pih;
      renumbered += best_q >= 6;
      if (diag_env("SAA_PLAN_DEBUG")) ++q_hist[best_q];
      part_a.swap(best.pa);
      part_b.swap(best.pb);
      const PackStats st_in = best.si, st_bd = best.sb;
      int32_t m_in = best.mi, m_bd = best.mb;
      if (m_in > first_round_cap) {  // idle slots of the packing pushed the first list past one round: its tail goes last
        for (int32_t sl = first_round_cap; sl < m_in; ++sl) {
          const uint16_t *it = &part_a[8 * static_cast<size_t>(sl)];
          if (it[5] == 2) continue;
          part_b.insert(part_b.end(), it, it + 8);
          ++m_bd;
        }
        part_a.resize(8 * static_cast<size_t>(first_round_cap));
        m_in = first_round_cap;
      }
      stats[t].add(st_in);
      stats[t].add(st_bd);
      out = part_a;
      out.insert(out.end(), part_b.begin(), part_b.end());
      n_interior[b] = m_in;
      n_items[b] = m_in + m_bd;
      n_paired[b] = paired;
    }
  };
  {
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n_thr; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
  }
  // blocks that re-ordered their halo list
  for (int32_t b = 0; b < n_blocks; ++b) {
    const std::vector<uint16_t> &ph = block_halo_perm[b];
    if (ph.empty()) continue;
    int32_t *seg = plan.halo_ids.data() + plan.blocks[b].halo_off;
    std::vector<int32_t> moved(ph.size());
    for (size_t h = 0; h < ph.size(); ++h) moved[ph[h]] = seg[h];
    std::copy(moved.begin(), moved.end(), seg);
  }
  if (diag_env("SAA_PLAN_DEBUG")) {
    fprintf(stderr, "plan: %d of %d blocks took a pseudo-lattice numbering; blocks per numbering 0..12 (12: decided while packing):", renumbered.load(), n_blocks);
    for (auto &h : q_hist) fprintf(stderr, " %d", h.load());
    fprintf(stderr, "\n");
  }
  // blocks that changed their internal order: the numbering (and with it every halo list that names their nodes) follows
  {
    int32_t n_changed = 0;
    for (int32_t b = 0; b < n_blocks; ++b) n_changed += !block_perm[b].empty();
    if (diag_env("SAA_PLAN_DEBUG")) fprintf(stderr, "plan: %d of %d blocks took another axis order\n", n_changed, n_blocks);
    plan.n_renumbered = n_changed;
    if (n_changed > 0) {
      std::vector<int32_t> moved(plan.new_to_old);
      for (int32_t b = 0; b < n_blocks; ++b) {
        const std::vector<uint16_t> &pb = block_perm[b];
        if (pb.empty()) continue;
        const int32_t s0 = plan.blocks[b].node_start;
        for (size_t l = 0; l < pb.size(); ++l) moved[s0 + pb[l]] = plan.new_to_old[s0 + l];
      }
      plan.new_to_old.swap(moved);
      for (int32_t i = 0; i < n_nodes; ++i) plan.old_to_new[plan.new_to_old[i]] = i;
      for (int32_t &g : plan.halo_ids) {
        const int32_t ob = static_cast<int32_t>(std::upper_bound(block_start.begin(), block_start.end(), g) - block_start.begin()) - 1;
        if (!block_perm[ob].empty()) g = block_start[ob] + block_perm[ob][g - block_start[ob]];
      }
    }
  }
  int64_t total_items = 0;
  for (int32_t b = 0; b < n_blocks; ++b) total_items += n_items[b];
  if (overflow && chunk_capacity > 0 && first_round_cap != INT32_MAX)
    for (int32_t b = 0; b < n_blocks; ++b) *overflow = std::max(*overflow, n_items[b] - chunk_capacity);
  plan.conn.resize(8 * static_cast<size_t>(total_items));
  int64_t pos = 0;
  for (int32_t b = 0; b < n_blocks; ++b) {
    BlockDesc &d = plan.blocks[b];
    d.elem_off = static_cast<int32_t>(pos);
    d.n_elem = n_items[b];
    d.n_interior = n_interior[b];
    d.reserved = 0;
    std::copy(block_items[b].begin(), block_items[b].end(), plan.conn.begin() + 8 * pos);
    pos += n_items[b];
    plan.n_pairs += n_paired[b];
  }
  plan.n_items = total_items;
  plan.conn_packed.resize(static_cast<size_t>(total_items));
  for (int64_t i = 0; i < total_items; ++i) {
    const uint16_t *it = &plan.conn[8 * static_cast<size_t>(i)];
    uint64_t w = static_cast<uint64_t>(it[5] & 3u) << 60;
    for (int a = 0; a < 5; ++a) w |= static_cast<uint64_t>(it[a] & 0xfffu) << (12 * a);
    plan.conn_packed[i] = w;
  }
  PackStats tot;
  for (const auto &st : stats) {
    tot.read_mult += st.read_mult;
    tot.read_cnt += st.read_cnt;
    tot.atomic_mult += st.atomic_mult;
    tot.atomic_cnt += st.atomic_cnt;
    tot.by_construction += st.by_construction;
  }
  plan.lds_conflict_factor = tot.read_cnt ? tot.read_mult / tot.read_cnt : 1.0;
  plan.lds_atomic_conflict_factor = tot.atomic_cnt ? tot.atomic_mult / tot.atomic_cnt : 1.0;
  plan.n_by_construction = tot.by_construction;
  if (diag_env("SAA_PLAN_DEBUG")) {
    // distribution of the per-block work (items) and of its interior / boundary split
    int32_t mn = INT32_MAX, mx = 0, mxi = 0, mxb = 0;
    double sum = 0;
    for (const auto &b : plan.blocks) {
      mn = std::min(mn, b.n_elem); mx = std::max(mx, b.n_elem); sum += b.n_elem;
      mxi = std::max(mxi, b.n_interior); mxb = std::max(mxb, b.n_elem - b.n_interior);
    }
    fprintf(stderr, "plan: items per block min %d mean %.1f max %d; max interior %d, max boundary %d\n", mn,
            sum / plan.blocks.size(), mx, mxi, mxb);
    {  // second-phase chunks of 64 items per block at 1024 threads (what decides how evenly the four SIMDs are loaded)
      int ch[64] = {};
      for (const auto &b : plan.blocks) {
        const int n_pre = std::min(b.n_interior, 1024), n_ir = b.n_interior - n_pre;
        const int n_post = ((n_ir + 63) & ~63) + (b.n_elem - b.n_interior);
        ++ch[std::min(63, (n_post + 63) / 64)];
      }
      fprintf(stderr, "plan: blocks by second-phase chunks:");
      for (int c = 0; c < 64; ++c)
        if (ch[c]) fprintf(stderr, " %d:%d", c, ch[c]);
      fprintf(stderr, "\n");
      int ch1[64] = {}, over = 0;  // with ONE rounding: ceil((items - 1024) / 64)
      for (const auto &b : plan.blocks) {
        ++ch1[std::min(63, (std::max(b.n_elem - 1024, 0) + 63) / 64)];
        over += b.n_elem > 2560;
      }
      fprintf(stderr, "plan: the same with the interior remainder unpadded:");
      for (int c = 0; c < 64; ++c)
        if (ch1[c]) fprintf(stderr, " %d:%d", c, ch1[c]);
      fprintf(stderr, "; blocks above 2560 items: %d\n", over);
    }
    int hist[8][8] = {};
    for (const auto &b : plan.blocks)
      hist[std::min(7, (b.n_interior + 1023) / 1024)][std::min(7, (b.n_elem - b.n_interior + 1023) / 1024)]++;
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 8; ++j)
        if (hist[i][j]) fprintf(stderr, "plan:   %d blocks with %d interior + %d boundary rounds of 1024 items\n", hist[i][j], i, j);
  }
  if (diag_env("SAA_PLAN_DEBUG"))
    fprintf(stderr,
            "plan: %lld element copies in %lld items (%lld pairs, %lld in clash-free halves by construction); read conflict "
            "factor %.3f, atomic %.3f (%u threads)\n",
            static_cast<long long>(plan.n_elem_copies), static_cast<long long>(plan.n_items),
            static_cast<long long>(plan.n_pairs), static_cast<long long>(tot.by_construction), plan.lds_conflict_factor,
            tot.atomic_cnt ? tot.atomic_mult / tot.atomic_cnt : 1.0, n_thr);
  return true;
}

}  // namespace

bool build_plan(int32_t n_nodes, int32_t n_elems, const double *xyz, const int32_t *tets,
                int32_t block_nodes, Plan &plan, std::string &err, const int32_t *extra_work) {
  if (n_nodes <= 0 || n_elems < 0 || !xyz || (n_elems > 0 && !tets)) {
    err = "build_plan: empty mesh or null pointer";
    return false;
  }
  for (int64_t i = 0; i < 4 * static_cast<int64_t>(n_elems); ++i)
    if (tets[i] < 0 || tets[i] >= n_nodes) {
      err = "build_plan: element " + std::to_string(i / 4) + " references node " +
            std::to_string(tets[i]) + " outside [0," + std::to_string(n_nodes) + ")";
      return false;
    }
  for (int64_t i = 0; i < 3 * static_cast<int64_t>(n_nodes); ++i)
    if (!std::isfinite(xyz[i])) {
      err = "build_plan: non-finite coordinate at node " + std::to_string(i / 3);
      return false;
    }
  int32_t bn = block_nodes > 0 ? block_nodes : auto_block_nodes(n_nodes);
  bn = std::min(bn, kMaxLocalNodes);
  int32_t margin = 8;
  bool retried = false;
  while (true) {
    bool too_big = false;
    int32_t overflow = 0;
    if (build_once(n_nodes, n_elems, xyz, tets, bn, plan, err, too_big, extra_work, margin, &overflow)) {
      // a few blocks a few items over the chunk count the others keep (more single elements than the margin allowed for:
      // meshes that pair less completely than a lattice): once more, with that much more room
      if (overflow > 0 && overflow <= 24 && !retried) {
        retried = true;
        margin += overflow + 2;
        if (diag_env("SAA_PLAN_DEBUG")) fprintf(stderr, "plan: fullest block %d items over its chunks: again with a margin of %d\n", overflow, margin);
        continue;
      }
      return true;
    }
    if (!too_big) return false;
    if (bn <= 8) {
      err = "build_plan: a node block exceeds the LDS budget even with 8 owned nodes "
            "(a node has more than ~2500 neighbours)";
      return false;
    }
    bn /= 2;
  }
}

}  // namespace saa
