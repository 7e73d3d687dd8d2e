// k-way element partition of a tetrahedral mesh: the role ParMETIS' part_mesh_kway plays in the reference
// (/root/reference Data_prepare.py:82-101: one part per rank, elements of a part go to one GPU).
//
// Graph partitioning on the DUAL graph (elements adjacent across a shared face), in the METIS family's style without
// the multilevel hierarchy: recursive bisection, each bisection = greedy graph growing from both ends of a
// pseudo-diameter (the better of the two seeds wins) followed by Fiduccia-Mattheyses boundary refinement with rollback
// to the best prefix.  What the solver cares about is the number of interface NODES (LSTM input width, bytes pushed to
// the neighbour ranks per step), which the face cut tracks closely.  Deterministic: same mesh, same partition on every
// rank - the ranks run it redundantly instead of communicating (the mesh is replicated, Data_prepare.py:76-79).
// Pure host C++, O(Ne log Ne) for the dual graph, O(Ne log k) afterwards.
#include "saa_partition.h"

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>

namespace saa {
namespace {

inline int64_t iabs(int64_t v) { return v < 0 ? -v : v; }

struct Dual {
  int32_t n = 0;
  std::vector<int32_t> adj;  // 4 slots per element, -1 = boundary face
  const int32_t *nb(int32_t e) const { return &adj[4 * static_cast<size_t>(e)]; }
};

void build_dual(int32_t n_elems, const int32_t *tets, Dual &g) {
  struct Face {
    int32_t a, b, c, e;
  };
  std::vector<Face> faces(4 * static_cast<size_t>(n_elems));
  for (int32_t e = 0; e < n_elems; ++e)
    for (int k = 0; k < 4; ++k) {
      int32_t t[3];
      int m = 0;
      for (int a = 0; a < 4; ++a)
        if (a != k) t[m++] = tets[4 * static_cast<size_t>(e) + a];
      if (t[0] > t[1]) std::swap(t[0], t[1]);
      if (t[1] > t[2]) std::swap(t[1], t[2]);
      if (t[0] > t[1]) std::swap(t[0], t[1]);
      faces[4 * static_cast<size_t>(e) + k] = {t[0], t[1], t[2], e};
    }
  std::sort(faces.begin(), faces.end(), [](const Face &x, const Face &y) {
    if (x.a != y.a) return x.a < y.a;
    if (x.b != y.b) return x.b < y.b;
    if (x.c != y.c) return x.c < y.c;
    return x.e < y.e;
  });
  g.n = n_elems;
  g.adj.assign(4 * static_cast<size_t>(n_elems), -1);
  std::vector<uint8_t> fill(n_elems, 0);
  for (size_t i = 0; i + 1 < faces.size(); ++i) {
    const Face &x = faces[i], &y = faces[i + 1];
    if (x.a != y.a || x.b != y.b || x.c != y.c || x.e == y.e) continue;
    if (fill[x.e] < 4 && fill[y.e] < 4) {  // (a face shared by three or more elements links consecutive pairs)
      g.adj[4 * static_cast<size_t>(x.e) + fill[x.e]++] = y.e;
      g.adj[4 * static_cast<size_t>(y.e) + fill[y.e]++] = x.e;
    }
    ++i;  // a face pairs two elements: skip the partner
  }
}

// One bisection of the elements carrying `tag` in `part` into sides 0 / 1 (written to `side`).
class Bisector {
 public:
  Bisector(const Dual &g, std::vector<int32_t> &part) : g_(g), part_(part), side_(g.n, 0), mark_(g.n, 0), gain_(g.n, 0), lock_(g.n, 0) {}

  // elems: the subset (all have part_[e] == tag).  On return side_[e] in {0,1}; returns the face cut.
  int64_t run(const std::vector<int32_t> &elems, int32_t tag, int64_t target_left) {
    tag_ = tag;
    const int32_t far0 = farthest(elems, elems.front());
    const int32_t far1 = farthest(elems, far0);
    int64_t best_cut = -1;
    std::vector<uint8_t> best;
    for (int trial = 0; trial < 3; ++trial) {
      // trial 0 / 1: side 0 grown from far0, or side 1 from far1 (mirror image: side 0 is what is left over);
      // trial 2: both ends at once - the elements in the order of (distance to far0) - (distance to far1), the first
      // target_left of them on side 0: the front is the equidistance surface of the two ends, a cross-section of an
      // elongated part where a one-sided growth gives a spherical cap
      if (trial < 2)
        grow(elems, trial == 0 ? far0 : far1,
             trial == 0 ? target_left : static_cast<int64_t>(elems.size()) - target_left, trial == 0 ? 0 : 1);
      else
        split_by_distance_difference(elems, far0, far1, target_left);
      const int64_t cut = refine(elems, target_left);
      if (best_cut < 0 || cut < best_cut) {
        best_cut = cut;
        best.resize(elems.size());
        for (size_t i = 0; i < elems.size(); ++i) best[i] = side_[elems[i]];
      }
    }
    for (size_t i = 0; i < elems.size(); ++i) side_[elems[i]] = best[i];
    return best_cut;
  }
  uint8_t side(int32_t e) const { return side_[e]; }

 private:
  bool in(int32_t e) const { return e >= 0 && part_[e] == tag_; }

  // last element reached by a breadth-first sweep from `start` inside the subset (other components: ignored)
  int32_t farthest(const std::vector<int32_t> &elems, int32_t start) {
    ++stamp_;
    queue_.clear();
    queue_.push_back(start);
    mark_[start] = stamp_;
    for (size_t h = 0; h < queue_.size(); ++h) {
      const int32_t e = queue_[h];
      for (int k = 0; k < 4; ++k) {
        const int32_t f = g_.nb(e)[k];
        if (in(f) && mark_[f] != stamp_) {
          mark_[f] = stamp_;
          queue_.push_back(f);
        }
      }
    }
    (void)elems;
    return queue_.back();
  }

  // breadth-first distances from `start` inside the subset into dist (unreached elements keep `far`)
  void distances(const std::vector<int32_t> &elems, int32_t start, std::vector<int32_t> &dist, int32_t far) {
    for (int32_t e : elems) dist[e] = far;
    queue_.clear();
    queue_.push_back(start);
    dist[start] = 0;
    for (size_t h = 0; h < queue_.size(); ++h) {
      const int32_t e = queue_[h];
      for (int k = 0; k < 4; ++k) {
        const int32_t f = g_.nb(e)[k];
        if (in(f) && dist[f] == far) {
          dist[f] = dist[e] + 1;
          queue_.push_back(f);
        }
      }
    }
  }

  void split_by_distance_difference(const std::vector<int32_t> &elems, int32_t a, int32_t b, int64_t target_left) {
    if (d0_.empty()) {
      d0_.assign(g_.n, 0);
      d1_.assign(g_.n, 0);
    }
    const int32_t far = g_.n + 1;
    distances(elems, a, d0_, far);
    distances(elems, b, d1_, far);
    order_.assign(elems.begin(), elems.end());
    // ties (a whole layer of equal difference): by the distance to `a`, then by id - keeps the split layer compact
    std::sort(order_.begin(), order_.end(), [&](int32_t x, int32_t y) {
      const int32_t kx = d0_[x] - d1_[x], ky = d0_[y] - d1_[y];
      if (kx != ky) return kx < ky;
      if (d0_[x] != d0_[y]) return d0_[x] < d0_[y];
      return x < y;
    });
    for (size_t i = 0; i < order_.size(); ++i) side_[order_[i]] = static_cast<int64_t>(i) < target_left ? 0 : 1;
  }

  // graph growing: the region (side `s`) is the `target` elements nearest to the seed in the dual graph (breadth-first
  // order, so its front is one BFS level thick at most - greedy max-gain growing was tried first and builds dendrites
  // on tetrahedra, whose four faces give too coarse a gain); disconnected subsets restart from any element not yet taken
  void grow(const std::vector<int32_t> &elems, int32_t seed, int64_t target, uint8_t s) {
    for (int32_t e : elems) side_[e] = static_cast<uint8_t>(1 - s);
    ++stamp_;
    queue_.clear();
    int64_t taken = 0;
    size_t head = 0, scan = 0;
    auto push = [&](int32_t e) {
      mark_[e] = stamp_;
      queue_.push_back(e);
    };
    if (target > 0) push(seed);
    while (taken < target) {
      if (head == queue_.size()) {  // component exhausted: next one
        while (scan < elems.size() && mark_[elems[scan]] == stamp_) ++scan;
        if (scan == elems.size()) break;
        push(elems[scan]);
      }
      const int32_t e = queue_[head++];
      side_[e] = s;
      ++taken;
      for (int k = 0; k < 4; ++k) {
        const int32_t f = g_.nb(e)[k];
        if (in(f) && mark_[f] != stamp_) push(f);
      }
    }
  }

  int cut_gain(int32_t e) const {  // faces cut now minus faces cut after moving e to the other side
    int g = 0;
    for (int k = 0; k < 4; ++k) {
      const int32_t f = g_.nb(e)[k];
      if (in(f)) g += side_[f] != side_[e] ? 1 : -1;
    }
    return g;
  }

  // Fiduccia-Mattheyses passes on the boundary; balance: | |side 0| - target_left | <= tol.
  int64_t refine(const std::vector<int32_t> &elems, int64_t target_left) {
    const int64_t n = static_cast<int64_t>(elems.size());
    const int64_t tol = n >= 64 ? std::max<int64_t>(1, n / 400) : 0;  // 0.25 % per bisection; tiny subsets stay exact
    int64_t left = 0, cut = 0;
    for (int32_t e : elems) {
      left += side_[e] == 0;
      for (int k = 0; k < 4; ++k) {
        const int32_t f = g_.nb(e)[k];
        cut += (in(f) && f > e && side_[f] != side_[e]);
      }
    }
    for (int pass = 0; pass < 12; ++pass) {
      ++stamp_;  // lock_ == stamp_: moved in this pass
      for (int s = 0; s < 2; ++s)
        for (auto &b : fm_[s]) b.clear();
      for (int32_t e : elems) {
        const int g = cut_gain(e);
        gain_[e] = g;
        bool boundary = false;
        for (int k = 0; k < 4; ++k) boundary |= (in(g_.nb(e)[k]) && side_[g_.nb(e)[k]] != side_[e]);
        if (boundary) fm_[side_[e]][g + 4].push_back(e);
      }
      moves_.clear();
      int64_t best_cut = cut, cur = cut, cur_left = left;
      size_t best_len = 0;
      int64_t best_dev = iabs(left - target_left);
      const size_t patience = static_cast<size_t>(std::min<int64_t>(n, 64 + n / 50));
      while (moves_.size() - best_len < patience) {
        // the better-gain head of the two sides among those the balance admits; when out of balance only moves
        // towards it are admitted
        int32_t pick = -1;
        for (int b = 8; b >= 0 && pick < 0; --b)
          for (int s = 0; s < 2 && pick < 0; ++s) {
            const int64_t after = cur_left + (s == 0 ? -1 : 1);
            if (iabs(after - target_left) > std::max(tol, iabs(cur_left - target_left) - 1)) continue;
            auto &q = fm_[s][b];
            while (!q.empty()) {
              const int32_t e = q.back();
              q.pop_back();
              if (lock_[e] != stamp_ && side_[e] == s && gain_[e] + 4 == b) {
                pick = e;
                break;
              }
            }
          }
        if (pick < 0) break;
        const uint8_t s = side_[pick];
        cur -= gain_[pick];
        cur_left += s == 0 ? -1 : 1;
        side_[pick] = static_cast<uint8_t>(1 - s);
        lock_[pick] = stamp_;
        moves_.push_back(pick);
        for (int k = 0; k < 4; ++k) {
          const int32_t f = g_.nb(pick)[k];
          if (!in(f) || lock_[f] == stamp_) continue;
          gain_[f] = cut_gain(f);
          fm_[side_[f]][gain_[f] + 4].push_back(f);
        }
        const int64_t dev = iabs(cur_left - target_left);
        if (cur < best_cut || (cur == best_cut && dev < best_dev)) {
          best_cut = cur;
          best_dev = dev;
          best_len = moves_.size();
        }
      }
      for (size_t i = moves_.size(); i > best_len; --i) {  // roll back to the best prefix
        const int32_t e = moves_[i - 1];
        side_[e] = static_cast<uint8_t>(1 - side_[e]);
      }
      left = 0;
      for (int32_t e : elems) left += side_[e] == 0;
      if (best_cut == cut && best_len == 0) break;
      const bool improved = best_cut < cut;
      cut = best_cut;
      if (!improved && iabs(left - target_left) <= tol) break;
    }
    return cut;
  }

  const Dual &g_;
  std::vector<int32_t> &part_;
  std::vector<uint8_t> side_;
  std::vector<int32_t> mark_, gain_, lock_;
  std::vector<int32_t> queue_, moves_, d0_, d1_, order_;
  std::vector<int32_t> fm_[2][9];
  int32_t stamp_ = 0, tag_ = 0;
};

void recurse(Bisector &bis, std::vector<int32_t> &part, std::vector<int32_t> &elems, int32_t k, int32_t tag) {
  if (k <= 1 || elems.empty()) return;
  const int32_t kl = k / 2;
  const int64_t target_left = (static_cast<int64_t>(elems.size()) * kl + k / 2) / k;
  bis.run(elems, tag, target_left);
  std::vector<int32_t> left, right;
  left.reserve(static_cast<size_t>(target_left) + 16);
  right.reserve(elems.size() - static_cast<size_t>(target_left) + 16);
  for (int32_t e : elems) (bis.side(e) == 0 ? left : right).push_back(e);
  std::vector<int32_t>().swap(elems);
  for (int32_t e : right) part[e] = tag + kl;
  recurse(bis, part, left, kl, tag);
  recurse(bis, part, right, k - kl, tag + kl);
}

}  // namespace

bool partition_kway(int32_t n_parts, int32_t n_elems, int32_t n_nodes, const int32_t *tets, std::vector<int32_t> &epart,
                    PartitionStats &st, std::string &err) {
  if (n_parts < 1 || n_elems < 0 || n_nodes <= 0 || (n_elems > 0 && !tets)) {
    err = "partition_kway: bad argument";
    return false;
  }
  for (int64_t i = 0; i < 4 * static_cast<int64_t>(n_elems); ++i)
    if (tets[i] < 0 || tets[i] >= n_nodes) {
      err = "partition_kway: element " + std::to_string(i / 4) + " references a node outside [0, n_nodes)";
      return false;
    }
  epart.assign(n_elems, 0);
  st = PartitionStats();
  if (n_elems == 0) return true;
  if (n_parts > n_elems) {  // a rank without elements fails much later (plan build) with an unrelated message
    err = "partition_kway: " + std::to_string(n_parts) + " parts asked of " + std::to_string(n_elems) + " elements";
    return false;
  }
  Dual g;
  build_dual(n_elems, tets, g);
  if (n_parts > 1) {
    Bisector bis(g, epart);
    std::vector<int32_t> all(n_elems);
    std::iota(all.begin(), all.end(), 0);
    recurse(bis, epart, all, n_parts, 0);
  }
  // statistics
  std::vector<int64_t> size(n_parts, 0);
  for (int32_t e = 0; e < n_elems; ++e) {
    ++size[epart[e]];
    for (int k = 0; k < 4; ++k) {
      const int32_t f = g.nb(e)[k];
      st.face_cut += (f > e && epart[f] != epart[e]);
    }
  }
  st.min_part = *std::min_element(size.begin(), size.end());
  st.max_part = *std::max_element(size.begin(), size.end());
  if (st.min_part == 0) {  // (a bisection of a very small or disconnected piece left one side empty)
    err = "partition_kway: a part came out empty (" + std::to_string(n_elems) + " elements in " + std::to_string(n_parts) +
          " parts)";
    return false;
  }
  std::vector<int32_t> first(n_nodes, -1);
  std::vector<uint8_t> shared(n_nodes, 0);
  for (int32_t e = 0; e < n_elems; ++e)
    for (int a = 0; a < 4; ++a) {
      const int32_t v = tets[4 * static_cast<size_t>(e) + a];
      if (first[v] < 0) first[v] = epart[e];
      else if (first[v] != epart[e]) shared[v] = 1;
    }
  for (int32_t v = 0; v < n_nodes; ++v) st.interface_nodes += shared[v];
  return true;
}

}  // namespace saa
