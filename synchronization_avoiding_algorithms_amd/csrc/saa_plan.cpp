// Host-side construction of the owner-computes block plan (see saa_plan.h).  Pure C++, no HIP.
#include "saa_plan.h"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace saa {
namespace {

struct Rcb {
  const double *xyz;
  std::vector<int32_t> &order;          // node ids being permuted in place
  std::vector<int32_t> &block_start;    // filled leaf by leaf, in order
  void split(int64_t lo, int64_t hi, int32_t nblk) {
    if (nblk <= 1) {
      std::sort(order.begin() + lo, order.begin() + hi);
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
    const double *c = xyz;
    std::nth_element(order.begin() + lo, order.begin() + lo + k, order.begin() + hi,
                     [c, ax](int32_t a, int32_t b) {
                       const double va = c[3 * static_cast<int64_t>(a) + ax];
                       const double vb = c[3 * static_cast<int64_t>(b) + ax];
                       return va < vb || (va == vb && a < b);
                     });
    split(lo, lo + k, left_blk);
    split(lo + k, hi, nblk - left_blk);
  }
};

// Re-order the element copies of one block for the LDS atomics of the element phase.
// A ds_add_f64 wave-instruction is executed per (vertex slot, component) over two 32-lane halves;
// lanes of a half whose accumulators share an LDS bank pair (8-byte words: (3*node+c) mod 32, i.e.
// node mod 32 for a fixed component) are serialised, lanes on the SAME address even more so
// (measured on gfx950, tools/lds_microbench.hip: 7 cycles per wave-instruction conflict-free, 22 for
// random nodes, 60 when 6 lanes hit one address - the natural order of the 6 tets around a cube diagonal).
// Greedy first-fit: every aligned run of 32 element slots (= one half-wave of one sweep) keeps, per
// vertex slot, a 32-bit mask of the banks already taken by OWNED nodes; an element goes to the first
// half where all its owned vertices find their bank free, else to the half with the fewest clashes.
void reorder_for_atomics(std::vector<uint16_t> &conn, int64_t off, int32_t n_elem, int32_t n_owned,
                         std::vector<uint16_t> &scratch) {
  constexpr int kHalf = 32;
  if (n_elem <= 1) return;
  const int32_t n_halves = (n_elem + kHalf - 1) / kHalf;
  const int32_t last_cap = n_elem - (n_halves - 1) * kHalf;
  std::vector<int32_t> fill(n_halves, 0);
  std::vector<uint32_t> taken(static_cast<size_t>(n_halves) * 4, 0);
  std::vector<int32_t> place(n_elem);
  auto cap = [&](int32_t h) { return h == n_halves - 1 ? last_cap : kHalf; };
  auto clashes = [&](int32_t h, const uint16_t *c) {
    int k = 0;
    for (int a = 0; a < 4; ++a)
      if (c[a] < n_owned && (taken[static_cast<size_t>(h) * 4 + a] >> (c[a] & 31) & 1u)) ++k;
    return k;
  };
  int32_t first_open = 0, cursor = 0;
  for (int32_t e = 0; e < n_elem; ++e) {
    const uint16_t *c = &conn[4 * static_cast<size_t>(off + e)];
    while (first_open < n_halves && fill[first_open] >= cap(first_open)) ++first_open;
    const int32_t span = n_halves - first_open;
    int32_t best = -1, best_k = 5;
    // rotating start: neighbouring elements (which share nodes) land in different halves
    for (int32_t t = 0; t < span; ++t) {
      const int32_t h = first_open + (cursor + t) % span;
      if (fill[h] >= cap(h)) continue;
      const int k = clashes(h, c);
      if (k < best_k) {
        best_k = k;
        best = h;
        if (k == 0) break;
      }
    }
    ++cursor;
    place[e] = best * kHalf + fill[best]++;
    for (int a = 0; a < 4; ++a)
      if (c[a] < n_owned) taken[static_cast<size_t>(best) * 4 + a] |= 1u << (c[a] & 31);
  }
  scratch.resize(4 * static_cast<size_t>(n_elem));
  for (int32_t e = 0; e < n_elem; ++e)
    for (int a = 0; a < 4; ++a)
      scratch[4 * static_cast<size_t>(place[e]) + a] = conn[4 * static_cast<size_t>(off + e) + a];
  std::copy(scratch.begin(), scratch.end(), conn.begin() + 4 * off);
}

int32_t choose_block_count(int32_t n_nodes, int32_t block_nodes) {
  int64_t nb = (static_cast<int64_t>(n_nodes) + block_nodes - 1) / block_nodes;
  // MI355X has 256 CUs: once there is more than about a chip-full of blocks, make the count a
  // multiple of 256 so that every CU gets the same number of (equal-sized) blocks.
  if (nb > 192) nb = (nb + 255) / 256 * 256;
  nb = std::max<int64_t>(1, std::min<int64_t>(nb, n_nodes));
  return static_cast<int32_t>(nb);
}

bool build_once(int32_t n_nodes, int32_t n_elems, const double *xyz, const int32_t *tets,
                int32_t block_nodes, Plan &plan, std::string &err, bool &too_big) {
  too_big = false;
  plan = Plan();
  plan.n_nodes = n_nodes;
  plan.n_elems = n_elems;
  const int32_t nb = choose_block_count(n_nodes, block_nodes);

  plan.new_to_old.resize(n_nodes);
  std::iota(plan.new_to_old.begin(), plan.new_to_old.end(), 0);
  std::vector<int32_t> block_start;
  block_start.reserve(nb + 1);
  Rcb rcb{xyz, plan.new_to_old, block_start};
  rcb.split(0, n_nodes, nb);
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

  plan.blocks.resize(n_blocks);
  plan.conn.resize(4 * static_cast<size_t>(plan.n_elem_copies));
  std::vector<int32_t> tmp;
  std::vector<uint16_t> reorder_scratch;
  std::vector<char> interior_flag;
  for (int32_t b = 0; b < n_blocks; ++b) {
    BlockDesc &d = plan.blocks[b];
    d.node_start = block_start[b];
    d.n_owned = block_start[b + 1] - block_start[b];
    d.elem_off = static_cast<int32_t>(off[b]);
    d.n_elem = static_cast<int32_t>(off[b + 1] - off[b]);
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
        int32_t loc;
        if (g >= lo && g < hi)
          loc = g - lo;
        else
          loc = d.n_owned + static_cast<int32_t>(std::lower_bound(tmp.begin(), tmp.end(), g) - tmp.begin());
        plan.conn[4 * static_cast<size_t>(c) + a] = static_cast<uint16_t>(loc);
      }
    // interior elements first (they can run before the halo records have arrived), then each part
    // gets its own conflict-avoiding order
    {
      uint16_t *cb = &plan.conn[4 * static_cast<size_t>(off[b])];
      reorder_scratch.assign(cb, cb + 4 * static_cast<size_t>(d.n_elem));
      int32_t lo_i = 0, hi_i = d.n_elem;
      for (int32_t e = 0; e < d.n_elem; ++e) {
        const uint16_t *c = &reorder_scratch[4 * static_cast<size_t>(e)];
        const bool interior = c[0] < d.n_owned && c[1] < d.n_owned && c[2] < d.n_owned && c[3] < d.n_owned;
        interior_flag.push_back(interior);
        if (interior) ++lo_i;
      }
      d.n_interior = lo_i;
      int32_t wi = 0, wb = lo_i;
      for (int32_t e = 0; e < d.n_elem; ++e) {
        const int32_t dst = interior_flag[e] ? wi++ : wb++;
        std::copy(&reorder_scratch[4 * static_cast<size_t>(e)], &reorder_scratch[4 * static_cast<size_t>(e)] + 4,
                  cb + 4 * static_cast<size_t>(dst));
      }
      (void)hi_i;
      interior_flag.clear();
    }
    d.pad_ = 0;
    reorder_for_atomics(plan.conn, off[b], d.n_interior, d.n_owned, reorder_scratch);
    reorder_for_atomics(plan.conn, off[b] + d.n_interior, d.n_elem - d.n_interior, d.n_owned, reorder_scratch);
    plan.halo_ids.insert(plan.halo_ids.end(), tmp.begin(), tmp.end());
    plan.max_owned = std::max(plan.max_owned, d.n_owned);
    plan.max_local = std::max(plan.max_local, d.n_owned + d.n_halo);
    plan.n_halo_total += d.n_halo;
  }
  return true;
}

}  // namespace

bool build_plan(int32_t n_nodes, int32_t n_elems, const double *xyz, const int32_t *tets,
                int32_t block_nodes, Plan &plan, std::string &err) {
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
  int32_t bn = block_nodes > 0 ? block_nodes : kDefaultBlockNodes;
  bn = std::min(bn, kMaxLocalNodes);
  while (true) {
    bool too_big = false;
    if (build_once(n_nodes, n_elems, xyz, tets, bn, plan, err, too_big)) return true;
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
