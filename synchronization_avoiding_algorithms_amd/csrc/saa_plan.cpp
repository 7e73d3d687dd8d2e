// Host-side construction of the owner-computes block plan (see saa_plan.h).  Pure C++, no HIP.
#include "saa_plan.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
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

// Re-order (and re-orient) the element copies of one block for the LDS traffic of the element phase.
// Every per-node quantity lives in LDS bank pair (slot mod 32) (saa_plan.h, LDS image), and an LDS
// wave-instruction (ds_read_b64 of one plane, ds_add_f64 of one force component) is executed per
// vertex slot over two 32-lane halves; lanes of a half that share a bank pair are serialised, lanes
// adding to the SAME address even more so (gfx950, tools/lds_microbench.hip: ds_add_f64 costs 7
// cycles conflict-free, 22 for random nodes, 60 when 6 lanes hit one address - the natural order of
// the 6 tets around a cube diagonal).
// Packing, one half-wave (32 element slots) at a time: scan the not yet placed elements and take
// those for which one of the 12 EVEN vertex permutations (orientation, hence signed detJ, is
// preserved; the nodal forces follow their vertices) puts all four vertices on bank pairs still
// free in that half.  If the scan window runs dry the cheapest clash is taken (a clash on an owned
// vertex costs a read and an atomic, on a halo vertex only a read).
// Returns through mult_sum / mult_cnt the worst bank multiplicity summed over (half, slot).
double g_atomic_mult_sum = 0.0;  // debug statistic (SAA_PLAN_DEBUG)
constexpr int kEvenPerms[12][4] = {{0, 1, 2, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {1, 0, 3, 2}, {1, 2, 0, 3}, {1, 3, 2, 0},
                                   {2, 0, 1, 3}, {2, 1, 3, 0}, {2, 3, 0, 1}, {3, 0, 2, 1}, {3, 1, 0, 2}, {3, 2, 1, 0}};

void reorder_for_lds(std::vector<uint16_t> &conn, int64_t off, int32_t n_elem, int32_t owned_limit,
                     std::vector<uint16_t> &scratch, double &mult_sum, int64_t &mult_cnt) {
  constexpr int kHalf = 32;
  constexpr int kWindow = 768;  // pool elements examined per half before clashes are accepted
  if (n_elem <= 0) return;
  const int32_t n_halves = (n_elem + kHalf - 1) / kHalf;
  std::vector<int32_t> pool(n_elem);
  for (int32_t e = 0; e < n_elem; ++e) pool[e] = e;
  scratch.resize(4 * static_cast<size_t>(n_elem));
  const uint16_t *src = &conn[4 * static_cast<size_t>(off)];
  int32_t out = 0;
  for (int32_t h = 0; h < n_halves; ++h) {
    const int32_t cap = std::min<int32_t>(kHalf, n_elem - out);
    uint32_t taken[4] = {0, 0, 0, 0};
    int32_t placed = 0;
    auto put = [&](int32_t pool_pos, int perm) {
      const uint16_t *c = src + 4 * static_cast<size_t>(pool[pool_pos]);
      for (int a = 0; a < 4; ++a) {
        const uint16_t v = c[kEvenPerms[perm][a]];
        scratch[4 * static_cast<size_t>(out) + a] = v;
        taken[a] |= 1u << (v & 31);
      }
      ++out;
      ++placed;
      pool[pool_pos] = -1;
    };
    // pass 1: clash-free placements
    const int32_t lim = std::min<int32_t>(static_cast<int32_t>(pool.size()), kWindow);
    for (int32_t p = 0; p < lim && placed < cap; ++p) {
      const uint16_t *c = src + 4 * static_cast<size_t>(pool[p]);
      const uint32_t r[4] = {c[0] & 31u, c[1] & 31u, c[2] & 31u, c[3] & 31u};
      for (int q = 0; q < 12; ++q) {
        const int *pm = kEvenPerms[q];
        if (((taken[0] >> r[pm[0]]) | (taken[1] >> r[pm[1]]) | (taken[2] >> r[pm[2]]) | (taken[3] >> r[pm[3]])) & 1u)
          continue;
        put(p, q);
        break;
      }
    }
    // pass 2: fill the rest of the half where it hurts least.  An LDS instruction costs its WORST bank
    // multiplicity, so once a slot has one doubled bank further doublings on other banks of that slot are
    // free.  Reads (6 x ds_read_b64, ~2 cycles per level) see every vertex, atomics (3 x ds_add_f64, ~7
    // cycles per level) only owned ones.
    if (placed < cap) {
      uint8_t cnt_all[4][32] = {}, cnt_own[4][32] = {};
      int max_all[4] = {0, 0, 0, 0}, max_own[4] = {0, 0, 0, 0};
      for (int32_t l = 0; l < placed; ++l)
        for (int a = 0; a < 4; ++a) {
          const uint16_t v = scratch[4 * static_cast<size_t>(out - placed + l) + a];
          max_all[a] = std::max<int>(max_all[a], ++cnt_all[a][v & 31]);
          if (v < owned_limit) max_own[a] = std::max<int>(max_own[a], ++cnt_own[a][v & 31]);
        }
      while (placed < cap) {
        int32_t best_p = -1, best_q = 0, best_k = 1 << 30;
        for (int32_t p = 0; p < lim && best_k > 0; ++p) {
          if (pool[p] < 0) continue;
          const uint16_t *c = src + 4 * static_cast<size_t>(pool[p]);
          for (int q = 0; q < 12; ++q) {
            int k = 0;
            for (int a = 0; a < 4; ++a) {
              const uint16_t v = c[kEvenPerms[q][a]];
              if (cnt_all[a][v & 31] + 1 > max_all[a]) k += 12;
              if (v < owned_limit && cnt_own[a][v & 31] + 1 > max_own[a]) k += 21;
            }
            if (k < best_k) {
              best_k = k;
              best_p = p;
              best_q = q;
              if (k == 0) break;
            }
          }
        }
        if (getenv("SAA_PLAN_DEBUG3")) fprintf(stderr, "  pass2 lane %d cost %d (max %d %d %d %d)\n", placed, best_k, max_all[0], max_all[1], max_all[2], max_all[3]);
        const uint16_t *c = src + 4 * static_cast<size_t>(pool[best_p]);
        for (int a = 0; a < 4; ++a) {
          const uint16_t v = c[kEvenPerms[best_q][a]];
          max_all[a] = std::max<int>(max_all[a], ++cnt_all[a][v & 31]);
          if (v < owned_limit) max_own[a] = std::max<int>(max_own[a], ++cnt_own[a][v & 31]);
        }
        put(best_p, best_q);
      }
    }
    pool.erase(std::remove(pool.begin(), pool.begin() + lim, -1), pool.begin() + lim);
    // quality of this half: worst multiplicity seen by the reads (all lanes) and by the atomics (owned lanes)
    for (int a = 0; a < 4; ++a) {
      int cnt[32] = {0}, cnt_o[32] = {0};
      int worst = 0, worst_o = 0;
      for (int32_t l = 0; l < cap; ++l) {
        const uint16_t v = scratch[4 * static_cast<size_t>(out - cap + l) + a];
        worst = std::max(worst, ++cnt[v & 31]);
        if (v < owned_limit) worst_o = std::max(worst_o, ++cnt_o[v & 31]);
      }
      mult_sum += worst;
      g_atomic_mult_sum += worst_o;
      ++mult_cnt;
      if (getenv("SAA_PLAN_DEBUG2")) fprintf(stderr, "%s%d/%d", a ? " " : "half: ", worst, worst_o);
    }
    if (getenv("SAA_PLAN_DEBUG2")) fprintf(stderr, "  (cap %d, pool left %zu)\n", cap, pool.size());
  }
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
  double mult_sum = 0.0;
  int64_t mult_cnt = 0;
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
    d.owned_limit = lds_index(d.n_owned);
    // connectivity becomes LDS slots (tile/plane layout of saa_plan.h); slot mod 32 = bank pair
    for (int64_t c = 4 * off[b]; c < 4 * off[b + 1]; ++c)
      plan.conn[c] = static_cast<uint16_t>(lds_index(plan.conn[c]));
    reorder_for_lds(plan.conn, off[b], d.n_interior, d.owned_limit, reorder_scratch, mult_sum, mult_cnt);
    reorder_for_lds(plan.conn, off[b] + d.n_interior, d.n_elem - d.n_interior, d.owned_limit, reorder_scratch,
                    mult_sum, mult_cnt);
    plan.halo_ids.insert(plan.halo_ids.end(), tmp.begin(), tmp.end());
    plan.max_owned = std::max(plan.max_owned, d.n_owned);
    plan.max_local = std::max(plan.max_local, d.n_owned + d.n_halo);
    plan.n_halo_total += d.n_halo;
  }
  plan.lds_conflict_factor = mult_cnt ? mult_sum / mult_cnt : 1.0;
  if (getenv("SAA_PLAN_DEBUG"))
    fprintf(stderr, "plan: read conflict factor %.3f, atomic conflict factor %.3f\n", plan.lds_conflict_factor,
            mult_cnt ? g_atomic_mult_sum / mult_cnt : 1.0);
  g_atomic_mult_sum = 0.0;
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
