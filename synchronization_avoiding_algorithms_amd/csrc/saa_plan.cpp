// Host-side construction of the owner-computes block plan (see saa_plan.h).  Pure C++, no HIP.
#include "saa_plan.h"

#include <algorithm>
#include <atomic>
#include <thread>
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

// Re-order (and re-orient) the element copies of one block for the LDS traffic of the element phase
// (LDS image and bank rules: saa_plan.h).  Measured on gfx950 (tools/lds_microbench.hip): ds_add_f64 costs
// 7 cycles per wave-instruction conflict-free, 22 for random nodes, 60 when 6 lanes hit one address (the
// natural order of the 6 tets around a cube diagonal); ds_read_b128 6 conflict-free, 11 random.
// Packing, one half-wave (32 element slots) at a time: scan the not yet placed elements and take those for
// which one of the 12 EVEN vertex permutations (orientation, hence signed detJ, is preserved; the nodal
// forces follow their vertices) and one of the half's two ds_read_b128 lane groups leaves every vertex on
// a free bank: node mod 16 free in the group (reads), node mod 32 free in the half for owned vertices
// (atomics).  When the scan window runs dry the placement that raises the worst multiplicities least is
// taken.  mult_sum / mult_cnt accumulate the worst read multiplicity over (lane group, vertex slot).
constexpr int kEvenPerms[12][4] = {{0, 1, 2, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {1, 0, 3, 2}, {1, 2, 0, 3}, {1, 3, 2, 0},
                                   {2, 0, 1, 3}, {2, 1, 3, 0}, {2, 3, 0, 1}, {3, 0, 2, 1}, {3, 1, 0, 2}, {3, 2, 1, 0}};
// ds_read_b128 lane groups of a 32-lane half (MI355X_MICROARCH.md, LDS): {0-3,12-15,20-27} and {4-11,16-19,28-31}
constexpr int kGroupLanes[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};

struct PackStats {
  double read_mult = 0.0, atomic_mult = 0.0;  // sums of worst multiplicities
  int64_t read_cnt = 0, atomic_cnt = 0;
};

void reorder_for_lds(std::vector<uint16_t> &conn, int64_t off, int32_t n_elem, int32_t n_owned,
                     std::vector<uint16_t> &scratch, PackStats &st) {
  constexpr int kHalf = 32;
  constexpr int kWindow = 768;  // pool elements examined per half before clashes are accepted
  if (n_elem <= 0) return;
  const int32_t n_halves = (n_elem + kHalf - 1) / kHalf;
  std::vector<int32_t> pool(n_elem);
  for (int32_t e = 0; e < n_elem; ++e) pool[e] = e;
  scratch.assign(4 * static_cast<size_t>(n_elem), 0);
  const uint16_t *src = &conn[4 * static_cast<size_t>(off)];
  int32_t done = 0;
  for (int32_t h = 0; h < n_halves; ++h) {
    const int32_t cap = std::min<int32_t>(kHalf, n_elem - done);
    // lanes available in this half: lane l exists if l < cap
    int32_t free_lane[2][16], n_free[2] = {0, 0};
    for (int g = 0; g < 2; ++g)
      for (int j = 0; j < 16; ++j)
        if (kGroupLanes[g][j] < cap) free_lane[g][n_free[g]++] = kGroupLanes[g][j];
    int used[2] = {0, 0};
    uint8_t cnt_rd[2][4][16] = {}, cnt_at[4][32] = {};
    int max_rd[2][4] = {}, max_at[4] = {};
    uint32_t taken_rd[2][4] = {}, taken_at[4] = {};
    int32_t placed = 0;
    auto put = [&](int32_t pool_pos, int perm, int g) {
      const uint16_t *c = src + 4 * static_cast<size_t>(pool[pool_pos]);
      const int32_t lane = free_lane[g][used[g]++];
      for (int a = 0; a < 4; ++a) {
        const uint16_t v = c[kEvenPerms[perm][a]];
        scratch[4 * static_cast<size_t>(done + lane) + a] = v;
        taken_rd[g][a] |= 1u << (v & 15);
        max_rd[g][a] = std::max<int>(max_rd[g][a], ++cnt_rd[g][a][v & 15]);
        if (v < n_owned) {
          taken_at[a] |= 1u << (v & 31);
          max_at[a] = std::max<int>(max_at[a], ++cnt_at[a][v & 31]);
        }
      }
      ++placed;
      pool[pool_pos] = -1;
    };
    const int32_t lim = std::min<int32_t>(static_cast<int32_t>(pool.size()), kWindow);
    // pass 1: clash-free placements
    for (int32_t p = 0; p < lim && placed < cap; ++p) {
      const uint16_t *c = src + 4 * static_cast<size_t>(pool[p]);
      bool ok = false;
      for (int q = 0; q < 12 && !ok; ++q) {
        const int *pm = kEvenPerms[q];
        uint32_t at = 0;
        for (int a = 0; a < 4; ++a)
          if (c[pm[a]] < n_owned) at |= taken_at[a] >> (c[pm[a]] & 31);
        if (at & 1u) continue;
        for (int g = 0; g < 2 && !ok; ++g) {
          if (used[g] >= n_free[g]) continue;
          if (((taken_rd[g][0] >> (c[pm[0]] & 15)) | (taken_rd[g][1] >> (c[pm[1]] & 15)) |
               (taken_rd[g][2] >> (c[pm[2]] & 15)) | (taken_rd[g][3] >> (c[pm[3]] & 15))) & 1u)
            continue;
          put(p, q, g);
          ok = true;
        }
      }
    }
    // pass 2: fill the rest where it raises the worst multiplicities least (an LDS instruction costs its
    // WORST bank multiplicity; reads 3 x ds_read_b128 per vertex ~4 cycles a level, atomics 3 x ds_add_f64 ~7)
    while (placed < cap) {
      int32_t best_p = -1, best_q = 0, best_g = 0, best_k = 1 << 30;
      for (int32_t p = 0; p < lim && best_k > 0; ++p) {
        if (pool[p] < 0) continue;
        const uint16_t *c = src + 4 * static_cast<size_t>(pool[p]);
        for (int q = 0; q < 12 && best_k > 0; ++q) {
          int k_at = 0;
          for (int a = 0; a < 4; ++a) {
            const uint16_t v = c[kEvenPerms[q][a]];
            if (v < n_owned && cnt_at[a][v & 31] + 1 > max_at[a]) k_at += 21;
          }
          for (int g = 0; g < 2; ++g) {
            if (used[g] >= n_free[g]) continue;
            int k = k_at;
            for (int a = 0; a < 4; ++a) {
              const uint16_t v = c[kEvenPerms[q][a]];
              if (cnt_rd[g][a][v & 15] + 1 > max_rd[g][a]) k += 12;
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
      put(best_p, best_q, best_g);
    }
    pool.erase(std::remove(pool.begin(), pool.begin() + lim, -1), pool.begin() + lim);
    for (int a = 0; a < 4; ++a) {
      for (int g = 0; g < 2; ++g)
        if (n_free[g] > 0) {
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

// Automatic block size.  Up to ~190k nodes (one MI355X-sized partition of ~1M tets) every CU gets exactly
// ONE block: fewer, larger blocks duplicate fewer border elements (1.25x at 744 owned nodes against
// 1.37x at 372) and all 256 run as one wave of workgroups of 1024 threads.  Larger partitions use
// 384-node blocks in several rounds per CU, which overlap each other's memory and LDS phases
// (measured: 8.2M tets 121 us/step with 377-node blocks against 142 us with 707-node blocks).
int32_t auto_block_nodes(int32_t n_nodes) {
  constexpr int32_t kCUs = 256, kBig = 760;
  if (n_nodes <= kDefaultBlockNodes) return n_nodes;  // tiny mesh: one block
  if (n_nodes <= kCUs * kBig) return std::max<int32_t>(96, (n_nodes + kCUs - 1) / kCUs);
  return kDefaultBlockNodes;
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
    plan.halo_ids.insert(plan.halo_ids.end(), tmp.begin(), tmp.end());
    plan.max_owned = std::max(plan.max_owned, d.n_owned);
    plan.max_local = std::max(plan.max_local, d.n_owned + d.n_halo);
    plan.n_halo_total += d.n_halo;
  }
  // LDS packing of every block's interior and boundary element lists: independent per block -> threads
  {
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const unsigned n_thr = static_cast<unsigned>(std::min<int64_t>(hw, std::max<int32_t>(1, n_blocks / 8)));
    std::vector<PackStats> stats(n_thr);
    std::atomic<int32_t> next{0};
    auto work = [&](unsigned t) {
      std::vector<uint16_t> scratch;
      for (int32_t b = next.fetch_add(1); b < n_blocks; b = next.fetch_add(1)) {
        const BlockDesc &d = plan.blocks[b];
        reorder_for_lds(plan.conn, off[b], d.n_interior, d.n_owned, scratch, stats[t]);
        reorder_for_lds(plan.conn, off[b] + d.n_interior, d.n_elem - d.n_interior, d.n_owned, scratch, stats[t]);
      }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n_thr; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    PackStats tot;
    for (const auto &st : stats) {
      tot.read_mult += st.read_mult;
      tot.read_cnt += st.read_cnt;
      tot.atomic_mult += st.atomic_mult;
      tot.atomic_cnt += st.atomic_cnt;
    }
    plan.lds_conflict_factor = tot.read_cnt ? tot.read_mult / tot.read_cnt : 1.0;
    if (getenv("SAA_PLAN_DEBUG"))
      fprintf(stderr, "plan: read conflict factor %.3f, atomic conflict factor %.3f (%u threads)\n",
              plan.lds_conflict_factor, tot.atomic_cnt ? tot.atomic_mult / tot.atomic_cnt : 1.0, n_thr);
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
  int32_t bn = block_nodes > 0 ? block_nodes : auto_block_nodes(n_nodes);
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
