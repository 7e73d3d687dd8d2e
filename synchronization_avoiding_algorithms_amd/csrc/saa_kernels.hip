// gfx950 (MI355X / CDNA4) kernels of the explicit linear-tet elastodynamics step.
//
// One fused kernel per time step (fused_step_kernel): a workgroup owns a contiguous block of nodes,
//   1. stages coordinates + displacement d^n of its owned and halo nodes in LDS (48-byte records); halo
//      gathers and the update operands travel in registers while the interior elements already run,
//   2. evaluates every element touching an owned node, matrix-free, two face-adjacent tets per lane
//      ("items", saa_plan.h: 5 node records and 5 force flushes for two elements):
//          f_a = (detJ/6) * sigma(grad u) * gradN_a        [= K_e d restricted to node a]
//      (closed form of Local_K_coronary, /root/reference Tools/Mat_construction.py:79-119, with the
//      B-matrix convention of :99-104; the 4-point rule of Tools/Qudrature.py:7-12 is exact for the
//      constant integrand, weights sum to 1/6) and accumulates f_a of OWNED nodes in LDS
//      (ds_add_f64) - no global atomics, no force vector in HBM,
//   3. applies the damped central-difference update of Tools/Dynamic_solver.py:13-20 to its owned
//      dofs in the reference's association order (no FMA contraction) and writes d^(n+1); shared nodes
//      additionally publish their partial force (synchronised mode; with the peer exchange they are pushed to
//      the neighbour ranks' memory and summed right here, PEER variant) or take the LSTM prediction and
//      record it as history (sync-avoiding mode, Online_predictor.py:298-301).
// The same kernel in FORCE_ONLY mode writes f_int instead (backs LocalK.dot, Dynamic_solver.py:12).
//
// Bandwidth-bound gather/scatter in fp64: MFMA is not used (and fp64 MFMA has the vector rate on
// gfx950 anyway).  64-wide waves; cross-workgroup data flows through kernel boundaries (cross-RANK data of the
// PEER variant through self-validating entries in fine-grained memory).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <stdint.h>

#include "saa_device.h"

#ifndef SAA_LB
#define SAA_LB 1024
#endif

namespace saa {

// ---------------------------------------------------------------------------------------------
// The update expression.  Must stay bit-identical to NumPy evaluating
//   (dt**2*(F_ext - F_int) + 2*l_M*d0 - l_M*dn + dt/2*l_M*alpha*dn)/(l_M + 0.5*alpha*l_M*dt)
// (Dynamic_solver.py:17): same association, every product and sum rounded separately.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double cd_update_dof(double f_int, double f_pre, double m, double d0, double dn,
                                                const StepConsts &k) {
#pragma clang fp contract(off)
  const double f_ext = f_pre * k.ramp;              // F_rankwise * linear_ramp(tn)   (:13)
  const double a = k.dt2 * (f_ext - f_int);         // dt**2*(F_ext - F_int)
  const double b = (2.0 * m) * d0;                  // 2*l_M*d0
  const double c = m * dn;                          // l_M*dn
  const double d = ((k.half_dt * m) * k.alpha) * dn;  // dt/2*l_M*alpha*dn
  const double num = ((a + b) - c) + d;
  const double den = m + (k.half_alpha * m) * k.dt;  // l_M + 0.5*alpha*l_M*dt
  return num / den;
}

// Internal force of one linear tet on its nodes 1..3 (node 0 gets minus their sum).
// x*, u*: coordinates / displacements of the 4 nodes.  lam6, mu6: Lame parameters divided by 6.
struct Vec3 {
  double x, y, z;
};
__device__ __forceinline__ Vec3 sub(const Vec3 &a, const Vec3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ Vec3 cross(const Vec3 &a, const Vec3 &b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// Core of tet_forces on the edge vectors e_a = x_a - x_0, the displacement differences w_a = u_a - u_0 and the first
// cofactor row c1 = e2 x e3, which the two tets of a work item share up to its sign (see item_forces).
// lam6, mu6: the Lame parameters divided by 6 (DeviceMesh: the host divides once).
__device__ __forceinline__ void tet_core(const Vec3 &e1, const Vec3 &e2, const Vec3 &e3, const Vec3 &c1, const Vec3 &w1,
                                         const Vec3 &w2, const Vec3 &w3, double lam6, double mu6, Vec3 &f1, Vec3 &f2, Vec3 &f3) {
  // J columns are the edges x_a - x_0 (Shape_function_Deriv.py:60-67); gradN_a = c_a / detJ with
  // c_1 = e2 x e3, c_2 = e3 x e1, c_3 = e1 x e2 (rows of adj J), detJ = e1 . c_1 (signed, :93).
  const Vec3 c2 = cross(e3, e1), c3 = cross(e1, e2);
  const double det = e1.x * c1.x + e1.y * c1.y + e1.z * c1.z;
  // r = 1/detJ: v_rcp_f64 (~27 bits) and ONE Newton step - the error after it is the square of the seed's, 2^-54, i.e.
  // r is within an ulp or two, the same class as the rounding of every other operation here (a second step, used until
  // round 3, bought nothing measurable in any parity figure); the factor 1/6 of s = (detJ/6)/detJ^2 sits in lam6 / mu6.
  // Three instructions where the IEEE division sequence takes eleven, several of them quarter-rate.
  double r = __builtin_amdgcn_rcp(det);
  r = __builtin_fma(r, __builtin_fma(-det, r, 1.0), r);
  // H' = detJ * grad u = sum_a w_a (x) c_a; the off-diagonal entries are only needed as the sums h_ij + h_ji of the
  // symmetric strain, each accumulated as ONE chain of six products
  const double h00 = w1.x * c1.x + w2.x * c2.x + w3.x * c3.x;
  const double h11 = w1.y * c1.y + w2.y * c2.y + w3.y * c3.y;
  const double h22 = w1.z * c1.z + w2.z * c2.z + w3.z * c3.z;
  const double g01 = w1.x * c1.y + w2.x * c2.y + w3.x * c3.y + w1.y * c1.x + w2.y * c2.x + w3.y * c3.x;
  const double g02 = w1.x * c1.z + w2.x * c2.z + w3.x * c3.z + w1.z * c1.x + w2.z * c2.x + w3.z * c3.x;
  const double g12 = w1.y * c1.z + w2.y * c2.z + w3.y * c3.z + w1.z * c1.y + w2.z * c2.y + w3.z * c3.y;
  // sigma' * s with sigma = lam tr(eps) I + 2 mu eps (commons.py:25-31, Voigt xx,yy,zz,yz,xz,xy)
  const double ls = lam6 * r, ms = mu6 * r, ms2 = ms + ms;
  const double ltr = ls * (h00 + h11 + h22);
  const double sxx = ltr + ms2 * h00, syy = ltr + ms2 * h11, szz = ltr + ms2 * h22;
  const double syz = ms * g12, sxz = ms * g02, sxy = ms * g01;
  f1 = {sxx * c1.x + sxy * c1.y + sxz * c1.z, sxy * c1.x + syy * c1.y + syz * c1.z,
        sxz * c1.x + syz * c1.y + szz * c1.z};
  f2 = {sxx * c2.x + sxy * c2.y + sxz * c2.z, sxy * c2.x + syy * c2.y + syz * c2.z,
        sxz * c2.x + syz * c2.y + szz * c2.z};
  f3 = {sxx * c3.x + sxy * c3.y + sxz * c3.z, sxy * c3.x + syy * c3.y + syz * c3.z,
        sxz * c3.x + syz * c3.y + szz * c3.z};
}
__device__ __forceinline__ void tet_forces(const Vec3 &x0, const Vec3 &x1, const Vec3 &x2, const Vec3 &x3,
                                           const Vec3 &u0, const Vec3 &u1, const Vec3 &u2, const Vec3 &u3,
                                           double lam6, double mu6, Vec3 &f1, Vec3 &f2, Vec3 &f3) {
  const Vec3 e1 = sub(x1, x0), e2 = sub(x2, x0), e3 = sub(x3, x0);
  tet_core(e1, e2, e3, cross(e2, e3), sub(u1, u0), sub(u2, u0), sub(u3, u0), lam6, mu6, f1, f2, f3);
}
// The two tets of a work item, A = (p; a, r, q) and B = (p; b, q, r), evaluated from their common vertex p: the face
// edges r - p and q - p, the displacement differences along them and their cross product are computed once - B's first
// cofactor row (q - p) x (r - p) is A's (r - p) x (q - p) with the sign changed.
struct PairShared {
  Vec3 er, eq, wr, wq, c;  // r - p, q - p, u_r - u_p, u_q - u_p, (r - p) x (q - p)
};
__device__ __forceinline__ PairShared pair_shared(const Vec3 &xp, const Vec3 &xq, const Vec3 &xr, const Vec3 &up,
                                                  const Vec3 &uq, const Vec3 &ur) {
  const Vec3 er = sub(xr, xp), eq = sub(xq, xp);
  return {er, eq, sub(ur, up), sub(uq, up), cross(er, eq)};
}
__device__ __forceinline__ void tet_a_forces(const PairShared &g, const Vec3 &xp, const Vec3 &xa, const Vec3 &up,
                                             const Vec3 &ua, double lam6, double mu6, Vec3 &fa, Vec3 &fr, Vec3 &fq) {
  tet_core(sub(xa, xp), g.er, g.eq, g.c, sub(ua, up), g.wr, g.wq, lam6, mu6, fa, fr, fq);
}
__device__ __forceinline__ void tet_b_forces(const PairShared &g, const Vec3 &xp, const Vec3 &xb, const Vec3 &up,
                                             const Vec3 &ub, double lam6, double mu6, Vec3 &fb, Vec3 &fq, Vec3 &fr) {
  const Vec3 nc = {-g.c.x, -g.c.y, -g.c.z};
  tet_core(sub(xb, xp), g.eq, g.er, nc, sub(ub, up), g.wq, g.wr, lam6, mu6, fb, fq, fr);
}

typedef __attribute__((address_space(3))) double lds_double;

__device__ __forceinline__ void lds_add(double *p, double v) {
  // ds_add_f64 (no return): LDS-side fp64 add
  __builtin_amdgcn_ds_atomic_fadd_f64((lds_double *)p, v);
}

// blockIdx -> plan block.  Blocks are dealt round-robin to the 8 XCDs (blockIdx % 8 shares an XCD);
// plan blocks are spatially ordered (RCB leaves), so give every XCD one contiguous run of them and
// halo re-reads hit that XCD's L2.  Only a speed matter: any bijection is correct.
__device__ __forceinline__ int plan_block(int bid, int n_blocks) {
  const int per = n_blocks >> 3, rem = n_blocks & 7;
  const int x = bid & 7, j = bid >> 3;
  // XCD x owns per + (x < rem) blocks, starting after the blocks of XCDs < x
  return x * per + (x < rem ? x : rem) + j;
}

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain the
// global loads this kernel deliberately keeps in flight across its phases.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// One work item (saa_plan.h): two face-adjacent tets A = (a; p,q,r) and B = (b; p,r,q) - or a single tet
// when the flag is 0.  Five node records (three ds_read_b128 each) instead of eight, and the forces of the
// shared face p,q,r are summed in registers before ONE ds_add_f64 per owned node and component.
// Both tets are evaluated from p (even re-orderings (p; a,r,q) and (p; b,q,r)), so the face edges are shared.
struct Item {
  unsigned a, p, q, r, b;
  bool pair, null;
};
// packed item (saa_plan.h): 5 x 12-bit block-local node index, flag in bits 60-61
__device__ __forceinline__ Item unpack(const uint2 w) {
  const unsigned flag = w.y >> 28;
  return {w.x & 0xfffu, (w.x >> 12) & 0xfffu, (w.x >> 24) | ((w.y & 0xfu) << 8), (w.y >> 4) & 0xfffu,
          (w.y >> 16) & 0xfffu, flag == 1u, flag == 2u};
}
struct Rec {
  Vec3 x, u;
};
__device__ __forceinline__ Rec load_rec(const double *rec, unsigned n) {
  const double2 *r = reinterpret_cast<const double2 *>(rec + 6 * n);
  const double2 a = r[0], b = r[1], c = r[2];
  return {{a.x, a.y, b.x}, {b.y, c.x, c.y}};
}
// force accumulators: [n_owned][3] doubles (one address computation per node, components at immediate offsets)
// ALL_OWNED: the caller knows that every vertex of the item is owned (interior items): no test, no branch
template <bool ALL_OWNED = false>
__device__ __forceinline__ void flush(double *acc, unsigned n, int n_owned, const Vec3 &f) {
  if (ALL_OWNED || (int)n < n_owned) {
    double *a = acc + 3 * n;
    lds_add(a, f.x);
    lds_add(a + 1, f.y);
    lds_add(a + 2, f.z);
  }
}
__device__ __forceinline__ Vec3 add3(const Vec3 &a, const Vec3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ Vec3 neg_sum3(const Vec3 &a, const Vec3 &b, const Vec3 &c) {
  return {-(a.x + b.x + c.x), -(a.y + b.y + c.y), -(a.z + b.z + c.z)};
}

// In-kernel stamp (diagnostic build ABLATE == 8 only): shader clock after all outstanding LDS work has drained.
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};
// `mid` runs once per call, half-way through the item (after tet A): the resident kernel issues its halo loads there.
template <int ABLATE, typename Hook = NoHook, bool ALL_OWNED = false>
__device__ __forceinline__ void item_forces(const uint2 w, const double *rec, double *acc, int n_owned,
                                            double lam6, double mu6, int tid, double &sink, unsigned long long *T = nullptr,
                                            Hook mid = Hook()) {
  Item it = unpack(w);
  if (it.null) {  // idle lane left by the LDS packing (saa_plan.cpp)
    mid();
    return;
  }
  if (ABLATE == 2) {  // every lane reads its own fixed records: no index-dependent LDS traffic
    it.a = tid & 63; it.p = it.a + 1; it.q = it.a + 2; it.r = it.a + 3; it.b = it.a + 4;
  }
  Vec3 fa, fp, fq, fr, fb = {0, 0, 0};
  if (ABLATE == 6) {  // VALU only: operands from registers, results to a register sink
    const double t = 1.0 + 1e-3 * tid + sink;
    Vec3 g1, g2, g3;
    tet_forces({t, 0.1, 0.2}, {1.1 * t, 0.3, 0.1}, {0.2, t, 0.3}, {0.1, 0.2, 1.3 * t}, {t, t, 0}, {0, t, t}, {t, 0, t},
               {t, t, t}, lam6, mu6, fa, fr, fq);
    tet_forces({t, 0.1, 0.2}, {1.2 * t, 0.3, 0.1}, {0.1, 0.2, 1.3 * t}, {0.2, t, 0.3}, {t, t, 0}, {0, 2 * t, t}, {t, t, t},
               {t, 0, t}, lam6, mu6, g1, g2, g3);
    sink += fa.x + fa.y + fa.z + fr.x + fr.y + fr.z + fq.x + fq.y + fq.z + g1.x + g1.y + g1.z + g2.x + g2.y + g2.z +
            g3.x + g3.y + g3.z;
    mid();
    return;
  }
  unsigned long long t0 = 0, t1 = 0;
  if (ABLATE == 8) t0 = stamp();
  const Rec rp = load_rec(rec, it.p), rq = load_rec(rec, it.q), rr = load_rec(rec, it.r);
  PairShared pg;
  {
    const Rec ra = load_rec(rec, it.a);
    if (ABLATE == 8) {
      t1 = stamp();
      T[0] += t1 - t0;  // 12 reads: issue + arrival
    }
    if (ABLATE == 7) {  // LDS only: reads and atomics without the element arithmetic
      fa = add3(ra.x, ra.u); fp = add3(rp.x, rp.u); fq = add3(rq.x, rq.u); fr = add3(rr.x, rr.u);
    } else {
      // A as (p; a, r, q): forces on a, r, q; p gets minus their sum
      pg = pair_shared(rp.x, rq.x, rr.x, rp.u, rq.u, rr.u);
      tet_a_forces(pg, rp.x, ra.x, rp.u, ra.u, lam6, mu6, fa, fr, fq);
    }
  }
  if (ABLATE == 8) {
    t0 = stamp();
    T[1] += t0 - t1;  // VALU of tet A
  }
  if (ABLATE != 1) flush<ALL_OWNED>(acc, it.a, n_owned, fa);
  else sink += fa.x + fa.y + fa.z;
  mid();  // (the resident kernel's halo requests: before tet A 8.43, here 7.84, behind tet B 8.04 us per step - round 3)
  if (it.pair) {
    const Rec rb = load_rec(rec, it.b);
    if (ABLATE == 8) {
      t1 = stamp();
      T[2] += t1 - t0;  // 3 atomics + 3 reads round trip
    }
    if (ABLATE == 7) {
      fb = add3(rb.x, rb.u);
    } else {
      // B as (p; b, q, r)
      Vec3 gq, gr;
      tet_b_forces(pg, rp.x, rb.x, rp.u, rb.u, lam6, mu6, fb, gq, gr);
      fq = add3(fq, gq);
      fr = add3(fr, gr);
    }
    if (ABLATE == 8) {
      t0 = stamp();
      T[3] += t0 - t1;  // VALU of tet B
    }
    if (ABLATE != 1) flush<ALL_OWNED>(acc, it.b, n_owned, fb);
    else sink += fb.x + fb.y + fb.z;
  }
  // the forces of an element sum to zero, those of the pair too: p gets minus the sum of the other four (fb = 0 without B)
  if (ABLATE != 7) fp = neg_sum3(add3(fa, fb), fq, fr);
  if (ABLATE == 1) {
    sink += fp.x + fp.y + fp.z + fq.x + fq.y + fq.z + fr.x + fr.y + fr.z;
    return;
  }
  flush<ALL_OWNED>(acc, it.p, n_owned, fp);
  flush<ALL_OWNED>(acc, it.q, n_owned, fq);
  flush<ALL_OWNED>(acc, it.r, n_owned, fr);
  if (ABLATE == 8) T[4] += stamp() - t0;  // remaining atomics: issue + drain
}

// Per-thread prefetch depth (dofs): the update operands of the first kPreOwn*blockDim owned dofs and
// the records of the first kPreHalo*blockDim halo dofs travel in registers while elements are computed.
#ifndef SAA_KPRE_OWN
#define SAA_KPRE_OWN 2
#define SAA_KPRE_HALO 2
#define SAA_KPRE_CONN 2
#endif
constexpr int kPreOwn = SAA_KPRE_OWN;
constexpr int kPreHalo = SAA_KPRE_HALO;
// Connectivity of the first kPreConn interior sweeps is fetched BEFORE those loads: vector-memory
// results return in issue order, so a connectivity load issued later would make the first interior
// element wait for every prefetch in front of it.
constexpr int kPreConn = SAA_KPRE_CONN;

// ---------------------------------------------------------------------------------------------
// Direct peer exchange (saa_device.h: PeerMap).  Inboxes are double-buffered by sequence parity: a neighbour can be
// at most one step ahead, because its step seq+1 cannot end before it has received this rank's push of step seq+1,
// which is issued only after this rank has consumed step seq.
// ---------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void peer_store(PeerEntry *d, double f, unsigned seq) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(f);
  const u32x4 w = {(unsigned)b, seq, (unsigned)(b >> 32), seq};
  // one 16-byte store, system scope (write-through to the peer); each 8-byte half validates itself
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(d), "v"(w) : "memory");
}
// (the peer map is addressed through a pointer and its fields are read where they are used: a by-value copy costs
// ~30 scalar registers for the whole tail of the step, and the step kernels have none to spare)
__device__ __forceinline__ void peer_push(const PeerMap *pm, const PeerPushRec &r, int q, int c, double f, unsigned seq) {
  const int64_t par = seq & 1u;
  const int n_nb = (r.info >> 16) & 0xff;
  if (n_nb > 0) peer_store(r.dst0 + par * r.pstride0 + c, f, seq);
  if (n_nb > 1) {  // node held by three or more ranks
    const int e0 = pm->nb_off[q];
    for (int e = e0 + 1; e < e0 + n_nb; ++e) peer_store(pm->push_dst[e] + par * pm->push_pstride[e] + c, f, seq);
  }
}
// One neighbour's value of this step: polls - bounded - until both halves carry the step's sequence number.
__device__ __forceinline__ double peer_wait(const PeerMap *pm, const PeerEntry *src, unsigned seq) {
  const long long t0 = wall_clock64();
  unsigned long long lo, hi;
  while (true) {
    lo = __hip_atomic_load(&src->lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    hi = __hip_atomic_load(&src->hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((unsigned)(lo >> 32) == seq && (unsigned)(hi >> 32) == seq) break;
    // slow path only: a wait that already failed somewhere in this launch is not repeated for every later value
    int32_t *errp = pm->err;
    if (__hip_atomic_load(errp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
    if (wall_clock64() - t0 > pm->timeout_ticks) {  // a neighbour died or never attached: report, do not hang
      __hip_atomic_store(errp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  return __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull)));
}
// Force of shared node q, component c, summed over the holding ranks in RANK ORDER (the order of syn_cpus,
// Distributed_tools.py:84-86: identical bits on every rank).
__device__ __forceinline__ double peer_collect(const PeerMap *pm, const PeerRecvRec &r, int q, int c, double own,
                                               unsigned seq) {
  const int rank = pm->rank;
  const PeerEntry *in = pm->inbox + (int64_t)(seq & 1u) * pm->parity_stride + c;
  const unsigned long long others = r.holders & ~(1ull << rank);
  if (others == 0ull) return own;  // declared shared, held by this rank only
  if ((others & (others - 1ull)) == 0ull) return own + peer_wait(pm, in + r.recv0, seq);  // one other holder: a + b == b + a
  int e = pm->nb_off[q];
  const int world = pm->world;
  double f = 0.0;
  bool first = true;
  for (int p = 0; p < world; ++p) {
    if (!((r.holders >> p) & 1ull)) continue;
    const double v = p == rank ? own : peer_wait(pm, in + pm->recv_idx[e++], seq);
    f = first ? v : f + v;
    first = false;
  }
  return f;
}

// ABLATE (diagnostic builds only, never launched by the product path): 1 = no LDS atomics,
// 2 = no indexed LDS reads, 3 = no staging loads, 4 = no update phase, 5 = no element phase.
// 10 = 3 and 4 together: the element phase alone with its real LDS traffic (what a workgroup whose memory phases were
// hidden completely would still take).
// PEER: synchronised step with the direct peer exchange - shared nodes are pushed to / collected from the
// neighbour ranks inside this kernel (one launch per step, no collective).
template <bool FORCE_ONLY, int ABLATE = 0, bool PEER = false>
__global__ void __launch_bounds__(SAA_LB) fused_step_kernel(DeviceMesh m, const double *__restrict__ d0, const double *__restrict__ dn,
                                  double *__restrict__ out, double *__restrict__ iface,
                                  const double *__restrict__ table_row, double *__restrict__ hist_row, StepConsts k,
                                  const PeerMap *__restrict__ pmap, unsigned seq, const double *__restrict__ tn_dev) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  // graph-replayed steps (saa_step_synced): the time of d^n lives in device memory, the ramp follows from it here
  // (linear_ramp, commons.py:7-11) instead of arriving as a launch argument
  if (tn_dev != nullptr) {
    const double t = *tn_dev;
    k.ramp = t <= 1 ? t : 1.0;
  }
  // (ABLATE == 9: eight steps' worth of blocks in one launch - what the launch boundary and its tail cost)
  const int pblock = plan_block(ABLATE == 9 ? blockIdx.x % m.n_blocks : blockIdx.x, m.n_blocks);
  const BlockDesc bd = m.blocks[pblock];
  const int tid = threadIdx.x, nt = blockDim.x;
  double *rec = lds;                       // [n_owned + n_halo][6]: x y z ux uy uz
  double *acc = lds + 6 * m.max_local;     // force accumulators [n_owned][3]
  const int n_own3 = 3 * bd.n_owned, n_halo3 = 3 * bd.n_halo;
  const int64_t base = 3 * (int64_t)bd.node_start;
  const int32_t *hid = m.halo_ids + bd.halo_off;

  // ---- 0. interior connectivity and halo ids first (see kPreConn) ------------------------------
  unsigned long long T[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tk = 0;
  if (ABLATE == 8) tk = stamp();
  const uint2 *conn = m.conn + bd.elem_off;
  uint2 cpre[kPreConn];
#pragma unroll
  // All prefetch loads are UNCONDITIONAL with clamped (always valid) indices: loads under a
  // divergent branch make hipcc fall back to s_waitcnt vmcnt(0) at the first use (the plan pads
  // conn / halo_ids by one entry so that index 0 exists even for an empty list).
  for (int j = 0; j < kPreConn; ++j) cpre[j] = conn[min(tid + j * nt, max(bd.n_elem - 1, 0))];
  int64_t hg[kPreHalo];
#pragma unroll
  for (int j = 0; j < kPreHalo; ++j) {
    const int i = min(tid + j * nt, max(n_halo3 - 1, 0));
    const int n = i / 3;
    hg[j] = 3 * (int64_t)hid[n] + (i - 3 * n);
  }

  // ---- 1. owned node records (contiguous) -> LDS; zero the accumulators -------------------------
  {
    const double *xo = m.xyz + base;
    const double *uo = d0 + base;
    for (int i = tid; i < n_own3; i += nt) {
      const int n = i / 3, c = i - 3 * n;
      rec[6 * n + c] = (ABLATE == 3 || ABLATE == 10) ? 1.0 * i : xo[i];
      rec[6 * n + 3 + c] = (ABLATE == 3 || ABLATE == 10) ? 1e-3 * i : uo[i];
      acc[3 * n + c] = 0.0;
    }
  }

  // ---- 2. issue the loads whose latency hides under the interior elements ---------------------
  double hx[kPreHalo], hu[kPreHalo];
#pragma unroll
  for (int j = 0; j < kPreHalo; ++j) {
    hx[j] = (ABLATE == 3 || ABLATE == 10) ? 1.0 * tid : m.xyz[hg[j]];
    hu[j] = (ABLATE == 3 || ABLATE == 10) ? 1e-3 * tid : d0[hg[j]];
  }
  double pm_[kPreOwn], pf[kPreOwn], pn[kPreOwn];
  int32_t ptag[kPreOwn];
  if (!FORCE_ONLY) {
#pragma unroll
    for (int j = 0; j < kPreOwn; ++j) {
      const int i = min(tid + j * nt, n_own3 - 1);
      pm_[j] = (ABLATE == 4 || ABLATE == 10) ? 1.0 : (m.mass_node ? m.mass_node[bd.node_start + i / 3] : m.mass[base + i]);
      pf[j] = (ABLATE == 4 || ABLATE == 10) ? 0.0 : (m.fext_yz ? (i % 3 == 0 ? 0.0 : m.fext_yz[bd.node_start + i / 3]) : m.fext[base + i]);
      pn[j] = (ABLATE == 4 || ABLATE == 10) ? 0.0 : dn[base + i];
      ptag[j] = (ABLATE == 4 || ABLATE == 10) ? 0 : m.tag[bd.node_start + i / 3];
    }
  }
  lds_barrier();
  if (ABLATE == 8) {
    const unsigned long long t = stamp();
    T[6] = t - tk;  // staging up to and including the first barrier
    tk = t;
  }

  // ---- 3. interior elements (all four nodes owned) --------------------------------------------
  double sink = 0.0;
  if (ABLATE != 5) {
#pragma unroll
    for (int j = 0; j < kPreConn; ++j)
      if (tid + j * nt < bd.n_interior)
        item_forces<ABLATE, NoHook, true>(cpre[j], rec, acc, bd.n_owned, m.lambda6, m.mu6, tid, sink, T);
    // (rare) further interior sweeps, software-pipelined: the next connectivity entry is in flight while
    // the current element computes - a dependent global load per sweep would expose its L2 latency
    if (tid + kPreConn * nt < bd.n_interior) {
      const int last = bd.n_elem - 1;
      uint2 cur = conn[tid + kPreConn * nt];
      for (int e = tid + kPreConn * nt; e < bd.n_interior; e += nt) {
        const uint2 nxt = conn[min(e + nt, last)];
        item_forces<ABLATE, NoHook, true>(cur, rec, acc, bd.n_owned, m.lambda6, m.mu6, tid, sink, T);
        cur = nxt;
      }
    }
  }
  if (ABLATE == 8) {
    const unsigned long long t = stamp();
    T[7] = t - tk;  // interior loop
    tk = t;
  }
  // first boundary sweep's connectivity: issued now, consumed after the halo records are in LDS
  // Wave balance: a wave's items are a serial chain, so the workgroup is as slow as its busiest wave.
  // The interior list fills waves 0,1,2,.. in 64-item chunks; the boundary list continues with the NEXT
  // wave instead of starting at wave 0 again (39 chunks over 16 waves: at most 3 per wave instead of 4).
  const int shift = (((bd.n_interior + 63) >> 6) % (nt >> 6)) << 6;
  const int e_b0 = bd.n_interior + (tid >= shift ? tid - shift : tid - shift + nt);
  uint2 bcur = conn[min(e_b0, max(bd.n_elem - 1, 0))];

  // ---- 4. halo records -> LDS --------------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < kPreHalo; ++j) {
    const int i = tid + j * nt;
    if (i < n_halo3) {
      const int n = i / 3, c = i - 3 * n;
      rec[6 * (bd.n_owned + n) + c] = hx[j];
      rec[6 * (bd.n_owned + n) + 3 + c] = hu[j];
    }
  }
  for (int i = tid + kPreHalo * nt; i < n_halo3; i += nt) {  // blocks with more halo than the prefetch depth
    const int n = i / 3, c = i - 3 * n;
    const int64_t g = 3 * (int64_t)hid[n] + c;
    rec[6 * (bd.n_owned + n) + c] = m.xyz[g];
    rec[6 * (bd.n_owned + n) + 3 + c] = d0[g];
  }
  lds_barrier();

  // update of one owned dof from its finished force (and publication / prediction for shared nodes)
  auto finish = [&](int i, double mass, double fpre, double dnv, int32_t tag) {
    if (PEER && (tag & kTagShared)) return;  // updated below from the force summed over the ranks
    const int n = i / 3, c = i - 3 * n;
    const double f = acc[3 * n + c];
    if (iface != nullptr && (tag & kTagShared)) iface[3 * (int64_t)(tag >> kTagSlotShift) + c] = f;
    double v = cd_update_dof(f, fpre, mass, rec[6 * n + 3 + c], dnv, k);
    if (tag & (1 << c)) v = 0.0;  // d1[Local_Dirichlet] = 0   (Dynamic_solver.py:20)
    if (ABLATE != 8 && table_row != nullptr && (tag & kTagShared)) {
      // predicted phase: d1[loc_dof_shared] = prediction, recorded as history (Online_predictor.py:298,301)
      const int64_t j = 3 * (int64_t)m.slot_sidx[tag >> kTagSlotShift] + c;
      v = table_row[j];
      if (hist_row != nullptr) hist_row[j] = v;
    }
    out[base + i] = v;
  };
  if (ABLATE == 8) {
    const unsigned long long t = stamp();
    T[8] = t - tk;  // halo records to LDS + barrier
    tk = t;
  }
  // ---- 5. boundary elements (at least one halo node) ----------------------------------------------
  if (ABLATE != 5) {
    const int last = bd.n_elem - 1;
    for (int e = e_b0; e < bd.n_elem; e += nt) {
      const uint2 nxt = conn[min(e + nt, last)];
      item_forces<ABLATE>(bcur, rec, acc, bd.n_owned, m.lambda6, m.mu6, tid, sink, T);
      bcur = nxt;
    }
  }
  if ((ABLATE == 1 || ABLATE == 6) && sink == 12345.678) acc[0] = sink;
  if (ABLATE == 8) {
    const unsigned long long t = stamp();
    T[9] = t - tk;  // boundary loop
    tk = t;
  }
  lds_barrier();
  if (ABLATE == 8) {
    const unsigned long long t = stamp();
    T[10] = t - tk;  // waiting for the slowest wave
    tk = t;
  }
  if (ABLATE == 4 || ABLATE == 10) {
    if (tid == 0) out[base] = acc[0];
    return;
  }

  // ---- 6. owned dofs: write f_int, or update --------------------------------------------------------
  if (FORCE_ONLY) {
    for (int i = tid; i < n_own3; i += nt) {
      const int n = i / 3, c = i - 3 * n;
      out[base + i] = acc[3 * n + c];
    }
    return;
  }
  // shared nodes of this block: partial forces leave for the neighbour ranks now, their values are collected
  // after the update of the other nodes (the xGMI flight time hides under it)
  // (the map is read from device memory HERE, not passed by value: a by-value copy would sit in SGPRs through
  // the element phase, which has none to spare)
  int sh0 = 0, n_sh3 = 0;
  if (PEER) {
    sh0 = pmap->blk_off[pblock];
    n_sh3 = 3 * (pmap->blk_off[pblock + 1] - sh0);
    for (int j = tid; j < n_sh3; j += nt) {
      const int q = sh0 + j / 3, c = j % 3;
      const PeerPushRec r = pmap->push_rec[q];
      peer_push(pmap, r, q, c, acc[3 * (r.info & 0xffff) + c], seq);
    }
  }
#pragma unroll
  for (int j = 0; j < kPreOwn; ++j) {
    const int i = tid + j * nt;
    if (i < n_own3) finish(i, pm_[j], pf[j], pn[j], ptag[j]);
  }
  for (int i = tid + kPreOwn * nt; i < n_own3; i += nt)
    finish(i, m.mass_node ? m.mass_node[bd.node_start + i / 3] : m.mass[base + i],
           m.fext_yz ? (i % 3 == 0 ? 0.0 : m.fext_yz[bd.node_start + i / 3]) : m.fext[base + i], dn[base + i],
           m.tag[bd.node_start + i / 3]);
  if (PEER) {
    for (int j = tid; j < n_sh3; j += nt) {
      const int q = sh0 + j / 3, c = j % 3;
      const PeerRecvRec r = pmap->recv_rec[q];
      const int n = pmap->push_rec[q].info & 0xffff, node = bd.node_start + n;
      const int64_t g = 3 * (int64_t)node + c;
      // operands first: their latency overlaps the poll
      const double fe = m.fext[g], ma = m.mass[g], dnv = dn[g];
      const int32_t tag = m.tag[node];
      const double f = peer_collect(pmap, r, q, c, acc[3 * n + c], seq);
      double v = cd_update_dof(f, fe, ma, rec[6 * n + 3 + c], dnv, k);  // Dynamic_solver.py:26-32
      if (tag & (1 << c)) v = 0.0;
      out[g] = v;
      if (hist_row != nullptr) hist_row[3 * (int64_t)r.sidx + c] = v;  // Online_predictor.py:260
    }
  }
  if (ABLATE == 8) {  // stamps leave through a buffer of their own (passed in place of the history row)
    T[11] = stamp() - tk;  // update phase
    if ((tid & 63) == 0) {
      unsigned long long *dbg = reinterpret_cast<unsigned long long *>(hist_row) +
                                12 * ((size_t)blockIdx.x * (nt >> 6) + (tid >> 6));
      for (int j = 0; j < 12; ++j) dbg[j] = T[j];
    }
  }
}

template __global__ void fused_step_kernel<false>(DeviceMesh, const double *, const double *, double *, double *,
                                                   const double *, double *, StepConsts, const PeerMap *, unsigned,
                                                   const double *);
template __global__ void fused_step_kernel<true>(DeviceMesh, const double *, const double *, double *, double *,
                                                  const double *, double *, StepConsts, const PeerMap *, unsigned,
                                                  const double *);
template __global__ void fused_step_kernel<false, 0, true>(DeviceMesh, const double *, const double *, double *, double *,
                                                            const double *, double *, StepConsts, const PeerMap *, unsigned,
                                                            const double *);

// Attach-time proof of the peer path: one exchange of known values (own[3*q+c], node-sorted order) -> the sums
// in the caller's shared order.
__global__ void peer_selftest_kernel(PeerMap pm, const double *__restrict__ own, double *__restrict__ out,
                                     unsigned seq) {
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gthreads = gridDim.x * blockDim.x;
  for (int j = gtid; j < 3 * pm.n_shared; j += gthreads) peer_push(&pm, pm.push_rec[j / 3], j / 3, j % 3, own[j], seq);
  for (int j = gtid; j < 3 * pm.n_shared; j += gthreads) {
    const PeerRecvRec r = pm.recv_rec[j / 3];
    out[3 * (int64_t)r.sidx + j % 3] = peer_collect(&pm, r, j / 3, j % 3, own[j], seq);
  }
}

// ---------------------------------------------------------------------------------------------
// Resident multi-step kernel.  Same arithmetic as fused_step_kernel (item_forces, cd_update_dof), different data
// flow: the workgroup keeps its block in LDS for the whole launch,
//   rec   [max_local][6]  x y z ux uy uz   (x static; u of owned nodes updated in place, u of halo nodes re-read)
//   acc   [max_owned][3] force accumulators,  dnl [3*max_owned] d^(n-1) of the owned dofs,
//   massl / fextl [max_owned] nodal mass and (0,v,v) load,  tagl [max_owned],  connl [max_items] work items,
//   hgl [3*max_halo] entry index of every halo dof,
// so that per step only 48 B per owned node leave the CU (new displacements as stamped entries) and 48 B per halo
// node enter it.
// Step s of a block needs d^(n+s) of its halo nodes, written by their owners at the end of step s-1.  No flags, no
// grid barrier: every published value is a 16-byte entry of two words, each = 32 bits of the double + the 32-bit
// step count (the protocol of the peer exchange above), so a reader can load speculatively - half-way through its
// first round of interior items, consumed at the end of that round - and recognise a value that has not arrived yet.
// Two entry buffers alternate by step parity: an owner cannot overwrite what a reader still needs, because it cannot
// get two steps ahead of a block it reads from.
// All workgroups must be co-resident (proved by a census launch at set-up); every wait is bounded
// (PersistArgs::timeout_ticks).
// ---------------------------------------------------------------------------------------------
// Cross-workgroup traffic of the resident kernel uses agent-scope relaxed accesses only (global_load/store ... sc1:
// served by / written through to the level all XCDs share) - no L2 write-back or invalidate inside the step loop
// (an acquire in a polling loop would invalidate the XCD's L2 on every iteration).

// (doubles) d^(n-1) of the owned dofs follows the node records and the [max_owned][3] force accumulators
// n_words 4-byte words of a contiguous global array straight into LDS by LDS-DMA (global_load_lds_dword: the data never
// passes through registers and nothing waits for it until the next vmcnt(0) - the workgroup barrier behind the image
// staging).  The hardware writes lane l of a wave-instruction to (wave-uniform LDS base) + 4*l, so every wave copies whole
// 64-word chunks; 4-byte granules because neither side is aligned better than that for every block.
typedef __attribute__((address_space(1))) const uint32_t *GlobalWords;
typedef __attribute__((address_space(3))) uint32_t *LdsWords;
__device__ __forceinline__ void dma_words(const void *src, void *dst_lds, int n_words, int tid, int nt) {
  GlobalWords s = (GlobalWords)src;
  LdsWords d = (LdsWords)dst_lds;
  for (int w0 = tid & ~63; w0 < n_words; w0 += nt) {  // w0: first word of this wave's chunk (wave-uniform)
    const int i = w0 + (tid & 63);
    if (i < n_words) __builtin_amdgcn_global_load_lds(s + i, d + w0, 4, 0, 0);
  }
}

__host__ __device__ inline int persist_off_dn(int max_local, int max_owned) { return 6 * max_local + 3 * max_owned; }

#ifndef SAA_PERSIST_PRE
#define SAA_PERSIST_PRE 2
#endif
constexpr int kPH = SAA_PERSIST_PRE;  // stamped halo entries per thread in flight during the first round
// PREDICT: the predicted phase (table / history rows); a separate instantiation keeps its pointers out of the plain
// kernel's scalar registers.  The argument block is read field by field where it is needed for the same reason.
// PEER: synchronised steps with the direct peer exchange (shared nodes pushed to / collected from the neighbour
// ranks inside the step loop, like fused_step_kernel<.., PEER>).
// The argument block travels BY VALUE in the kernel-argument segment (no separate kernel or copy to place it in device
// memory: 4.8 of the 183 us of a 20-step call) but is read through an opaque constant-address-space pointer into that
// segment, field by field where needed: named as a parameter the compiler loads it up front and holds ~60 scalar
// registers through the item loops, which have none to spare.
struct PersistKernArgs {  // layout of the kernel-argument segment
  DeviceMesh m;
  StepConsts k;
  PersistArgs a;
};
typedef const __attribute__((address_space(4))) PersistArgs *PersistArgsPtr;
typedef const __attribute__((address_space(4))) StepConsts *StepConstsPtr;
template <bool PREDICT, bool PEER>
__global__ void __launch_bounds__(SAA_LB) persistent_steps_kernel(DeviceMesh m, StepConsts k, PersistArgs /*see ap*/) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  PersistArgsPtr ap;
  {
    const __attribute__((address_space(4))) char *seg =
        (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
    seg += offsetof(PersistKernArgs, a);
    asm volatile("" : "+s"(seg));
    ap = reinterpret_cast<PersistArgsPtr>(seg);
  }
  struct {
    double *g0, *g1;
    PeerEntry *entries;
    int64_t entry_stride;
    int32_t step_base, nsteps, ramp_on, max_items;
    double tn0;
  } a = {ap->g0, ap->g1, ap->entries, ap->entry_stride, ap->step_base, ap->nsteps, ap->ramp_on, ap->max_items, ap->tn0};
  const int pblock = plan_block(blockIdx.x, m.n_blocks);
  const BlockDesc bd = m.blocks[pblock];
  const int tid = threadIdx.x, nt = blockDim.x;
  double *rec = lds;
  double *acc = lds + 6 * m.max_local;
  // offsets (in doubles) of the arrays behind the accumulators
  int o_dn = persist_off_dn(m.max_local, m.max_owned), o_mass = o_dn + 3 * m.max_owned, o_fext = o_mass + m.max_owned,
      o_conn = o_fext + m.max_owned;
  int64_t base = 3 * (int64_t)bd.node_start;
  if (PEER) {
    // the variant shortest of scalar registers keeps what only the halo and update phases use - wave-uniform values -
    // in vector registers, of which it has a dozen to spare (the OFFSETS, not the pointers: an opaque pointer is a generic
    // one, and the LDS arrays behind it would be read with flat loads)
    asm volatile("" : "+v"(o_dn), "+v"(o_mass), "+v"(o_fext), "+v"(base));
  }
  double *dnl = lds + o_dn;
  double *massl = lds + o_mass;
  double *fextl = lds + o_fext;
  uint2 *connl = reinterpret_cast<uint2 *>(lds + o_conn);
  int o_tag = 2 * (o_conn + a.max_items), o_hg = o_tag + m.max_owned;  // in 4-byte words
  if (PEER) asm volatile("" : "+v"(o_tag), "+v"(o_hg));
  int32_t *tagl = reinterpret_cast<int32_t *>(lds) + o_tag;
  int32_t *hgl = reinterpret_cast<int32_t *>(lds) + o_hg;  // [3 * max_halo] entry index 3*node+c of every halo dof
  const int n_own3 = 3 * bd.n_owned, n_halo3 = 3 * bd.n_halo;
  const int32_t *hid = m.halo_ids + bd.halo_off;

  // ---- census launch (once per handle, at set-up): are ALL workgroups of this grid on the chip at the same time?
  //      Every workgroup checks in and waits - bounded - until the count is complete.  The launch is a plain one (the
  //      cooperative-launch API costs 15-19 us per launch and, per MI355X_MICROARCH.md "Residency", accepts grids one
  //      block per CU larger than what the hardware admits when the kernel's SGPR count sits in the 97-112 band - this
  //      kernel's does): the census is the ground truth the step loop's stamped waits rely on. ---------------------
  if (ap->census != nullptr) {
    if (threadIdx.x == 0) {
      int32_t *cnt = ap->census;
      __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const long long t0 = wall_clock64();
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (int32_t)gridDim.x) {
        if (__hip_atomic_load(ap->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;  // already decided
        if (wall_clock64() - t0 > ap->timeout_ticks) {  // some workgroup is still queued behind the resident ones
          __hip_atomic_store(ap->err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    return;
  }

  // ---- once: the block's image (halo displacements of the first step included) -----------------------
  //      A launch of 20 steps pays this staging with 4 % of its time, so every request leaves before the first answer is
  //      used: what is a plain copy of a contiguous array - d^(n-1), nodal mass and load, tags, work items: half of the
  //      image - goes global -> LDS by LDS-DMA (no registers, nothing to wait for until the barrier below), the node
  //      records, which interleave coordinates and displacements, through registers, all sweeps requested before the
  //      first is stored (in loops that load and store per sweep the requests of one sweep only left when the previous
  //      sweep's data had arrived: eight memory round trips in a row at the start of every launch).
  {
    constexpr int kOwnSweeps = 3, kHaloSweeps = 2;
    int64_t hgi[kHaloSweeps];
    int32_t hnode[kHaloSweeps];
    double ox[kOwnSweeps], ou[kOwnSweeps], hx[kHaloSweeps], hu[kHaloSweeps];
    // (the halo ids first: the only requests something else depends on - while an LDS-DMA is in flight the compiler
    // waits for ALL outstanding memory operations at the first use of a loaded value, so they travel with everything else)
    // (all of them unconditional, with clamped indices - the plan pads its lists by one entry: a load under a branch is
    // waited for where the branch ends)
#pragma unroll
    for (int j = 0; j < kHaloSweeps; ++j) hnode[j] = hid[min(tid + j * nt, max(n_halo3 - 1, 0)) / 3];
    dma_words(a.g1 + base, dnl, 2 * n_own3, tid, nt);
    dma_words(m.mass_node + bd.node_start, massl, 2 * bd.n_owned, tid, nt);
    dma_words(m.fext_yz + bd.node_start, fextl, 2 * bd.n_owned, tid, nt);
    dma_words(m.tag + bd.node_start, tagl, bd.n_owned, tid, nt);
    dma_words(m.conn + bd.elem_off, connl, 2 * bd.n_elem, tid, nt);
#pragma unroll
    for (int j = 0; j < kOwnSweeps; ++j) {
      const int i = min(tid + j * nt, n_own3 - 1);
      ox[j] = m.xyz[base + i];
      ou[j] = a.g0[base + i];
    }
#pragma unroll
    for (int j = 0; j < kHaloSweeps; ++j) {
      hgi[j] = 3 * (int64_t)hnode[j] + (min(tid + j * nt, max(n_halo3 - 1, 0)) % 3);
      hx[j] = m.xyz[hgi[j]];
      hu[j] = a.g0[hgi[j]];
    }
#pragma unroll
    for (int j = 0; j < kOwnSweeps; ++j) {
      const int i = tid + j * nt;
      if (i < n_own3) {
        const int n = i / 3, c = i - 3 * n;
        rec[6 * n + c] = ox[j];
        rec[6 * n + 3 + c] = ou[j];
        acc[i] = 0.0;
      }
    }
    for (int i = tid + kOwnSweeps * nt; i < n_own3; i += nt) {  // (blocks larger than the depth above)
      const int n = i / 3, c = i - 3 * n;
      rec[6 * n + c] = m.xyz[base + i];
      rec[6 * n + 3 + c] = a.g0[base + i];
      acc[i] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < kHaloSweeps; ++j) {
      const int i = tid + j * nt;
      if (i < n_halo3) {
        const int n = i / 3, c = i - 3 * n;
        rec[6 * (bd.n_owned + n) + c] = hx[j];
        rec[6 * (bd.n_owned + n) + 3 + c] = hu[j];
        hgl[i] = (int32_t)hgi[j];
      }
    }
    for (int i = tid + kHaloSweeps * nt; i < n_halo3; i += nt) {
      const int n = i / 3, c = i - 3 * n;
      const int64_t g = 3 * (int64_t)hid[n] + c;
      rec[6 * (bd.n_owned + n) + c] = m.xyz[g];
      rec[6 * (bd.n_owned + n) + 3 + c] = a.g0[g];
      hgl[i] = (int32_t)g;
    }
  }
  if (tid == 0 && n_halo3 == 0) hgl[0] = 0;  // the clamped prefetch below reads index 0 even without a halo
  if (PEER) {
    // the block's push / receive records (static; one pair per shared node it owns) behind the image: the shared nodes
    // are a serial tail of a face block's step, and two dependent global loads less in it shorten every block's step
    const PeerMap *pm0 = ap->peer;
    const int s0 = pm0->blk_off[pblock], ns = pm0->blk_off[pblock + 1] - s0;
    PeerPushRec *prl0 = reinterpret_cast<PeerPushRec *>(reinterpret_cast<char *>(lds) + ap->peer_rec_off);
    PeerRecvRec *rrl0 = reinterpret_cast<PeerRecvRec *>(prl0 + ns);
    PeerSecondRec *srl0 = reinterpret_cast<PeerSecondRec *>(rrl0 + ns);
    for (int q = tid; q < ns; q += nt) {
      prl0[q] = pm0->push_rec[s0 + q];
      rrl0[q] = pm0->recv_rec[s0 + q];
      srl0[q] = pm0->second_rec[s0 + q];
    }
  }
  __syncthreads();

  const int n_pre = min(bd.n_interior, nt);  // items of the first round: they need no halo record
  const int n_ir = bd.n_interior - n_pre, n_ir_pad = (n_ir + 63) & ~63;  // interior items left for the second phase
  const int n_post = n_ir_pad + (bd.n_elem - bd.n_interior);
  double tn = a.tn0;
  double sink = 0.0;
  // The work items of a lane are the same in every step of the launch: the one of the first round and the first two of
  // the second phase stay in registers (two each) instead of being read from LDS again - not for the read's bandwidth
  // but for its round trip, which every wave sat out right behind a barrier, twice per step, before it could request a
  // single node record.  (Made opaque once per step below: otherwise the compiler hoists their unpacking - fifteen more
  // registers - out of the step loop as well.)
  const uint2 kNullItem = {0u, 2u << 28};
  uint2 w_first = tid < n_pre ? connl[tid] : kNullItem, w_post[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int p = tid + j * nt;
    const int e = p < n_ir_pad ? (p < n_ir ? n_pre + p : -1) : (p < n_post ? bd.n_interior + (p - n_ir_pad) : -1);
    w_post[j] = e >= 0 ? connl[e] : kNullItem;
  }
#ifdef SAA_PERSIST_STAMPS
  unsigned long long T[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tk = stamp();
#define PSTAMP(j) { const unsigned long long t_ = stamp(); T[j] += t_ - tk; tk = t_; }
#else
#define PSTAMP(j)
#endif
  // what the first half of a step needs of the argument block (PEER: re-read at the end of the step before, in front of
  // its last barrier, through a pointer made opaque once per step - see the update phase)
  int h_ramp_on;
  unsigned h_step_base;
  const PeerEntry *h_entries;
  int64_t h_entry_stride;
  auto load_head = [&]() {
    PersistArgsPtr ah = ap;
    if (PEER) asm volatile("" : "+s"(ah));
    h_ramp_on = ah->ramp_on;
    h_step_base = (unsigned)ah->step_base;
    h_entries = ah->entries;
    h_entry_stride = ah->entry_stride;
  };
  load_head();
  for (int s = 0; s < a.nsteps; ++s) {
    const double ramp_now = h_ramp_on ? (tn <= 1 ? tn : 1.0) : 1.0;  // commons.py:7-11 at the time of d^n
    if (!PEER) k.ramp = ramp_now;
    // Opaque copy of the thread index for the halo and update phases: whatever is derived from it is recomputed
    // every step.  Derived from `tid` the compiler hoists those per-thread constants (indices, addresses) out of
    // the step loop, keeps them alive through the item loops and spills.
    int ltid = tid;
    asm volatile("" : "+v"(ltid));
    const unsigned want = h_step_base + (unsigned)s;  // stamp of d^(n+s), written by its owner in step s-1
    const PeerEntry *ein = h_entries + (int64_t)(s & 1) * h_entry_stride;
    // ---- 1. first round: interior items; the halo displacements are requested half-way through it (late enough
    //         for the neighbours' stores of the previous step to have landed, early enough to arrive by its end) --
    unsigned long long hlo[kPH], hhi[kPH];
    auto fetch = [&]() {
      if (s > 0) {
#pragma unroll
        for (int j = 0; j < kPH; ++j) {
          const PeerEntry *e = ein + hgl[min(ltid + j * nt, max(n_halo3 - 1, 0))];
          hlo[j] = __hip_atomic_load(&e->lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          hhi[j] = __hip_atomic_load(&e->hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    };
    // (interior items: the variant of the item code without ownership tests; a lane without an item of this round holds
    // the null item, which only runs the hook)
    asm volatile("" : "+v"(w_first.x), "+v"(w_first.y), "+v"(w_post[0].x), "+v"(w_post[0].y), "+v"(w_post[1].x), "+v"(w_post[1].y));
    item_forces<0, decltype(fetch), true>(w_first, rec, acc, bd.n_owned, m.lambda6, m.mu6, tid, sink, nullptr, fetch);
    PSTAMP(0)
    PSTAMP(1)
    // ---- 2. halo displacements -> LDS (an entry still carrying an older stamp is simply read again) ------
    if (s > 0) {
      // (the entry address is recomputed on the rare retry path: keeping it live would cost registers)
      auto settle = [&](int i, unsigned long long lo, unsigned long long hi) {
        if ((unsigned)(lo >> 32) != want || (unsigned)(hi >> 32) != want) {
          const PeerEntry *e = ein + hgl[i];
          const long long t0 = wall_clock64();
          do {
            int32_t *errp = *(int32_t *volatile const __attribute__((address_space(4))) *)&ap->err;
            if (__hip_atomic_load(errp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;  // failed before
            if (wall_clock64() - t0 > *(volatile const __attribute__((address_space(4))) int64_t *)&ap->timeout_ticks) {
              // workgroups not co-resident, or a fault elsewhere: report, do not hang
              __hip_atomic_store(errp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              break;
            }
            __builtin_amdgcn_s_sleep(1);
            lo = __hip_atomic_load(&e->lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hi = __hip_atomic_load(&e->hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } while ((unsigned)(lo >> 32) != want || (unsigned)(hi >> 32) != want);
        }
        return __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull)));
      };
#pragma unroll
      for (int j = 0; j < kPH; ++j) {
        const int i = ltid + j * nt;
        if (i < n_halo3) {
          const int n = i / 3, c = i - 3 * n;
          rec[6 * (bd.n_owned + n) + 3 + c] = settle(i, hlo[j], hhi[j]);
        }
      }
      for (int i = ltid + kPH * nt; i < n_halo3; i += nt) {
        const int n = i / 3, c = i - 3 * n;
        rec[6 * (bd.n_owned + n) + 3 + c] = settle(i, 0ull, 0ull);  // stamp 0 never matches: loads on the retry path
      }
    }
    PSTAMP(2)
    lds_barrier();
    PSTAMP(3)
    // ---- 3. the other items in ONE list: the rest of the interior ones, then the boundary ones - no barrier and no
    //         partly filled round between them -----------------------------------------------------------------
    //         The boundary part starts on a wave boundary (the interior part is padded to a multiple of 64 slots):
    //         the plan packed each list for the LDS banks from ITS first item, in groups of 16 / 32 lanes.
    //         (the ownership-test-free variant of the interior items, which the first round and the fused kernel use,
    //         was measured here too, behind a wave-uniform branch: 8.60-8.67 against 8.43 us/step in round 2, 8.06 against
    //         8.03 in round 3 with 104 instead of 112 registers in use)
    if (tid < n_post) item_forces<0>(w_post[0], rec, acc, bd.n_owned, m.lambda6, m.mu6, tid, sink);
    if (tid + nt < n_post) item_forces<0>(w_post[1], rec, acc, bd.n_owned, m.lambda6, m.mu6, tid, sink);
    for (int p = tid + 2 * nt; p < n_post; p += nt) {  // (blocks with more than two sweeps in this phase)
      const int e = p < n_ir_pad ? (p < n_ir ? n_pre + p : -1) : bd.n_interior + (p - n_ir_pad);
      if (e >= 0) item_forces<0>(connl[e], rec, acc, bd.n_owned, m.lambda6, m.mu6, tid, sink);
    }
    PSTAMP(4)
    // ---- 5. update of the owned dofs: LDS operands; the new value leaves as a plain double (state) and as a
    //         stamped entry (what the neighbouring workgroups read in the next step) ------------------------
    // (PEER, the variant shortest of scalar registers: the argument block is addressed through a pointer made opaque once
    // per step, so that its fields are loaded here, where they are used, instead of being held through the item loops -
    // requested in front of the barrier, whose wait covers their latency; the other variants let the compiler hoist
    // these loads out of the step loop)
    PersistArgsPtr aq = ap;
    if (PEER) asm volatile("" : "+s"(aq));
    double *gnext = (s & 1) ? aq->g0 : aq->g1;
    PeerEntry *eout = aq->entries + (int64_t)((s + 1) & 1) * aq->entry_stride + base;
    const unsigned stampw = (unsigned)(aq->step_base + s + 1);
    // the state buffers only need the last two steps of the launch (d^n and d^(n-1) for whoever comes next)
    const bool keep = s + 2 >= aq->nsteps;
    // new value of owned dof i: state buffer, stamped entry for the neighbouring workgroups, LDS image
    // trajectory recorder: is this step's result a column of the caller's matrix?
    double *traj_col = nullptr;
    int64_t traj_ld = 0;
    if (aq->traj != nullptr) {
      const int64_t idx = aq->step_index0 + s, every = aq->save_every;
      if (idx % every == 0 && idx / every < aq->traj_cols) {
        traj_col = aq->traj + idx / every;
        traj_ld = aq->traj_cols;
      }
    }
    if (PEER) {  // step constants from the argument block, not from registers held since the launch
      StepConstsPtr kp = &aq->consts;
      asm volatile("" : "+s"(kp));
      k.dt = kp->dt; k.dt2 = kp->dt2; k.half_dt = kp->half_dt; k.alpha = kp->alpha; k.half_alpha = kp->half_alpha;
      k.ramp = ramp_now;
    }
    // PEER: so is the range of this block's shared nodes in the peer map - a chain of three dependent scalar loads that
    // every wave of every block would otherwise sit out between the barrier and its first dof (0.4 us per step)
    const PeerMap *pm = nullptr;
    int sh0 = 0, n_sh3 = 0;
    unsigned pseq = 0;
    if (PEER) {
      pm = aq->peer;
      pseq = aq->peer_seq_base + (unsigned)s + 1u;  // the host keeps a launch clear of the wrap to 0 ("never written")
      sh0 = pm->blk_off[pblock];
      n_sh3 = 3 * (pm->blk_off[pblock + 1] - sh0);
    }
    lds_barrier();
    PSTAMP(5)
    asm volatile("" : "+v"(ltid));
    auto commit = [&](int i, int n, int c, double u, double v) {
      if (keep) gnext[base + i] = v;
      if (traj_col != nullptr) traj_col[(3 * (int64_t)aq->new_to_old[bd.node_start + n] + c) * traj_ld] = v;
      {
        // one 16-byte store, agent scope (write-through to the level all XCDs share); each half validates itself
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const u32x4 w4 = {(unsigned)b, stampw, (unsigned)(b >> 32), stampw};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(eout + i), "v"(w4) : "memory");
      }
      dnl[i] = u;
      rec[6 * n + 3 + c] = v;
      acc[3 * n + c] = 0.0;
    };
    // PEER: partial forces of this block's shared nodes leave for the neighbour ranks first; their values are
    // collected after the update of the other nodes (the xGMI flight time hides under it)
    const PeerPushRec *prl = nullptr;
    const PeerRecvRec *rrl = nullptr;
    const PeerSecondRec *srl = nullptr;
    if (PEER) {
      int off = aq->peer_rec_off;  // (kept in a vector register like the other offsets of this variant)
      asm volatile("" : "+v"(off));
      prl = reinterpret_cast<const PeerPushRec *>(reinterpret_cast<const char *>(lds) + off);
      rrl = reinterpret_cast<const PeerRecvRec *>(prl + n_sh3 / 3);
      srl = reinterpret_cast<const PeerSecondRec *>(rrl + n_sh3 / 3);
      for (int j = ltid; j < n_sh3; j += nt) {
        const int c = j % 3;
        const PeerPushRec r = prl[j / 3];
        const double f = acc[3 * (r.info & 0xffff) + c];
        if (((r.info >> 16) & 0xff) == 2) {  // three holders: both neighbours from the LDS records
          const PeerSecondRec r2 = srl[j / 3];
          const int64_t par = pseq & 1u;
          peer_store(r.dst0 + par * r.pstride0 + c, f, pseq);
          peer_store(r2.dst1 + par * r2.pstride1 + c, f, pseq);
        } else {
          peer_push(pm, r, sh0 + j / 3, c, f, pseq);
        }
      }
    }
#ifdef SAA_PEER_EMULATE_LATENCY  // tools/peer_latency.py: nothing pushed now counts as visible before t_push + that many
    const long long t_push = wall_clock64();  // ticks of the 100 MHz wall clock (a stand-in for xGMI's delivery time)
#endif
    // (one dof at a time: requesting the operands of a lane's two or three dofs up front and interleaving their
    // division chains was measured - 9.35 against 8.95 us/step at 1M tets in round 2, 7.99 against 7.87 in round 3 with
    // the first two dofs of a lane as a pair)
    auto update_dof = [&](int i, int n, int c, double u, int32_t tag, double f, double fe, double ma, double dnv) {
      if (PEER && (tag & kTagShared)) return;  // below, from the force summed over the ranks
      double v = cd_update_dof(f, fe, ma, u, dnv, k);
      if (tag & (1 << c)) v = 0.0;  // d1[Local_Dirichlet] = 0   (Dynamic_solver.py:20)
      if (PREDICT && (tag & kTagShared)) {
        // predicted phase: d1[loc_dof_shared] = prediction, recorded as history (Online_predictor.py:298,301)
        const int64_t j = 3 * (int64_t)m.slot_sidx[tag >> kTagSlotShift] + c, w = aq->width;
        v = aq->table[(aq->table_row0 + s) * w + j];
        if (aq->hist != nullptr) aq->hist[(aq->hist_row0 + s) * w + j] = v;
      }
      commit(i, n, c, u, v);
    };
    for (int i = ltid; i < n_own3; i += nt) {
      const int n = i / 3, c = i - 3 * n;
      update_dof(i, n, c, rec[6 * n + 3 + c], tagl[n], acc[i], c == 0 ? 0.0 : fextl[n], massl[n], dnl[i]);
    }
    if (PEER) {
      for (int j = ltid; j < n_sh3; j += nt) {
        const int q = sh0 + j / 3, c = j % 3;
        const PeerRecvRec r = rrl[j / 3];
        const int info = prl[j / 3].info, n = info & 0xffff, i = 3 * n + c;
#ifdef SAA_PEER_EMULATE_LATENCY
        while (wall_clock64() - t_push < SAA_PEER_EMULATE_LATENCY) __builtin_amdgcn_s_sleep(1);
#endif
        double f;
        if (((info >> 16) & 0xff) == 2) {
          // three holders: the two neighbours' values, summed with this rank's own in RANK order like peer_collect does
          // (the neighbour entries are in rank order; a + b == b + a)
          int recv1 = srl[j / 3].recv1;
          asm volatile("" : "+v"(recv1));
          const PeerEntry *in = pm->inbox + (int64_t)(pseq & 1u) * pm->parity_stride + c;
          const double own = acc[3 * n + c];
          const double va = peer_wait(pm, in + r.recv0, pseq);
          const double vb = peer_wait(pm, in + recv1, pseq);
          f = (info & kPeerInfoHighest) ? (va + vb) + own : (own + va) + vb;
        } else {
          f = peer_collect(pm, r, q, c, acc[3 * n + c], pseq);
        }
        const double u = rec[6 * n + 3 + c];
        double v = cd_update_dof(f, c == 0 ? 0.0 : fextl[n], massl[n], u, dnl[i], k);  // Dynamic_solver.py:26-32
        if (tagl[n] & (1 << c)) v = 0.0;
        if (aq->hist != nullptr) aq->hist[(aq->hist_row0 + s) * aq->width + 3 * (int64_t)r.sidx + c] = v;  // Online_predictor.py:260
        commit(i, n, c, u, v);
      }
    }
    tn = tn + k.dt;  // Data_prepare.py:235
    PSTAMP(6)
    if (PEER) load_head();
    lds_barrier();
    PSTAMP(7)
  }
#ifdef SAA_PERSIST_STAMPS
  if ((tid & 63) == 0 && ap->hist != nullptr) {  // diagnostic build: per-wave cycle totals leave through `hist`
    unsigned long long *dbg = reinterpret_cast<unsigned long long *>(ap->hist) + 8 * ((size_t)blockIdx.x * (nt >> 6) + (tid >> 6));
    for (int j = 0; j < 8; ++j) dbg[j] = T[j];
  }
#endif
  if (sink == 12345.678) acc[0] = sink;
}

// ---------------------------------------------------------------------------------------------
// Deterministic mode (saa_set_deterministic): the same element arithmetic, but no floating-point atomics.  The item
// kernel writes the five force vectors of every item to global memory; the node kernel adds, for every owned node, the
// vectors addressed to it in a FIXED order (ascending item, then slot: the list the host built from the plan) and
// applies the update.  Bit-identical results run after run and on every device, at several times the cost of the fused
// kernel (78 MB of item forces per step at 1M tets): a verification mode ("hard parts").
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SAA_LB) det_items_kernel(DeviceMesh m, const double *__restrict__ d0,
                                                           double *__restrict__ item_force) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const BlockDesc bd = m.blocks[blockIdx.x];
  const int tid = threadIdx.x, nt = blockDim.x;
  double *rec = lds;
  const int64_t base = 3 * (int64_t)bd.node_start;
  const int32_t *hid = m.halo_ids + bd.halo_off;
  for (int i = tid; i < 3 * bd.n_owned; i += nt) {
    const int n = i / 3, c = i - 3 * n;
    rec[6 * n + c] = m.xyz[base + i];
    rec[6 * n + 3 + c] = d0[base + i];
  }
  for (int i = tid; i < 3 * bd.n_halo; i += nt) {
    const int n = i / 3, c = i - 3 * n;
    const int64_t g = 3 * (int64_t)hid[n] + c;
    rec[6 * (bd.n_owned + n) + c] = m.xyz[g];
    rec[6 * (bd.n_owned + n) + 3 + c] = d0[g];
  }
  __syncthreads();
  for (int e = tid; e < bd.n_elem; e += nt) {
    const Item it = unpack(m.conn[bd.elem_off + e]);
    if (it.null) continue;
    // exactly the evaluation order of item_forces: A as (p; a, r, q), B as (p; b, q, r), face forces summed in registers
    const Rec rp = load_rec(rec, it.p), rq = load_rec(rec, it.q), rr = load_rec(rec, it.r), ra = load_rec(rec, it.a);
    Vec3 fa, fp, fq, fr, fb = {0, 0, 0};
    const PairShared pg = pair_shared(rp.x, rq.x, rr.x, rp.u, rq.u, rr.u);
    tet_a_forces(pg, rp.x, ra.x, rp.u, ra.u, m.lambda6, m.mu6, fa, fr, fq);
    if (it.pair) {
      const Rec rb = load_rec(rec, it.b);
      Vec3 gq, gr;
      tet_b_forces(pg, rp.x, rb.x, rp.u, rb.u, m.lambda6, m.mu6, fb, gq, gr);
      fq = add3(fq, gq);
      fr = add3(fr, gr);
    }
    fp = neg_sum3(add3(fa, fb), fq, fr);
    double *o = item_force + 15 * (int64_t)(bd.elem_off + e);
    o[0] = fa.x; o[1] = fa.y; o[2] = fa.z; o[3] = fp.x; o[4] = fp.y; o[5] = fp.z; o[6] = fq.x; o[7] = fq.y; o[8] = fq.z;
    o[9] = fr.x; o[10] = fr.y; o[11] = fr.z; o[12] = fb.x; o[13] = fb.y; o[14] = fb.z;
  }
}

template <bool FORCE_ONLY>
__global__ void det_nodes_kernel(DeviceMesh m, DetLists det, const double *__restrict__ d0, const double *__restrict__ dn,
                                 double *__restrict__ out, double *__restrict__ iface, const double *__restrict__ table_row,
                                 double *__restrict__ hist_row, StepConsts k) {
  const int64_t node = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (node >= m.n_nodes) return;
  double f[3] = {0.0, 0.0, 0.0};
  for (int64_t j = det.contrib_off[node]; j < det.contrib_off[node + 1]; ++j) {  // fixed order: item, then slot
    const int32_t id = det.contrib[j];
    const double *v = det.item_force + 15 * (int64_t)(id >> 3) + 3 * (id & 7);
    f[0] += v[0];
    f[1] += v[1];
    f[2] += v[2];
  }
  const int32_t tag = m.tag[node];
  for (int c = 0; c < 3; ++c) {
    const int64_t i = 3 * node + c;
    if (FORCE_ONLY) {
      out[i] = f[c];
      continue;
    }
    if (iface != nullptr && (tag & kTagShared)) iface[3 * (int64_t)(tag >> kTagSlotShift) + c] = f[c];
    double v = cd_update_dof(f[c], m.fext[i], m.mass[i], d0[i], dn[i], k);
    if (tag & (1 << c)) v = 0.0;  // d1[Local_Dirichlet] = 0   (Dynamic_solver.py:20)
    if (table_row != nullptr && (tag & kTagShared)) {  // Online_predictor.py:298,301
      const int64_t j = 3 * (int64_t)m.slot_sidx[tag >> kTagSlotShift] + c;
      v = table_row[j];
      if (hist_row != nullptr) hist_row[j] = v;
    }
    out[i] = v;
  }
}

// After the all-reduce: shared nodes get the update from the summed force (Dynamic_solver.py:26-32),
// optional history record (Online_predictor.py:260); slots of shared nodes this rank does not hold
// are zeroed so that the next all-reduce sees only fresh partial forces.
// tn_in / tn_out (graph-replayed steps): the device-side clock - every thread takes the ramp from *tn_in, one thread
// writes *tn_out = *tn_in + dt (Data_prepare.py:235; another slot: no thread of this launch reads what it writes).
__global__ void iface_finish_kernel(DeviceMesh m, SharedMap sh, const double *__restrict__ d0,
                                    const double *__restrict__ dn, double *__restrict__ d1,
                                    double *__restrict__ iface, double *__restrict__ hist_row, StepConsts k,
                                    const double *__restrict__ tn_in, double *__restrict__ tn_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (tn_in != nullptr) {
    const double t = *tn_in;
    k.ramp = t <= 1 ? t : 1.0;
    if (i == 0) *tn_out = t + k.dt;
  }
  const int n_local = 3 * sh.n_shared;
  if (i < n_local) {
    const int s = i / 3, c = i - 3 * s;
    const int node = sh.node[s];
    const int64_t g = 3 * (int64_t)node + c;
    const double f = iface[3 * (int64_t)sh.slot[s] + c];
    double v = cd_update_dof(f, m.fext[g], m.mass[g], d0[g], dn[g], k);
    if (m.tag[node] & (1 << c)) v = 0.0;
    d1[g] = v;
    if (hist_row) hist_row[i] = v;
  } else if (i < n_local + 3 * sh.n_foreign) {
    const int j = i - n_local;
    iface[3 * (int64_t)sh.foreign_slot[j / 3] + (j % 3)] = 0.0;
  }
}

// Predicted phase: d1[loc_dof_shared] = table row (Online_predictor.py:298), history record (:301).
__global__ void halo_overwrite_kernel(SharedMap sh, const double *__restrict__ row, double *__restrict__ d1,
                                      double *__restrict__ hist_row) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 3 * sh.n_shared) {
    const double v = row[i];
    d1[3 * (int64_t)sh.node[i / 3] + (i % 3)] = v;
    if (hist_row) hist_row[i] = v;
  }
}

__global__ void halo_gather_kernel(SharedMap sh, const double *__restrict__ d, double *__restrict__ row) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 3 * sh.n_shared) row[i] = d[3 * (int64_t)sh.node[i / 3] + (i % 3)];
}

// Standalone update on arrays in internal order (backs saa_cd_update).
__global__ void cd_update_kernel(DeviceMesh m, const double *__restrict__ f_int, const double *__restrict__ d0,
                                 const double *__restrict__ dn, double *__restrict__ d1, StepConsts k) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < 3 * (int64_t)m.n_nodes) {
    const int n = (int)(i / 3), c = (int)(i - 3 * (int64_t)n);
    double v = cd_update_dof(f_int[i], m.fext[i], m.mass[i], d0[i], dn[i], k);
    if (m.tag[n] & (1 << c)) v = 0.0;
    d1[i] = v;
  }
}

// internal order -> caller order
__global__ void unpermute_kernel(int n_nodes, const int32_t *__restrict__ new_to_old,
                                 const double *__restrict__ in, double *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < 3 * (int64_t)n_nodes) {
    const int n = (int)(i / 3), c = (int)(i - 3 * (int64_t)n);
    out[3 * (int64_t)new_to_old[n] + c] = in[i];
  }
}

// trajectory recorder of the per-step paths: internal-order state -> one column of the caller's (3n, n_cols) matrix
__global__ void record_column_kernel(int n_nodes, const int32_t *__restrict__ new_to_old, const double *__restrict__ d,
                                     double *__restrict__ traj, int64_t n_cols, int64_t col) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < 3 * (int64_t)n_nodes) {
    const int n = (int)(i / 3), c = (int)(i - 3 * (int64_t)n);
    traj[(3 * (int64_t)new_to_old[n] + c) * n_cols + col] = d[i];
  }
}

// caller order -> internal order
__global__ void permute_kernel(int n_nodes, const int32_t *__restrict__ new_to_old, const double *__restrict__ in,
                               double *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < 3 * (int64_t)n_nodes) {
    const int n = (int)(i / 3), c = (int)(i - 3 * (int64_t)n);
    out[i] = in[3 * (int64_t)new_to_old[n] + c];
  }
}

// ---------------------------------------------------------------------------------------------
// launchers (called from saa_api.cpp through saa_device.h)
// ---------------------------------------------------------------------------------------------
hipError_t configure_kernels(int lds_bytes) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_step_kernel<false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_step_kernel<true>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_step_kernel<false, 0, true>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
}

void launch_fused_step(const DeviceMesh &m, int threads, int lds_bytes, hipStream_t st, const double *d0,
                       const double *dn, double *d1, double *iface, const double *table_row, double *hist_row,
                       const StepConsts &k, const double *tn_dev) {
  hipLaunchKernelGGL(fused_step_kernel<false>, dim3(m.n_blocks), dim3(threads), lds_bytes, st, m, d0, dn, d1,
                     iface, table_row, hist_row, k, static_cast<const PeerMap *>(nullptr), 0u, tn_dev);
}

__global__ void set_scalar_kernel(double *p, double v) { *p = v; }
void launch_set_scalar(hipStream_t st, double *p, double v) {
  hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, st, p, v);
}

hipError_t configure_det_kernels(int lds_bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(&det_items_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             lds_bytes);
}

void launch_det_step(const DeviceMesh &m, const DetLists &det, int threads, int lds_bytes, hipStream_t st, const double *d0,
                     const double *dn, double *out, double *iface, const double *table_row, double *hist_row,
                     const StepConsts &k, bool force_only) {
  hipLaunchKernelGGL(det_items_kernel, dim3(m.n_blocks), dim3(threads), lds_bytes, st, m, d0, det.item_force);
  const unsigned nb = (unsigned)((m.n_nodes + 255) / 256);
  if (force_only)
    hipLaunchKernelGGL(det_nodes_kernel<true>, dim3(nb), dim3(256), 0, st, m, det, d0, dn, out, iface, table_row, hist_row, k);
  else
    hipLaunchKernelGGL(det_nodes_kernel<false>, dim3(nb), dim3(256), 0, st, m, det, d0, dn, out, iface, table_row, hist_row, k);
}

void launch_fused_step_peer(const DeviceMesh &m, int threads, int lds_bytes, hipStream_t st, const double *d0,
                            const double *dn, double *d1, double *hist_row, const StepConsts &k,
                            const PeerMap *pm_dev, unsigned seq) {
  hipLaunchKernelGGL((fused_step_kernel<false, 0, true>), dim3(m.n_blocks), dim3(threads), lds_bytes, st, m, d0, dn,
                     d1, static_cast<double *>(nullptr), static_cast<const double *>(nullptr), hist_row, k, pm_dev, seq,
                     static_cast<const double *>(nullptr));
}

void launch_peer_selftest(const PeerMap &pm, hipStream_t st, const double *own, double *out, unsigned seq) {
  const int n = 3 * pm.n_shared;
  const int blocks = n == 0 ? 1 : ((n + 255) / 256 < 64 ? (n + 255) / 256 : 64);
  hipLaunchKernelGGL(peer_selftest_kernel, dim3(blocks), dim3(256), 0, st, pm, own, out, seq);
}

#ifdef SAA_DIAGNOSTICS
// Diagnostic build only (tools/ablate.py): the step kernel with one phase removed; results are garbage.
void launch_fused_step_ablated(int variant, const DeviceMesh &m, int threads, int lds_bytes, hipStream_t st,
                               const double *d0, const double *dn, double *d1, const StepConsts &k, double *dbg) {
  double *none = nullptr;
  const double *cnone = nullptr;
  // SAA_ABLATE_EXTRA_LDS (bytes): unused dynamic LDS on top of the image, e.g. to leave room for ONE workgroup per CU only
  if (const char *env = getenv("SAA_ABLATE_EXTRA_LDS")) lds_bytes = std::min(160 * 1024, lds_bytes + atoi(env));
#define SAA_ABL(V)                                                                                         \
  case V:                                                                                                  \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_step_kernel<false, V>),                \
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);                      \
    hipLaunchKernelGGL((fused_step_kernel<false, V>), dim3(V == 9 ? 8 * m.n_blocks : m.n_blocks), dim3(threads), lds_bytes, st, m, d0, \
                       dn, d1, none, cnone, V == 8 ? dbg : none, k, static_cast<const PeerMap *>(nullptr), 0u, cnone); \
    break;
  switch (variant) {
    SAA_ABL(0) SAA_ABL(1) SAA_ABL(2) SAA_ABL(3) SAA_ABL(4) SAA_ABL(5) SAA_ABL(6) SAA_ABL(7) SAA_ABL(8) SAA_ABL(9) SAA_ABL(10)
    default: break;
  }
#undef SAA_ABL
}
#endif  // SAA_DIAGNOSTICS

int persistent_lds_bytes(int max_local, int max_owned, int max_items, int max_halo) {
  const long long bytes = 8ll * (persist_off_dn(max_local, max_owned) + 3 * max_owned + 2 * max_owned) + 8ll * max_items +
                          4ll * max_owned + 12ll * max_halo + 16;
  return bytes <= 160 * 1024 ? (int)((bytes + 15) / 16 * 16) : 0;
}

hipError_t configure_persistent_peer(int lds_bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(&persistent_steps_kernel<false, true>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
}

int persistent_max_blocks(int device, int threads, int lds_bytes) {
  int per_cu = 0, cus = 0;
  const void *fns[3] = {reinterpret_cast<const void *>(&persistent_steps_kernel<false, false>),
                        reinterpret_cast<const void *>(&persistent_steps_kernel<true, false>),
                        reinterpret_cast<const void *>(&persistent_steps_kernel<false, true>)};
  for (const void *fn : fns)
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return 0;
  int occ[3] = {0, 0, 0};
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ[0], persistent_steps_kernel<false, false>, threads, lds_bytes) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ[1], persistent_steps_kernel<true, false>, threads, lds_bytes) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ[2], persistent_steps_kernel<false, true>, threads, lds_bytes) != hipSuccess)
    return 0;
  per_cu = occ[0] < occ[1] ? occ[0] : occ[1];
  per_cu = per_cu < occ[2] ? per_cu : occ[2];
  // MI355X_MICROARCH.md "Residency": the occupancy query is one block per CU high when the kernel's SGPR count is in
  // the 81-112 band (the three variants report 102-106).  Waves per SIMD the scalar register file really admits:
  // floor(800 / (ceil(sgpr/16)*16 + 16)) with the band's upper edge, and never more than 8; a block of T threads puts
  // T/256 waves on every SIMD (T >= 256) or a wave on T/64 of the four SIMDs.
  constexpr int kSgprBand = 112, kWavesPerSimd = 800 / (kSgprBand + 16) < 8 ? 800 / (kSgprBand + 16) : 8;
  const int by_sgpr = threads >= 256 ? kWavesPerSimd / (threads / 256) : kWavesPerSimd * (256 / threads);
  per_cu = per_cu < by_sgpr ? per_cu : by_sgpr;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
  return per_cu * cus;
}

// The argument block travels in the kernel-argument segment (captured when the launch is enqueued): nothing on the host
// has to outlive the call, nothing is staged through pageable memory, no second launch.
hipError_t launch_persistent_steps(const DeviceMesh &m, int threads, int lds_bytes, hipStream_t st, const StepConsts &k,
                                   const PersistArgs &a, int mode) {
  static_assert(offsetof(PersistKernArgs, a) == sizeof(DeviceMesh) + sizeof(StepConsts) && alignof(PersistArgs) == 8,
                "kernel-argument segment: the argument block follows the mesh and the step constants without padding");
  // mode 0: plain steps, 1: predicted phase, 2: synchronised steps with the peer exchange.  Plain launches: co-residency
  // of the grid was established by the census launch at set-up (persistent_census), every wait in the kernel is bounded.
  if (mode == 1)
    hipLaunchKernelGGL((persistent_steps_kernel<true, false>), dim3(m.n_blocks), dim3(threads), lds_bytes, st, m, k, a);
  else if (mode == 2)
    hipLaunchKernelGGL((persistent_steps_kernel<false, true>), dim3(m.n_blocks), dim3(threads), lds_bytes, st, m, k, a);
  else
    hipLaunchKernelGGL((persistent_steps_kernel<false, false>), dim3(m.n_blocks), dim3(threads), lds_bytes, st, m, k, a);
  return hipGetLastError();
}

void launch_force_only(const DeviceMesh &m, int threads, int lds_bytes, hipStream_t st, const double *d,
                       double *f) {
  StepConsts k{};
  hipLaunchKernelGGL(fused_step_kernel<true>, dim3(m.n_blocks), dim3(threads), lds_bytes, st, m, d, d, f,
                     static_cast<double *>(nullptr), static_cast<const double *>(nullptr),
                     static_cast<double *>(nullptr), k, static_cast<const PeerMap *>(nullptr), 0u,
                     static_cast<const double *>(nullptr));
}

void launch_iface_finish(const DeviceMesh &m, const SharedMap &sh, hipStream_t st, const double *d0,
                         const double *dn, double *d1, double *iface, double *hist_row, const StepConsts &k,
                         const double *tn_in, double *tn_out) {
  const int n = 3 * (sh.n_shared + sh.n_foreign);
  if (n == 0 && tn_in == nullptr) return;  // (with a device clock the launch is also what advances it)
  hipLaunchKernelGGL(iface_finish_kernel, dim3(((n > 0 ? n : 1) + 255) / 256), dim3(256), 0, st, m, sh, d0, dn, d1, iface,
                     hist_row, k, tn_in, tn_out);
}

void launch_halo_overwrite(const SharedMap &sh, hipStream_t st, const double *row, double *d1, double *hist_row) {
  const int n = 3 * sh.n_shared;
  if (n == 0) return;
  hipLaunchKernelGGL(halo_overwrite_kernel, dim3((n + 255) / 256), dim3(256), 0, st, sh, row, d1, hist_row);
}

void launch_halo_gather(const SharedMap &sh, hipStream_t st, const double *d, double *row) {
  const int n = 3 * sh.n_shared;
  if (n == 0) return;
  hipLaunchKernelGGL(halo_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, st, sh, d, row);
}

void launch_cd_update(const DeviceMesh &m, hipStream_t st, const double *f_int, const double *d0,
                      const double *dn, double *d1, const StepConsts &k) {
  const int64_t n = 3 * (int64_t)m.n_nodes;
  hipLaunchKernelGGL(cd_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m, f_int, d0, dn,
                     d1, k);
}

void launch_record_column(int n_nodes, const int32_t *new_to_old, hipStream_t st, const double *d_internal, double *traj,
                          int64_t n_cols, int64_t col) {
  const int64_t n = 3 * (int64_t)n_nodes;
  hipLaunchKernelGGL(record_column_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n_nodes, new_to_old,
                     d_internal, traj, n_cols, col);
}

void launch_permute(int n_nodes, const int32_t *new_to_old, hipStream_t st, const double *in, double *out) {
  const int64_t n = 3 * (int64_t)n_nodes;
  hipLaunchKernelGGL(permute_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n_nodes, new_to_old, in,
                     out);
}

void launch_unpermute(int n_nodes, const int32_t *new_to_old, hipStream_t st, const double *in, double *out) {
  const int64_t n = 3 * (int64_t)n_nodes;
  hipLaunchKernelGGL(unpermute_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n_nodes,
                     new_to_old, in, out);
}

}  // namespace saa
