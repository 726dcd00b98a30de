// Parameter gradients of the shape functions from the per-piece moments (gfx950): the last step of the table path's
// backward pass, i.e. of autograd through GNAN.py:57-62 w.r.t. the f_k parameters (trainer.py:66).
//
// On piece t of feature k (anchor a) the network is affine, f(x) = f(a) + s (x - a), with a fixed activation pattern
// (D1, D2 = 0/1 masks of the two hidden layers at any interior point of the piece), so the loss of all nodes that
// fell into the piece depends on the parameters only through  <M0, f(a)> + <M1, s>  (M0 = sum g, M1 = sum g (x - a):
// gnan_fpwl_moments).  With  h1a = D1 (w1 a + b1),  h1' = D1 w1,  h2a = D2 (W2 h1a + b2),  h2' = D2 W2 h1'
// (value and x-derivative of the hidden layers at the anchor) the exact gradient of that expression is
//     dW3 = M0 (x) h2a + M1 (x) h2'        db3 = M0
//     e0 = D2 W3^T M0,  e1 = D2 W3^T M1    dW2 = e0 (x) h1a + e1 (x) h1'     db2 = e0
//     q0 = D1 W2^T e0,  q1 = D1 W2^T e1    dw1 = q0 a + q1                   db1 = q0
// summed over the pieces of the feature — one reverse pass for the value, one for the slope, sharing the masks.
// One workgroup per feature walks its pieces (empty ones are skipped), float64 throughout, every gradient element
// owned by one thread: no atomics, bit-reproducible.  It replaces ~110 tiny framework launches of the torch
// restatement of the same sum (gnan_amd/pwl.py:parameter_grads_from_moments, two probe points per piece through the
// batched MLP in float64 — still the route for L >= 4 and the reference this kernel is tested against).
//
// gnan_fpwl_moment_scales: the two power-of-two scales of the fixed-point moments, from max|grad| and max|x - anchor|,
// in one pass + one single-thread kernel instead of ~20 framework launches.
#include "common.hpp"

#include <cfloat>
#include <cmath>

namespace {

constexpr int kBI = 16;   // elements of a 4-lane split: ceil(64 / 4)

struct GradParams {
  const int32_t* off;
  const float* anchor;
  const float* M;         // [T, 2, C] float32, or null
  const int64_t* Mi;      // [T, 2, C] fixed point, or null
  const double* scales;   // [2] (with Mi)
  const float* w1;
  const float* b1;
  const float* W2;
  const float* b2;
  const float* Wl;
  const float* bl;
  int F, L, H, C;
  float* d_w1;
  float* d_b1;
  float* d_W2;
  float* d_b2;
  float* d_Wl;
  float* d_bl;
};

__device__ __forceinline__ double quad_sum(double v) {   // the 4 lanes of a quad are adjacent: fixed butterfly
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  return v;
}

// moments of piece t -> Mv[0..C) = M0, Mv[C..2C) = M1 (float64, LDS); returns whether any of them is non-zero
__device__ __forceinline__ bool piece_moment(const GradParams& p, int64_t t, int c2, double inv0, double inv1, double* val) {
  const int C = p.C;
  if (p.Mi) {
    const int64_t v = p.Mi[t * 2 * C + c2];
    *val = static_cast<double>(v) * (c2 < C ? inv0 : inv1);
    return v != 0;
  }
  const float v = p.M[t * 2 * C + c2];
  *val = static_cast<double>(v);
  return v != 0.f;
}

// anchor of piece li (of P) and a point strictly inside it: piece 0 is the ray left of the first kink (anchored at that
// kink), the last piece the ray right of the last kink; a zero-width piece (coinciding kinks) holds no node
// A piece one float32 step wide, [a, nextafter(a)), holds the nodes with x == a and nothing else (the table builder puts one
// behind every anchor on which a hidden unit's pre-activation is exactly zero, pwl_build.hip): its masks are taken AT a,
// strictly (z > 0), which is torch's relu'(0) = 0 — with zero biases (GNAN.py:49-53) and one-hot features that is every
// unit of most nodes, and the masks of the piece to the right would hand their bias gradients to the wrong units.
__device__ __forceinline__ void piece_points(const float* A, int li, int P, double* a, double* xi) {
  *a = static_cast<double>(A[li]);
  if (li == 0) *xi = *a - 1.0;
  else if (li == P - 1) *xi = *a + 1.0;
  else if (A[li + 1] <= nextafterf(A[li], INFINITY)) *xi = *a;
  else *xi = 0.5 * (*a + static_cast<double>(A[li + 1]));
}

// live[0 .. *n_live) = the pieces of [base, base + P) that hold at least one node (a non-zero moment), ascending: one parallel
// pass over the moments and an ordered compaction by ballots (a single thread walking P flags in LDS cost ~5 us of the 64 the
// arxiv shape's launch took).  Called by all kThreads threads of the workgroup; the caller's barrier publishes the list.
template <int kThreads>
__device__ __forceinline__ void list_live_pieces(const GradParams& p, int base, int P, double inv0, double inv1, int* live,
                                                 int* wave_live, int* n_live) {
  constexpr int kWaves = kThreads / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, C = p.C;
  int running = 0;
  for (int l0 = 0; l0 < P; l0 += kThreads) {
    const int li = l0 + tid;
    bool nz = false;
    if (li < P) {
      double unused;
      for (int c2 = 0; c2 < 2 * C; ++c2) nz |= piece_moment(p, base + li, c2, inv0, inv1, &unused);
    }
    const unsigned long long b = __ballot(nz);
    if (lane == 0) wave_live[wave] = __popcll(b);
    __syncthreads();
    int before = running, all = running;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const int c = wave_live[w];
      before += w < wave ? c : 0;
      all += c;
    }
    if (nz) live[before + __popcll(b & ((1ull << lane) - 1ull))] = li;
    running = all;
    __syncthreads();                                // wave_live is rewritten by the next chunk
  }
  if (tid == 0) *n_live = running;
}

// ---- L == 3, H <= 64: NG groups of 256 threads walk the pieces NG at a time; in a group thread (j, ib) = (unit of
// layer 2, quarter of the layer-1 units).  One group per workgroup needed 110 us on the arxiv shape (129 features x ~130
// pieces, two barriers per piece, 129 workgroups on 256 CUs); several groups share the weights in LDS and a piece's
// latency, the live pieces are listed by ballots and a piece's moments, anchor and inner point are requested a round
// ahead: 110 -> 64 -> 50 us (measured; two and four groups alike — what is left is the ~130 LDS reads and float64 fmas
// per thread and piece).  The groups' accumulators meet in LDS in group order (fixed order: bit-reproducible).
// CQ = ceil(C / 4) channel accumulators per thread: a template parameter so that the common one-channel case does not pay
// 32 registers for them.  1024 threads leave 128 VGPRs per lane: W2 is read from its padded LDS copy (conflict-free: row
// stride H + 1) rather than kept in 16 registers, which spilled 86 at four groups; CQ = 16 spills 120 there and runs two.
template <int kNG, int CQ>
__global__ __launch_bounds__(256 * kNG) void fpwl_grad3_kernel(const GradParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int H = p.H, C = p.C, HS = H + 1;
  float* w1 = reinterpret_cast<float*>(smem_raw);
  float* b1 = w1 + H;
  float* b2 = b1 + H;
  float* W2s = b2 + H;                 // [H][HS]: W2[j][i], padded rows (column reads in the W2^T products)
  float* Wl = W2s + H * HS;            // [C][H]
  double* vec = reinterpret_cast<double*>(smem_raw + (((3 * H + H * HS + C * H) * sizeof(float) + 7) & ~size_t(7)));
  const int tid = threadIdx.x, k = blockIdx.x;
  const int grp = tid >> 8, lt = tid & 255;
  double* h1i = vec + grp * (5 * H + 2 * C);      // per group: h1i | h1a | h1p | e0 | e1 | Mv[2C]
  double* h1a = h1i + H;
  double* h1p = h1a + H;
  double* e0 = h1p + H;
  double* e1 = e0 + H;
  double* Mv = e1 + H;
  double* red = vec + kNG * (5 * H + 2 * C);      // [kNG - 1][256]: one accumulator of every thread of groups 1.. at a time
  int* live = reinterpret_cast<int*>(red + (kNG - 1) * 256);   // [P]: the pieces that hold at least one node, ascending
  __shared__ int n_live;
  __shared__ int wave_live[4 * kNG];
  const int j = lt >> 2, ib = lt & 3;
  const int BI = (H + 3) >> 2;
  const int64_t kH = static_cast<int64_t>(k) * H;

  for (int i = tid; i < H; i += 256 * kNG) {
    w1[i] = p.w1[kH + i];
    b1[i] = p.b1 ? p.b1[kH + i] : 0.f;
    b2[i] = p.b2 ? p.b2[kH + i] : 0.f;
  }
  for (int i = tid; i < H * H; i += 256 * kNG) W2s[(i / H) * HS + i % H] = p.W2[kH * H + i];
  for (int i = tid; i < C * H; i += 256 * kNG) Wl[i] = p.Wl[kH * C + i];
  double dW2[kBI], dW3[CQ];
#pragma unroll
  for (int r = 0; r < kBI; ++r) dW2[r] = 0.0;
#pragma unroll
  for (int r = 0; r < CQ; ++r) dW3[r] = 0.0;
  double db2 = 0.0, dw1 = 0.0, db1 = 0.0, db3 = 0.0;
  const int base = p.off[k], P = p.off[k + 1] - base;
  const double inv0 = p.Mi ? 1.0 / p.scales[0] : 1.0, inv1 = p.Mi ? 1.0 / p.scales[1] : 1.0;
  // Most pieces hold no node (the kinks of a feature spread far beyond the range of its values): list the others first —
  // one parallel pass over the moments instead of a dependent global read and a barrier per piece (the kernel spent its
  // time there: 110 us on the arxiv shape with ~130 pieces of which ~25 are live).
  list_live_pieces<256 * kNG>(p, base, P, inv0, inv1, live, wave_live, &n_live);
  __syncthreads();
  const int nl = n_live;
  double mval = 0.0;                                // this thread's moment of the group's NEXT piece, requested a round ahead
  double a_next = 0.0, xi_next = 0.0;               // ... and its anchor and inner point (two dependent global reads otherwise)
  if (grp < nl) {
    if (lt < 2 * C) piece_moment(p, base + live[grp], lt, inv0, inv1, &mval);
    piece_points(p.anchor + base, live[grp], P, &a_next, &xi_next);
  }

  for (int l0 = 0; l0 < nl; l0 += kNG) {
    const int at = l0 + grp;                        // this group's piece of the round (a group beyond the list: zeros)
    const int li = at < nl ? live[at] : -1;
    __syncthreads();                                // the previous round's readers of the LDS vectors are done
    const double a = a_next, xi = xi_next;          // (a group beyond the list: zero moments, any point)
    if (lt < 2 * C) Mv[lt] = li >= 0 ? mval : 0.0;
    mval = 0.0;
    if (at + kNG < nl) {
      const int ln = live[at + kNG];
      if (lt < 2 * C) piece_moment(p, base + ln, lt, inv0, inv1, &mval);
      piece_points(p.anchor + base, ln, P, &a_next, &xi_next);
    }
    if (lt < H) {
      const double wv = static_cast<double>(w1[lt]), bv = static_cast<double>(b1[lt]);
      const double z = fma(wv, xi, bv);
      const bool on = z > 0.0;
      h1i[lt] = on ? z : 0.0;
      h1a[lt] = on ? fma(wv, a, bv) : 0.0;
      h1p[lt] = on ? wv : 0.0;
    }
    __syncthreads();
    if (j < H) {
      double si = 0.0, sa = 0.0, sp = 0.0;
#pragma unroll
      for (int r = 0; r < kBI; ++r) {
        const int i = ib * BI + r;
        if (r < BI && i < H) {
          const double w = static_cast<double>(W2s[j * HS + i]);
          si = fma(w, h1i[i], si);
          sa = fma(w, h1a[i], sa);
          sp = fma(w, h1p[i], sp);
        }
      }
      si = quad_sum(si); sa = quad_sum(sa); sp = quad_sum(sp);
      const double bj = static_cast<double>(b2[j]);
      const bool on2 = si + bj > 0.0;
      const double h2a = on2 ? sa + bj : 0.0, h2p = on2 ? sp : 0.0;
      double p0 = 0.0, p1 = 0.0;
      for (int c = ib; c < C; c += 4) {
        const double wl = static_cast<double>(Wl[c * H + j]);
        p0 = fma(wl, Mv[c], p0);
        p1 = fma(wl, Mv[C + c], p1);
      }
      p0 = quad_sum(p0); p1 = quad_sum(p1);
      const double e0v = on2 ? p0 : 0.0, e1v = on2 ? p1 : 0.0;
#pragma unroll
      for (int r = 0; r < kBI; ++r) {
        const int i = ib * BI + r;
        if (r < BI && i < H) dW2[r] = fma(e0v, h1a[i], fma(e1v, h1p[i], dW2[r]));
      }
#pragma unroll
      for (int r = 0; r < CQ; ++r) {
        const int c = ib + 4 * r;
        if (c < C) dW3[r] = fma(Mv[c], h2a, fma(Mv[C + c], h2p, dW3[r]));
      }
      if (ib == 0) {
        db2 += e0v;
        e0[j] = e0v;
        e1[j] = e1v;
      }
    }
    if (lt < C) db3 += Mv[lt];
    __syncthreads();
    if (j < H) {                        // same quad, other role: i = j, the quad splits the layer-2 units
      const int i = j;
      double q0 = 0.0, q1 = 0.0;
#pragma unroll
      for (int r = 0; r < kBI; ++r) {
        const int jj = ib * BI + r;
        if (r < BI && jj < H) {
          const double w = static_cast<double>(W2s[jj * HS + i]);
          q0 = fma(w, e0[jj], q0);
          q1 = fma(w, e1[jj], q1);
        }
      }
      q0 = quad_sum(q0); q1 = quad_sum(q1);
      if (ib == 0 && fma(static_cast<double>(w1[i]), xi, static_cast<double>(b1[i])) > 0.0) {
        dw1 += fma(q0, a, q1);
        db1 += q0;
      }
    }
    // the next round's first barrier separates these reads of e0 / e1 from their next writes
  }

  // groups 1 .. NG-1 hand their accumulators to group 0, one value per thread at a time; group 0 adds them in group order
  auto gather = [&](double& v) {
    __syncthreads();
    if (grp > 0) red[(grp - 1) * 256 + lt] = v;
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int g = 1; g < kNG; ++g) v += red[(g - 1) * 256 + lt];
    }
  };
#pragma unroll
  for (int r = 0; r < kBI; ++r)
    if (r < BI) gather(dW2[r]);
#pragma unroll
  for (int r = 0; r < CQ; ++r)
    if (r * 4 < C) gather(dW3[r]);
  gather(db2); gather(dw1); gather(db1); gather(db3);
  if (grp != 0) return;

  if (j < H) {
#pragma unroll
    for (int r = 0; r < kBI; ++r) {
      const int i = ib * BI + r;
      if (r < BI && i < H) p.d_W2[(kH + j) * H + i] = static_cast<float>(dW2[r]);
    }
#pragma unroll
    for (int r = 0; r < CQ; ++r) {
      const int c = ib + 4 * r;
      if (c < C) p.d_Wl[(static_cast<int64_t>(k) * C + c) * H + j] = static_cast<float>(dW3[r]);
    }
    if (ib == 0) {
      if (p.d_b2) p.d_b2[kH + j] = static_cast<float>(db2);
      p.d_w1[kH + j] = static_cast<float>(dw1);
      if (p.d_b1) p.d_b1[kH + j] = static_cast<float>(db1);
    }
  }
  if (lt < C && p.d_bl) p.d_bl[static_cast<int64_t>(k) * C + lt] = static_cast<float>(db3);
}

// ---- L == 3, H <= 64, C <= 4, at most 1024 pieces per feature: the same sums in O(live pieces x H + H^2) ------------------
// Along the pieces of a feature (ascending x) every first-layer unit switches ONCE: its mask at a piece's inner point,
// fma(w1_i, xi, b1_i) > 0, is monotone in xi.  So
//   * the second layer's pre-activation at piece n is an affine form  z_j(x) = Aj x + Bj  that changes by W2[j, i] (w1_i, b1_i)
//     when unit i switches — one wave (lane = unit j of layer 2) sweeps the live pieces in order, keeping (Aj, Bj) and the
//     masks D2, e0, e1, h2a, h2' of the derivation above without a single dot product over the first layer;
//   * the sums over pieces that carry a first-layer mask,  SG[j][i] = sum_n D1_n[i] (e0_n[j] a_n + e1_n[j])  and
//     SE[j][i] = sum_n D1_n[i] e0_n[j],  are prefix sums of the sweep taken when unit i switches (or the total minus them), and
//         dW2[j][i] = w1_i SG[j][i] + b1_i SE[j][i]      dw1[i] = sum_j W2[j, i] SG[j][i]      db1[i] = sum_j W2[j, i] SE[j][i]
//     come out of them once per feature.
// fpwl_grad3_kernel spends three mat-vecs and a rank-2 update of H x H per live piece (7 us per round of four pieces on the arxiv
// shape, LDS-issue bound: 50 us); this one ~0.1 us per piece and ~5 us per feature.  Fixed orders: bit-reproducible.
constexpr int kSweepPieces = 1024;     // live pieces whose points fit LDS
constexpr int kSweepChunk = 128;       // live pieces whose moments are staged at a time

__global__ __launch_bounds__(256) void fpwl_grad3_sweep_kernel(const GradParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int H = p.H, C = p.C, HS = H + 1;
  double* xi_s = reinterpret_cast<double*>(smem_raw);          // [kSweepPieces] inner point of live piece n
  double* an_s = xi_s + kSweepPieces;                           // [kSweepPieces] its anchor
  double* mom = an_s + kSweepPieces;                            // [kSweepChunk][8]: M0[0..4) | M1[0..4) of a chunk's pieces
  double* PG = mom + kSweepChunk * 8;                           // [H][HS] prefix of G at unit i's switch, per unit j
  double* PE = PG + H * HS;                                     // [H][HS]
  double* TG = PE + H * HS;                                     // [H] totals
  double* TE = TG + H;                                          // [H]
  float* w1 = reinterpret_cast<float*>(TE + H);
  float* b1 = w1 + H;
  float* b2 = b1 + H;
  float* Wl = b2 + H;                  // [4][H]
  float* W2s = Wl + 4 * H;             // [H][HS]: W2[j][i], padded rows
  int* bnd = reinterpret_cast<int*>(W2s + H * HS);              // [H] live pieces before unit i's switch
  int* order = bnd + H;                                         // [H] units by (bnd, index)
  int* sw_bnd = order + H;                                      // [H + 1] bnd of the r-th switch (sentinel behind the last)
  float* sw_w = reinterpret_cast<float*>(sw_bnd + H + 1);       // [H] signed factors of the r-th switch
  float* sw_b = sw_w + H;                                       // [H]
  int* live = reinterpret_cast<int*>(sw_b + H);                 // [P]
  __shared__ int n_live;
  __shared__ int wave_live[4];
  const int tid = threadIdx.x, k = blockIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t kH = static_cast<int64_t>(k) * H;

  for (int i = tid; i < H; i += 256) {
    w1[i] = p.w1[kH + i];
    b1[i] = p.b1 ? p.b1[kH + i] : 0.f;
    b2[i] = p.b2 ? p.b2[kH + i] : 0.f;
  }
  for (int i = tid; i < H * H; i += 256) W2s[(i / H) * HS + i % H] = p.W2[kH * H + i];
  for (int i = tid; i < C * H; i += 256) Wl[i] = p.Wl[kH * C + i];
  const int base = p.off[k], P = p.off[k + 1] - base;
  const double inv0 = p.Mi ? 1.0 / p.scales[0] : 1.0, inv1 = p.Mi ? 1.0 / p.scales[1] : 1.0;
  list_live_pieces<256>(p, base, P, inv0, inv1, live, wave_live, &n_live);
  __syncthreads();
  const int nl = n_live;
  for (int n = tid; n < nl; n += 256) piece_points(p.anchor + base, live[n], P, &an_s[n], &xi_s[n]);
  __syncthreads();
  // where each first-layer unit switches: units with w1 < 0 (and the constant ones with b1 > 0) are on for the first bnd
  // live pieces, the others off for the first bnd
  if (tid < H) {
    const double w = static_cast<double>(w1[tid]), b = static_cast<double>(b1[tid]);
    int on = 0;
#pragma unroll 8
    for (int n = 0; n < nl; ++n) on += fma(w, xi_s[n], b) > 0.0 ? 1 : 0;
    const bool starts_on = w1[tid] < 0.f || (w1[tid] == 0.f && b1[tid] > 0.f);
    bnd[tid] = starts_on ? on : nl - on;
  }
  __syncthreads();
  if (tid < H) {
    const int mine = bnd[tid];
    int r = 0;
#pragma unroll 8
    for (int i = 0; i < H; ++i) {
      const int o = bnd[i];
      r += (o < mine || (o == mine && i < tid)) ? 1 : 0;
    }
    order[r] = tid;
  }
  __syncthreads();

  // ---- the sweep: wave 0, lane = unit j of layer 2; all waves stage the moments of a chunk --------------------------------
  // the form left of all switches: quad = unit j, its four lanes split the first-layer units (fixed butterfly); handed to wave 0
  // through TG / TE, which hold nothing yet
  {
    const int jq = tid >> 2, ibq = tid & 3, BIq = (H + 3) >> 2;
    double a0 = 0.0, b0 = 0.0;
#pragma unroll
    for (int r = 0; r < kBI; ++r) {
      const int i = ibq * BIq + r;
      const int ic = (r < BIq && i < H) ? i : 0;
      const float w = w1[ic], b = b1[ic];
      const bool use = r < BIq && i < H && jq < H && (w < 0.f || (w == 0.f && b > 0.f));
      const double w2v = static_cast<double>(W2s[(jq < H ? jq : 0) * HS + ic]);
      const double w2 = use ? w2v : 0.0;
      a0 = fma(w2, static_cast<double>(w), a0);
      b0 = fma(w2, static_cast<double>(b), b0);
    }
    a0 = quad_sum(a0);
    b0 = quad_sum(b0);
    if (ibq == 0 && jq < H) {
      TG[jq] = a0;
      TE[jq] = b0 + static_cast<double>(b2[jq]);
    }
  }
  // the switches in sweep order with what they add to the form's coefficients' factors: one level of LDS reads in the sweep
  if (tid < H) {
    const int i = order[tid];
    sw_bnd[tid] = bnd[i];
    sw_w[tid] = w1[i] > 0.f ? w1[i] : -w1[i];             // |w1|: switching on adds W2 w1 (w1 > 0), switching off subtracts W2 w1 (w1 < 0)
    sw_b[tid] = w1[i] > 0.f ? b1[i] : -b1[i];
  }
  __syncthreads();
  const int j = lane;
  const bool jl = wave == 0 && j < H;
  double Aj = jl ? TG[j] : 0.0, Bj = jl ? TE[j] : 0.0;
  double RG = 0.0, RE = 0.0, db2 = 0.0;
  double dW3[4] = {0.0, 0.0, 0.0, 0.0}, db3[4] = {0.0, 0.0, 0.0, 0.0};
  float wl[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) wl[c] = (jl && c < C) ? Wl[c * H + j] : 0.f;
  int cursor = 0;
  // the next switch, requested one ahead: its position, unit and this lane's W2 entry
  int nx_bnd = sw_bnd[0], nx_i = order[0];
  double nx_w2 = jl ? static_cast<double>(W2s[j * HS + nx_i]) : 0.0;
  double nx_w = static_cast<double>(sw_w[0]), nx_b = static_cast<double>(sw_b[0]);
  for (int c0 = 0; c0 < nl; c0 += kSweepChunk) {
    const int cn = nl - c0 < kSweepChunk ? nl - c0 : kSweepChunk;
    __syncthreads();                                  // the previous chunk's moments have been read
    for (int it = tid; it < cn * 8; it += 256) {
      const int n = it >> 3, c2 = it & 7;
      const int c = c2 & 3;
      double v = 0.0;
      if (c < C) piece_moment(p, base + live[c0 + n], (c2 >> 2) * C + c, inv0, inv1, &v);
      mom[it] = v;
    }
    __syncthreads();
    if (jl) {
      for (int n = 0; n < cn; ++n) {
        const int g = c0 + n;
        const double x = xi_s[g], a = an_s[g];
        double m0[4], m1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          m0[c] = mom[n * 8 + c];                     // (zero beyond C)
          m1[c] = mom[n * 8 + 4 + c];
        }
        while (cursor < H && nx_bnd == g) {           // units that switch between live pieces g - 1 and g
          PG[j * HS + nx_i] = RG;
          PE[j * HS + nx_i] = RE;
          Aj += nx_w2 * nx_w;                         // (exact products of two float32 factors)
          Bj += nx_w2 * nx_b;
          ++cursor;
          const int cc = cursor < H ? cursor : H - 1;
          nx_bnd = cursor < H ? sw_bnd[cc] : -1;
          nx_i = order[cc];
          nx_w2 = static_cast<double>(W2s[j * HS + nx_i]);
          nx_w = static_cast<double>(sw_w[cc]);
          nx_b = static_cast<double>(sw_b[cc]);
        }
        const bool on2 = fma(Aj, x, Bj) > 0.0;
        const double h2a = on2 ? fma(Aj, a, Bj) : 0.0, h2p = on2 ? Aj : 0.0;
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          p0 = fma(static_cast<double>(wl[c]), m0[c], p0);
          p1 = fma(static_cast<double>(wl[c]), m1[c], p1);
          dW3[c] = fma(m0[c], h2a, fma(m1[c], h2p, dW3[c]));
          db3[c] += m0[c];
        }
        const double e0 = on2 ? p0 : 0.0, e1 = on2 ? p1 : 0.0;
        db2 += e0;
        RG += fma(e0, a, e1);
        RE += e0;
      }
    }
  }
  if (jl) {
    for (; cursor < H; ++cursor) {                    // units that never switch within the live pieces
      const int i = order[cursor];
      PG[j * HS + i] = RG;
      PE[j * HS + i] = RE;
    }
    TG[j] = RG;
    TE[j] = RE;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < C) p.d_Wl[(static_cast<int64_t>(k) * C + c) * H + j] = static_cast<float>(dW3[c]);
    if (p.d_b2) p.d_b2[kH + j] = static_cast<float>(db2);
    if (j == 0 && p.d_bl) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < C) p.d_bl[static_cast<int64_t>(k) * C + c] = static_cast<float>(db3[c]);
    }
  }
  __syncthreads();

  // ---- once per feature: the H x H outputs from the prefix sums ------------------------------------------------------------
  // a unit that starts on carries the pieces BEFORE its switch (the prefix), the others the pieces after it (total - prefix)
  const int u = tid >> 2, ib = tid & 3;               // quad = one unit; its four lanes split the other index
  const int BI = (H + 3) >> 2;
  if (u < H) {
    // row j = u of dW2 (loads unconditional at clamped indices: a guarded load waits for everything before it)
    const double tg = TG[u], te = TE[u];
#pragma unroll 4
    for (int r = 0; r < kBI; ++r) {
      const int i = ib * BI + r;
      const bool ok = r < BI && i < H;
      const int ic = ok ? i : 0;
      const float w = w1[ic], b = b1[ic];
      const double pg = PG[u * HS + ic], pe = PE[u * HS + ic];
      const bool starts_on = w < 0.f || (w == 0.f && b > 0.f);
      const double sg = starts_on ? pg : tg - pg;
      const double se = starts_on ? pe : te - pe;
      if (ok) p.d_W2[(kH + u) * H + i] = static_cast<float>(fma(static_cast<double>(w), sg, static_cast<double>(b) * se));
    }
    // column i = u: dw1, db1
    const bool starts_on = w1[u] < 0.f || (w1[u] == 0.f && b1[u] > 0.f);
    double q0 = 0.0, q1 = 0.0;
#pragma unroll 4
    for (int r = 0; r < kBI; ++r) {
      const int jj = ib * BI + r;
      const bool ok = r < BI && jj < H;
      const int jc = ok ? jj : 0;
      const double wv = static_cast<double>(W2s[jc * HS + u]);
      const double w = ok ? wv : 0.0;
      const double pg = PG[jc * HS + u], pe = PE[jc * HS + u];
      const double sg = starts_on ? pg : TG[jc] - pg;
      const double se = starts_on ? pe : TE[jc] - pe;
      q1 = fma(w, sg, q1);
      q0 = fma(w, se, q0);
    }
    q0 = quad_sum(q0);
    q1 = quad_sum(q1);
    if (ib == 0) {
      p.d_w1[kH + u] = static_cast<float>(q1);
      if (p.d_b1) p.d_b1[kH + u] = static_cast<float>(q0);
    }
  }
}

// ---- L == 2, H <= 128: thread (i, cb) = (hidden unit, quarter of the output channels) --------------------------------
__global__ __launch_bounds__(512) void fpwl_grad2_kernel(const GradParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ double Mv[128];
  __shared__ int n_live;
  __shared__ int wave_live[8];
  int* pieces = reinterpret_cast<int*>(smem_raw);   // [P]: the pieces that hold at least one node
  const int H = p.H, C = p.C;
  const int tid = threadIdx.x, k = blockIdx.x;
  const int i = tid >> 2, cb = tid & 3;
  const int64_t kH = static_cast<int64_t>(k) * H;
  const bool live = i < H;
  const double wv = live ? static_cast<double>(p.w1[kH + i]) : 0.0;
  const double bv = (live && p.b1) ? static_cast<double>(p.b1[kH + i]) : 0.0;
  float wlr[kBI];
  double dWl[kBI];
#pragma unroll
  for (int r = 0; r < kBI; ++r) {
    const int c = cb + 4 * r;
    wlr[r] = (live && c < C) ? p.Wl[(static_cast<int64_t>(k) * C + c) * H + i] : 0.f;
    dWl[r] = 0.0;
  }
  double dw1 = 0.0, db1 = 0.0, db3 = 0.0;
  const int base = p.off[k], P = p.off[k + 1] - base;
  const double inv0 = p.Mi ? 1.0 / p.scales[0] : 1.0, inv1 = p.Mi ? 1.0 / p.scales[1] : 1.0;
  list_live_pieces<512>(p, base, P, inv0, inv1, pieces, wave_live, &n_live);   // (see fpwl_grad3_kernel)
  __syncthreads();
  const int nl = n_live;
  double mval = 0.0;
  if (nl > 0 && tid < 2 * C) piece_moment(p, base + pieces[0], tid, inv0, inv1, &mval);
  for (int at = 0; at < nl; ++at) {
    const int li = pieces[at];
    __syncthreads();                                 // the previous piece's readers of Mv are done
    double a, xi;
    piece_points(p.anchor + base, li, P, &a, &xi);
    if (tid < 2 * C) {
      Mv[tid] = mval;
      if (at + 1 < nl) piece_moment(p, base + pieces[at + 1], tid, inv0, inv1, &mval);
    }
    __syncthreads();
    const bool on = live && fma(wv, xi, bv) > 0.0;
    const double h1a = on ? fma(wv, a, bv) : 0.0, h1p = on ? wv : 0.0;
    double p0 = 0.0, p1 = 0.0;
#pragma unroll
    for (int r = 0; r < kBI; ++r) {
      const int c = cb + 4 * r;
      if (c < C) {
        const double wl = static_cast<double>(wlr[r]);
        p0 = fma(wl, Mv[c], p0);
        p1 = fma(wl, Mv[C + c], p1);
        dWl[r] = fma(Mv[c], h1a, fma(Mv[C + c], h1p, dWl[r]));
      }
    }
    p0 = quad_sum(p0); p1 = quad_sum(p1);
    if (on && cb == 0) {
      dw1 += fma(p0, a, p1);
      db1 += p0;
    }
    if (tid < C) db3 += Mv[tid];
  }
  if (live) {
#pragma unroll
    for (int r = 0; r < kBI; ++r) {
      const int c = cb + 4 * r;
      if (c < C) p.d_Wl[(static_cast<int64_t>(k) * C + c) * H + i] = static_cast<float>(dWl[r]);
    }
    if (cb == 0) {
      p.d_w1[kH + i] = static_cast<float>(dw1);
      if (p.d_b1) p.d_b1[kH + i] = static_cast<float>(db1);
    }
  }
  if (tid < C && p.d_bl) p.d_bl[static_cast<int64_t>(k) * C + tid] = static_cast<float>(db3);
}

// ---- scales of the fixed-point moments --------------------------------------------------------------------------------
// blk[2 b] = max |grad| over workgroup b's share of the [n, width] gradient, blk[2 b + 1] = max |anchor| over its share of
// the T anchors, as the bit patterns of non-negative floats (their order is the order of the values; a NaN ends up on top
// and poisons the scales, as it would poison the sums).  Every workgroup writes its own pair — no atomics, nothing to zero
// first (a zeroing launch and its 8-byte target used to precede this one) — and, on the way, clears the moment accumulators
// the next launch adds into (`zero`, int64 words: a framework fill launch otherwise).
__device__ __forceinline__ void finish_scales(const unsigned* blk, int n_blk, const double* x_abs_max, int nbits, double* scales);

__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ g, int64_t n, int width, int64_t stride,
                                                     const float* __restrict__ anchor, int64_t T, const int32_t* off_end,
                                                     unsigned* blk, unsigned long long* zero, int64_t zero_words,
                                                     unsigned* arrive, const double* x_abs_max, int nbits, double* scales) {
  const int64_t total = n * width;
  if (off_end && *off_end < T) T = *off_end;       // tables held in a buffer of full capacity: only off[F] anchors are real
  float m = 0.f, ma = 0.f;
  const int64_t step = static_cast<int64_t>(gridDim.x) * 256;
  int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (width == stride) {                         // dense rows: eight loads in flight per thread (one at a time: 22 us for 40 MB)
    for (; e + 7 * step < total; e += 8 * step) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = fabsf(g[e + u * step]);
#pragma unroll
      for (int u = 0; u < 8; ++u) m = (t[u] > m || t[u] != t[u]) ? t[u] : m;
    }
  }
  for (; e < total; e += step) {
    const float v = fabsf(width == stride ? g[e] : g[(e / width) * stride + e % width]);
    m = (v > m || v != v) ? v : m;
  }
  for (int64_t a = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; a < T; a += step) {
    const float v = fabsf(anchor[a]);
    ma = (v > ma || v != v) ? v : ma;
  }
  for (int64_t z = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; z < zero_words; z += step) zero[z] = 0ull;
  unsigned u = __float_as_uint(m), ua = __float_as_uint(ma);
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned o = __shfl_xor(u, off), oa = __shfl_xor(ua, off);
    u = o > u ? o : u;
    ua = oa > ua ? oa : ua;
  }
  __shared__ unsigned wmax[4][2];
  if ((threadIdx.x & 63) == 0) { wmax[threadIdx.x >> 6][0] = u; wmax[threadIdx.x >> 6][1] = ua; }
  __syncthreads();
  if (threadIdx.x < 2) {
    unsigned m4 = wmax[0][threadIdx.x];
    for (int w = 1; w < 4; ++w) m4 = wmax[w][threadIdx.x] > m4 ? wmax[w][threadIdx.x] : m4;
    blk[2 * blockIdx.x + threadIdx.x] = m4;
  }
  // with an arrival counter the last workgroup of the pass takes the scales itself (scales_kernel otherwise: a launch of 5 us)
  if (arrive != nullptr && gnan::last_block(arrive)) finish_scales(blk, static_cast<int>(gridDim.x), x_abs_max, nbits, scales);
}

// one workgroup: the maxima of the n_blk pairs, then the two scales
__device__ __forceinline__ void finish_scales(const unsigned* blk, int n_blk, const double* x_abs_max, int nbits, double* scales) {
  unsigned u = 0u, ua = 0u;
  for (int b = threadIdx.x; b < n_blk; b += 256) {
    const unsigned v = blk[2 * b], va = blk[2 * b + 1];
    u = v > u ? v : u;
    ua = va > ua ? va : ua;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned o = __shfl_xor(u, off), oa = __shfl_xor(ua, off);
    u = o > u ? o : u;
    ua = oa > ua ? oa : ua;
  }
  __shared__ unsigned wmax2[4][2];
  if ((threadIdx.x & 63) == 0) { wmax2[threadIdx.x >> 6][0] = u; wmax2[threadIdx.x >> 6][1] = ua; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  for (int w = 1; w < 4; ++w) {
    u = wmax2[w][0] > u ? wmax2[w][0] : u;
    ua = wmax2[w][1] > ua ? wmax2[w][1] : ua;
  }
  const float gf = __uint_as_float(u), af = __uint_as_float(ua);
  const double xm = *x_abs_max;
  if (gf != gf || af != af || xm != xm) {            // a NaN in the gradient poisons the sums, as it would in float
    scales[0] = scales[1] = static_cast<double>(NAN);
    return;
  }
  nbits = nbits < 50 ? nbits : 50;                   // fpwl_moments_c1_kernel converts terms exactly only below 2^51
  const double g = fmax(static_cast<double>(gf), DBL_MIN);
  const double d = fmax(xm + static_cast<double>(af), DBL_MIN);
  const double e0 = fmin(fmax(floor(nbits - log2(g)), -1000.0), 1000.0);
  const double e1 = fmin(fmax(floor(nbits - log2(g * d)), -1000.0), 1000.0);
  scales[0] = ldexp(1.0, static_cast<int>(e0));
  scales[1] = ldexp(1.0, static_cast<int>(e1));
}


__global__ __launch_bounds__(256) void scales_kernel(const unsigned* blk, int n_blk, const double* x_abs_max, int nbits,
                                                     double* scales) {
  finish_scales(blk, n_blk, x_abs_max, nbits, scales);
}

}  // namespace

extern "C" int gnan_fpwl_param_grads(const gnan_fpwl_grad_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr, "fpwl_param_grads: null args");
  GNAN_REQUIRE(a->F >= 1 && a->H >= 1 && a->C >= 1, "fpwl_param_grads: bad sizes");
  if (a->L != 2 && a->L != 3) return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl_param_grads: kernel covers L in {2, 3} (got %d)", a->L);
  if (a->C > 64 || a->H > (a->L == 3 ? 64 : 128))
    return gnan::fail(GNAN_ERR_UNSUPPORTED, "fpwl_param_grads: H <= %d and C <= 64 (got H=%d, C=%d)", a->L == 3 ? 64 : 128, a->H, a->C);
  GNAN_REQUIRE(a->off && a->anchor && a->w_first && a->w_last && a->d_w_first && a->d_w_last, "fpwl_param_grads: null pointer");
  GNAN_REQUIRE(a->max_pieces >= 1 && a->max_pieces <= 8192, "fpwl_param_grads: max_pieces must be in [1, 8192] (got %d)", a->max_pieces);
  GNAN_REQUIRE((a->moments != nullptr) != (a->moments_fixed != nullptr), "fpwl_param_grads: exactly one of moments / moments_fixed");
  GNAN_REQUIRE(a->moments_fixed == nullptr || a->scales != nullptr, "fpwl_param_grads: moments_fixed needs scales");
  if (a->L == 3) GNAN_REQUIRE(a->w_mid && a->d_w_mid, "fpwl_param_grads: L == 3 needs w_mid / d_w_mid");
  GNAN_REQUIRE((a->b_first == nullptr) == (a->d_b_first == nullptr) && (a->b_last == nullptr) == (a->d_b_last == nullptr) &&
               (a->L == 2 || (a->b_mid == nullptr) == (a->d_b_mid == nullptr)),
               "fpwl_param_grads: d_b_* must be NULL exactly where the bias is NULL");
  GradParams p;
  p.off = a->off; p.anchor = a->anchor; p.M = a->moments; p.Mi = a->moments_fixed; p.scales = a->scales;
  p.w1 = a->w_first; p.b1 = a->b_first; p.W2 = a->w_mid; p.b2 = a->b_mid; p.Wl = a->w_last; p.bl = a->b_last;
  p.F = a->F; p.L = a->L; p.H = a->H; p.C = a->C;
  p.d_w1 = a->d_w_first; p.d_b1 = a->d_b_first; p.d_W2 = a->d_w_mid; p.d_b2 = a->d_b_mid;
  p.d_Wl = a->d_w_last; p.d_bl = a->d_b_last;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (a->L == 3) {
    const size_t H = a->H, C = a->C;
    size_t lds = ((3 * H + H * (H + 1) + C * H) * sizeof(float) + 7) & ~size_t(7);
    // few channels: four groups (1024 threads, 128 registers per lane); many: two (256 registers — CQ = 16 spilled 120 at four)
    // (measured on the arxiv shape, 128 features x ~130 pieces: 50 us with either group count — the pieces' LDS reads and
    // float64 fmas bound it, not the barriers per round; four groups halve the rounds of features with few live pieces)
    if (C <= 4 && a->max_pieces <= kSweepPieces) {
      const size_t HS = H + 1;
      const size_t sweep = (2 * kSweepPieces + kSweepChunk * 8 + 2 * H * HS + 2 * H) * sizeof(double) +
                           (3 * H + 4 * H + H * HS) * sizeof(float) + (5 * H + 1 + static_cast<size_t>(a->max_pieces) + 8) * sizeof(int);
      if (sweep > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fpwl_grad3_sweep_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(sweep));
        if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "fpwl_param_grads: hipFuncSetAttribute: %s", hipGetErrorString(e));
      }
      hipLaunchKernelGGL(fpwl_grad3_sweep_kernel, dim3(a->F), dim3(256), sweep, st, p);
      return gnan::check_launch("fpwl_grad3_sweep_kernel");
    }
    const int ng = C <= 4 ? 4 : 2;
    lds += (ng * (5 * H + 2 * C) + (ng - 1) * 256) * sizeof(double) + (static_cast<size_t>(a->max_pieces) + 8) * sizeof(int);   // + live pieces
    const dim3 grid(a->F), block(256 * ng);
    if (C <= 4) hipLaunchKernelGGL((fpwl_grad3_kernel<4, 1>), grid, block, lds, st, p);
    else if (C <= 8) hipLaunchKernelGGL((fpwl_grad3_kernel<2, 2>), grid, block, lds, st, p);
    else if (C <= 16) hipLaunchKernelGGL((fpwl_grad3_kernel<2, 4>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((fpwl_grad3_kernel<2, 16>), grid, block, lds, st, p);
    return gnan::check_launch("fpwl_grad3_kernel");
  }
  hipLaunchKernelGGL(fpwl_grad2_kernel, dim3(a->F), dim3(512), (static_cast<size_t>(a->max_pieces) + 8) * sizeof(int), st, p);
  return gnan::check_launch("fpwl_grad2_kernel");
}

extern "C" int gnan_fpwl_moment_scales(const gnan_moment_scales_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr, "fpwl_moment_scales: null args");
  const int64_t n = a->n, T = a->T;
  GNAN_REQUIRE(n >= 0 && a->width >= 1 && a->grad_stride >= a->width && T >= 0, "fpwl_moment_scales: bad sizes");
  GNAN_REQUIRE((a->grad || n == 0) && (a->anchor || T == 0) && a->x_abs_max && a->scales, "fpwl_moment_scales: null pointer");
  GNAN_REQUIRE(a->workspace && a->workspace_bytes >= GNAN_MOMENT_SCALES_WORKSPACE_BYTES,
               "fpwl_moment_scales: workspace of GNAN_MOMENT_SCALES_WORKSPACE_BYTES bytes needed");
  GNAN_REQUIRE(a->bits >= 1 && a->bits <= 62, "fpwl_moment_scales: bits must be in [1, 62]");
  GNAN_REQUIRE(a->zero_bytes == 0 || (a->zero && a->zero_bytes % 8 == 0 && reinterpret_cast<uintptr_t>(a->zero) % 8 == 0),
               "fpwl_moment_scales: zero must be 8-byte aligned, zero_bytes a multiple of 8");
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned* blk = static_cast<unsigned*>(a->workspace);
  const int64_t zero_words = static_cast<int64_t>(a->zero_bytes / 8);
  int64_t work = n * a->width > T ? n * a->width : T;
  work = work > zero_words ? work : zero_words;
  int64_t blocks = (work + 256 * 8 - 1) / (256 * 8);
  constexpr int64_t kMaxBlocks = GNAN_MOMENT_SCALES_WORKSPACE_BYTES / 8;
  blocks = blocks < 1 ? 1 : (blocks > kMaxBlocks ? kMaxBlocks : blocks);
  hipLaunchKernelGGL(absmax_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, a->grad, n, a->width, a->grad_stride,
                     a->anchor, T, a->n_anchors, blk, static_cast<unsigned long long*>(a->zero), zero_words,
                     reinterpret_cast<unsigned*>(a->arrive_counter), a->x_abs_max, a->bits, a->scales);
  if (int rc = gnan::check_launch("absmax_kernel")) return rc;
  if (a->arrive_counter != nullptr) return GNAN_OK;
  hipLaunchKernelGGL(scales_kernel, dim3(1), dim3(256), 0, st, blk, static_cast<int>(blocks), a->x_abs_max, a->bits, a->scales);
  return gnan::check_launch("scales_kernel");
}
