// Exact piecewise-linear tabulation of the shape functions, on the device (gfx950).
//
// f_k is a ReLU MLP of a scalar (GNAN.py:24-34), hence piecewise linear; this kernel finds its kinks and
// tabulates it for gnan_fpwl_fwd, once per forward, from the current weights.  One workgroup per feature,
// everything in LDS, float64 arithmetic:
//   1. kinks of the first layer  t_j = -b1_j / w1_j, bitonic-sorted;
//   2. (L = 3) between consecutive kinks every second-layer pre-activation z_j(x) is affine: threads walk the
//      sample points, detect sign changes, append the roots, sort again;
//   3. the network itself is evaluated at the float32-rounded kinks (+ one point beyond each end) and turned
//      into (anchor, value, slope) per piece.
// It replaces ~150 tiny framework launches of the torch restatement of the same procedure
// (gnan_amd/pwl.py:_build_padded, which remains the reference implementation and the L >= 4 path).
//
// L = 3, H <= 64 ("affine" route): between two consecutive FIRST-layer kinks every second-layer pre-activation is affine in x,
// z_j(x) = A[s][j] x + B[s][j], and crossing the kink of unit k changes (A, B) by W2[j, k] (w1_k, b1_k).  The (H + 1) x H forms
// are built incrementally — one direct sum for the leftmost interval, two fmas per unit and kink after it — and BOTH the root
// search of step 2 and the table evaluation of step 3 read z_j(x) off them: O(H) per node instead of the O(H^2) dot products
// that were 37 of the kernel's 50 us on the arxiv shape (LDS-issue bound; profiles/r06 notes in DESIGN_HISTORY R6.6).
// Sorting is by ranking (every element counts the elements before it: broadcast LDS reads, one barrier) instead of a bitonic
// network (21-55 barrier rounds).
#include "common.hpp"
#include "fpwl_bucket.hpp"

#include <cmath>

namespace {

constexpr int kCap = 1024;        // breakpoints per feature this kernel can hold
constexpr int kOverBit = 1 << 30; // on a feature's piece count: the feature ran out of room
// nodes evaluated per pass (p.chunk): 64 while two [chunk, H] float64 tiles fit next to W2 (H <= 64: 64 x 64 / 4 = one
// (two nodes, two units) item per thread), else 32; the root search overlaps consecutive passes by one node
constexpr int kBT = 1024;         // threads of the build workgroup: one workgroup per feature means one wave per SIMD at
                                  // 256 threads, and every LDS round trip of the dot products is exposed; 16 waves hide it

struct BuildParams {
  const float* w_first;  // [F, H]
  const float* b_first;  // [F, H] or null
  const float* w_mid;    // [F, H, H] (L == 3) or null
  const float* b_mid;    // [F, H] or null
  const float* w_last;   // [F, C, H]
  const float* b_last;   // [F, C] or null
  int F, L, H, C;
  int cap;               // pieces - 1 allowed per feature (<= kCap)
  float* anchor;         // [F, cap + 1]
  float* val;            // [F, cap + 1, C]
  float* slope;          // [F, cap + 1, C]
  int32_t* pieces;       // [F]
  int32_t* overflow;     // [1]
  double* scratch;       // [F, cap + 2, C] network values at the table nodes
  int hid_offset;        // byte offset of the two [chunk, H] float64 tiles in dynamic LDS (8-byte aligned)
  int chunk;             // nodes per pass
  int wl_wide;           // 5..64 channels and room in LDS: the last layer's rows staged there for the output sums
  int affine;            // L == 3 by the interval forms: the tile region holds A | B [(H + 1), H] | zt [chunk, H] | k1 [H] | unit_at [H]
};

// Stable merge by ranking of a SORTED run a[0, na) and an unsorted one b[0, nb) into out (a before b on ties, ties inside b by
// index): an element of a keeps its index and adds the elements of b below it, an element of b adds its place in a (binary
// search) to the elements of b before it.  All lanes read the same element of b at the same time (LDS broadcast); out must
// not alias a or b.
__device__ __forceinline__ void rank_merge(const double* a, int na, const double* b, int nb, double* out, int tid) {
  for (int e = tid; e < na + nb; e += kBT) {
    const bool from_a = e < na;
    const int ie = from_a ? e : e - na;
    const double v = from_a ? a[ie] : b[ie];
    int r = ie;
    if (!from_a) {
      int lo = 0, hi = na;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
      }
      r = lo;
    }
    const int tie = from_a ? 0 : ie;                        // elements of b equal to v count if their index is below this
#pragma unroll 16
    for (int i = 0; i < nb; ++i) {
      const double x = b[i];
      r += (x < v || (x == v && i < tie)) ? 1 : 0;
    }
    out[r] = v;
  }
}

// Second-layer pre-activations of a run of nodes (L == 3):  zt[ni, j] = b2_j + sum_k W2[j, k] relu(w1_k x_ni + b1_k).
// The first-layer activations of a node are the same for all H units, so they are computed ONCE per node into h1
// (float64, LDS) and every unit's dot product reads them back: one fma and two LDS reads per term instead of two
// fmas, three float->double conversions and a select (the kernel is bound by float64 issue: 25 + 45 us of its 87 us
// were these dot products).  Same operations in the same order per value as evaluating each unit on its own.
// W2 sits TRANSPOSED in LDS (W2t[k*H + j]): lanes holding consecutive units read consecutive banks, the h1 reads
// are broadcasts.  Ends with a barrier; h1 holds relu(layer 1) and zt the pre-activations of layer 2.
template <typename NodeFn>
__device__ __forceinline__ void eval_nodes(NodeFn node, int n0, int nn, int H, bool three, const float* w1, const float* b1,
                                           const float* b2, const float* W2t, double* h1, double* zt, int tid) {
  for (int it = tid; it < nn * H; it += kBT) {
    const int ni = it / H, kk = it % H;
    const double h = fma(static_cast<double>(w1[kk]), node(n0 + ni), static_cast<double>(b1[kk]));
    h1[it] = h > 0.0 ? h : 0.0;
  }
  __syncthreads();
  if (three && H % 2 == 0) {
    // thread = (two nodes, two units): one 8-byte read of W2t and two broadcast reads of h1 feed four fma chains —
    // 3 LDS instructions per 4 terms instead of 8 (the dot products are bound by the LDS issue rate), and the four
    // independent chains hide the float64 latency.  Each value still sees the same operations in the same order.
    const int H2 = H / 2, nn2 = (nn + 1) / 2;
    for (int it = tid; it < nn2 * H2; it += kBT) {
      const int np = it / H2, jp = it % H2;
      const int na = 2 * np, nb = 2 * np + 1 < nn ? 2 * np + 1 : na;
      const double* ha = h1 + na * H;
      const double* hb = h1 + nb * H;
      const float2* wc = reinterpret_cast<const float2*>(W2t + 2 * jp);   // 8-byte aligned: H is even
      double z00 = b2[2 * jp], z01 = b2[2 * jp + 1], z10 = z00, z11 = z01;
#pragma unroll 8
      for (int kk = 0; kk < H; ++kk) {
        const float2 w = wc[kk * H2];
        const double a = ha[kk], b = hb[kk];
        z00 = fma(static_cast<double>(w.x), a, z00);
        z01 = fma(static_cast<double>(w.y), a, z01);
        z10 = fma(static_cast<double>(w.x), b, z10);
        z11 = fma(static_cast<double>(w.y), b, z11);
      }
      zt[na * H + 2 * jp] = z00;
      zt[na * H + 2 * jp + 1] = z01;
      if (nb != na) {
        zt[nb * H + 2 * jp] = z10;
        zt[nb * H + 2 * jp + 1] = z11;
      }
    }
    __syncthreads();
  } else if (three) {
    for (int it = tid; it < nn * H; it += kBT) {
      const int ni = it / H, j = it % H;
      const double* hrow = h1 + ni * H;
      const float* wcol = W2t + j;
      double z = b2[j];
#pragma unroll 8
      for (int kk = 0; kk < H; ++kk) z = fma(static_cast<double>(wcol[kk * H]), hrow[kk], z);
      zt[it] = z;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(kBT) void pwl_build_kernel(const BuildParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* bp = reinterpret_cast<double*>(smem_raw);        // [kCap]
  double* cand = bp + kCap;                                 // [kCap]
  double* tmp = cand + kCap;                                // [kCap] where the rank sorts put their result
  float* w1 = reinterpret_cast<float*>(tmp + kCap);         // [H]
  float* b1 = w1 + p.H;                                     // [H]
  float* b2 = b1 + p.H;                                     // [H]
  float* wl = b2 + p.H;                                     // [4 * H + 4] last layer's rows and biases of up to four channels
  float* W2 = wl + 4 * p.H + 4;                             // [H*H] (L == 3), transposed: W2[k * H + j]
  float* wl_wide = W2 + (p.L == 3 ? p.H * p.H : 0);        // [C][H + 1] + [C] (5..64 channels: p.wl_wide): rows padded against bank conflicts
  double* h1 = reinterpret_cast<double*>(smem_raw + p.hid_offset);    // [chunk, H] relu(layer 1)
  double* zt = h1 + p.chunk * p.H;                                     // [chunk, H] layer-2 pre-activations
  // the affine route's use of the same region
  const int HS = p.H + 1;                                              // row stride of the forms: lanes = nodes read rows of
  double* A = h1;                                                      // different intervals at the same unit, conflict-free
  double* B = A + (p.H + 1) * HS;                                      // [(H + 1), HS] each
  double* k1 = B + (p.H + 1) * HS + p.chunk * p.H;                     // [H] first-layer kinks, sorted
  int* unit_at = reinterpret_cast<int*>(k1 + p.H);                     // [H] the unit whose kink is the r-th
  if (p.affine) zt = B + (p.H + 1) * HS;
  __shared__ int n_cand, n_bp, over;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int k = blockIdx.x;
  const int H = p.H, C = p.C;
  const double INF = INFINITY;

  for (int i = tid; i < H; i += kBT) {
    w1[i] = p.w_first[k * H + i];
    b1[i] = p.b_first ? p.b_first[k * H + i] : 0.f;
    b2[i] = (p.L == 3 && p.b_mid) ? p.b_mid[k * H + i] : 0.f;
  }
  if (p.wl_wide) {
    for (int i = tid; i < C * H; i += kBT) wl_wide[(i / H) * (H + 1) + i % H] = p.w_last[static_cast<int64_t>(k) * C * H + i];
    if (tid < C) wl_wide[C * (H + 1) + tid] = p.b_last ? p.b_last[k * C + tid] : 0.f;
  }
  if (C <= 4) {                                             // (a global read inside the output sums is a microsecond each)
    for (int i = tid; i < C * H; i += kBT) wl[i] = p.w_last[static_cast<int64_t>(k) * C * H + i];
    if (tid < C) wl[4 * H + tid] = p.b_last ? p.b_last[k * C + tid] : 0.f;
  }
  // W2 (16 KB at H = 64, one cold read per workgroup) is requested now and stored behind step 1, which needs only w1 and b1
  const bool w2_ahead = p.L == 3 && H * H <= 4 * kBT;
  float w2r[4] = {0.f, 0.f, 0.f, 0.f};
  if (w2_ahead) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (tid + q * kBT < H * H) w2r[q] = p.w_mid[static_cast<int64_t>(k) * H * H + tid + q * kBT];
  } else if (p.L == 3) {
    for (int i = tid; i < H * H; i += kBT)       // coalesced read of W[j][kk], transposed write
      W2[(i % H) * H + i / H] = p.w_mid[static_cast<int64_t>(k) * H * H + i];
  }
  if (tid == 0) { n_cand = 0; over = 0; n_bp = 0; }
  __syncthreads();

  // ---- 1. first-layer kinks, sorted by ranking (ties by unit) -----------------------------------
  for (int j = tid; j < H; j += kBT) {
    const double t = w1[j] != 0.f ? -static_cast<double>(b1[j]) / static_cast<double>(w1[j]) : INF;
    cand[j] = isfinite(t) ? t : INF;
  }
  __syncthreads();
  for (int j = tid; j < H; j += kBT) {
    const double t = cand[j];
    if (t < INF) {
      int r = 0;
#pragma unroll 16
      for (int i = 0; i < H; ++i) {
        const double x = cand[i];
        r += (x < t || (x == t && i < j)) ? 1 : 0;
      }
      bp[r] = t;
      if (p.affine) unit_at[r] = j;
      atomicAdd(&n_bp, 1);
    }
  }
  if (w2_ahead) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = tid + q * kBT;
      if (i < H * H) W2[(i % H) * H + i / H] = w2r[q];
    }
  }
  __syncthreads();
  const int P1 = n_bp;                                       // first-layer kinks (finite ones)

  // ---- 2. second-layer kinks ------------------------------------------------------------------
  if (p.L == 3) {
    const int P = P1;
    const double t_first = P ? bp[0] : 0.0, t_last = P ? bp[P - 1] : 0.0;
    const int n_nodes = P + 4;
    auto node = [&](int i) -> double {
      if (i == 0) return t_first - 2.0;
      if (i == 1) return t_first - 1.0;
      if (i < P + 2) return bp[i - 2];
      return i == P + 2 ? t_last + 1.0 : t_last + 2.0;
    };
    auto push = [&](double r) {
      if (isfinite(r)) {
        const int at = atomicAdd(&n_cand, 1);
        if (at < kCap) cand[at] = r; else over = 1;
      }
    };
    // interval (i - 1, i) of the nodes, unit j, with the pre-activations at its two ends
    auto roots = [&](int i, double e_prev, double e, double z_prev, double z) {
      if (z_prev * z < 0.0) push(e_prev + (e - e_prev) * (z_prev / (z_prev - z)));
      if (i == n_nodes - 1) {                               // right ray: extrapolate the outermost affine piece
        const double dr = z - z_prev;
        if (dr != 0.0 && z / dr < 0.0) push(e - z / dr * (e - e_prev));
      }
      if (i == 1) {                                         // left ray
        const double dl = z - z_prev;
        if (dl != 0.0 && z_prev / dl > 0.0) push(e_prev - z_prev / dl * (e - e_prev));
      }
    };
    if (p.affine) {
      // the interval forms: z_j(x) = A[s][j] x + B[s][j] for x between first-layer kinks s - 1 and s.  Left of all kinks the
      // units with w1 < 0 are active (and the constant ones with b1 > 0); crossing the kink of unit k switches it on (w1 > 0)
      // or off.  float x float products are exact in float64: every step is one rounding per form.
      // Two levels over the 16 waves (a chain of 2 H dependent LDS round trips on one wave was 11 of the kernel's 33 us): wave w
      // sums units 4w .. 4w+3 of the leftmost form and the steps of kinks 4w .. 4w+3; the partial sums meet in the tile region
      // and every wave adds what lies before its own steps, in wave order (fixed order: bit-reproducible).
      for (int i = tid; i < P; i += kBT) k1[i] = bp[i];
      double* pa = zt;                                      // [16][H] partial sums of the leftmost form
      double* pb = pa + 16 * H;
      double* sa = pb + 16 * H;                             // [16][H] sums of a wave's steps
      double* sb = sa + 16 * H;
      double la[4] = {0.0, 0.0, 0.0, 0.0}, lb[4] = {0.0, 0.0, 0.0, 0.0};      // running sums of this wave's steps
      if (lane < H) {
        const int j = lane;
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kk = 4 * wave + u;
          if (kk < H) {
            const float w = w1[kk], bb = b1[kk];
            const double w2 = (w < 0.f || (w == 0.f && bb > 0.f)) ? static_cast<double>(W2[kk * H + j]) : 0.0;
            a = fma(w2, static_cast<double>(w), a);
            b = fma(w2, static_cast<double>(bb), b);
          }
        }
        pa[wave * H + j] = a;
        pb[wave * H + j] = b;
        a = 0.0;
        b = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int sk = 4 * wave + u;
          if (sk < P) {
            const int kk = unit_at[sk];
            const double w2 = w1[kk] > 0.f ? static_cast<double>(W2[kk * H + j]) : -static_cast<double>(W2[kk * H + j]);
            a += w2 * static_cast<double>(w1[kk]);          // (exact products: two float32 factors)
            b += w2 * static_cast<double>(b1[kk]);
          }
          la[u] = a;
          lb[u] = b;
        }
        sa[wave * H + j] = a;
        sb[wave * H + j] = b;
      }
      __syncthreads();
      if (lane < H) {
        const int j = lane;
        double a = 0.0, b = b2[j];
#pragma unroll
        for (int w = 0; w < 16; ++w) {
          a += pa[w * H + j];
          b += pb[w * H + j];
        }
        if (wave == 0) {
          A[j] = a;
          B[j] = b;
        }
        for (int w = 0; w < wave; ++w) {
          a += sa[w * H + j];
          b += sb[w * H + j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int sk = 4 * wave + u;
          if (sk < P) {
            A[(sk + 1) * HS + j] = a + la[u];
            B[(sk + 1) * HS + j] = b + lb[u];
          }
        }
      }
      __syncthreads();                                      // (cand is free again: its kinks were read before the forms)
      for (int i = 1 + wave; i < n_nodes; i += kBT / 64) {    // wave = interval (i - 1, i), lane = unit
        const int s = i - 2 < 0 ? 0 : (i - 2 > P ? P : i - 2);
        const double e_prev = node(i - 1), e = node(i);
        if (lane < H) {
          const double a = A[s * HS + lane], b = B[s * HS + lane];
          roots(i, e_prev, e, fma(a, e_prev, b), fma(a, e, b));
        }
      }
    } else {
      // passes of chunk - 1 intervals (chunk nodes, one node of overlap); thread = (interval, unit)
      for (int c0 = 0; c0 < n_nodes - 1; c0 += p.chunk - 1) {
        const int nn = n_nodes - c0 < p.chunk ? n_nodes - c0 : p.chunk;                    // nodes of this pass
        eval_nodes(node, c0, nn, H, true, w1, b1, b2, W2, h1, zt, tid);
        for (int it = tid; it < (nn - 1) * H; it += kBT) {
          const int li = it / H + 1, j = it % H, i = c0 + li;                              // interval (i - 1, i)
          roots(i, node(i - 1), node(i), zt[(li - 1) * H + j], zt[li * H + j]);
        }
        __syncthreads();                                      // the tiles are rewritten by the next chunk
      }
    }
    __syncthreads();
    const int nc = n_cand < kCap ? n_cand : kCap;
    if (P + nc > p.cap) { if (tid == 0) over = 1; }
    const int total = P + nc < kCap ? P + nc : kCap;
    rank_merge(bp, P, cand, total - P, tmp, tid);
    { double* t = bp; bp = tmp; tmp = t; }                  // (nothing reads bp beyond its n_bp entries from here on)
    if (tid == 0) n_bp = total < p.cap ? total : p.cap;
    __syncthreads();
  }

  // ---- 3. table: float32 anchors, float64 network values ----------------------------------------------
  // Anchors are rounded UP: a node goes to piece #{anchors <= x}, so x >= anchor must imply x >= the true kink — a float32
  // x that equals an anchor is then never on the wrong side of a kink lying strictly between two float32 numbers.
  for (int i = tid; i < n_bp; i += kBT) {
    double t = bp[i];
    t = t > 3.0e38 ? 3.0e38 : (t < -3.0e38 ? -3.0e38 : t);
    float f = static_cast<float>(t);
    if (static_cast<double>(f) < t) f = nextafterf(f, INFINITY);
    bp[i] = static_cast<double>(f);
  }
  // Point pieces (pwl.py:_with_point_pieces): behind every anchor a at which some hidden unit's pre-activation is EXACTLY
  // zero — zero biases put every first-layer kink at x = 0 (GNAN.py:49-53), and one-hot / bag-of-words features are mostly
  // exact zeros — nextafter(a) becomes an anchor too: the piece [a, nextafter(a)) holds the nodes with x == a and nothing
  // else, and gnan_fpwl_param_grads differentiates it AT a (relu'(0) = 0, as torch does) instead of inside the piece to the
  // right.  The pre-activations are those of the table evaluation below; only if an anchor was marked is the table
  // evaluated a second time, with the new anchors in place.
  int* on_kink = reinterpret_cast<int*>(cand);              // [kCap] (the candidates of step 2 are in bp by now)
  float* extra = reinterpret_cast<float*>(cand) + kCap;     // [kCap]
  double* V = p.scratch + static_cast<int64_t>(k) * (p.cap + 2) * C;
  const float* Wl = p.w_last + static_cast<int64_t>(k) * C * H;
  for (int round = 0; round < 2; ++round) {
    __syncthreads();
    const int P = n_bp;
    const bool mark = round == 0 && P > 0;
    if (mark) {
      for (int i = tid; i < P; i += kBT) on_kink[i] = 0;
      if (tid == 0) n_cand = 0;
      __syncthreads();
    }
    const double t_first = P ? bp[0] : 0.0, t_last = P ? bp[P - 1] : 0.0;
    auto tnode = [&](int i) -> double {                     // P + 2 nodes (P == 0: -1, 0, +1 with a virtual kink at 0)
      if (i == 0) return t_first - 1.0;
      if (i <= (P ? P : 1)) return P ? bp[i - 1] : 0.0;
      return t_last + 1.0;
    };
    const int Pn = P ? P : 1;                               // table nodes between the two outer ones
    // the network at the table nodes, p.chunk nodes at a time: the last hidden layer into LDS, then (node, channel) pairs
    // take the output dot products
    // affine route, C <= 4: one pass, a thread per node keeps the output sums in registers; more channels: relu(z) goes to the
    // tile chunk by chunk and (node, channel) pairs take the dot products as below
    const bool inline_out = p.affine && C <= 4;
    const int step = inline_out ? Pn + 2 : p.chunk;
    for (int n0 = 0; n0 < Pn + 2; n0 += step) {
      const int nn = Pn + 2 - n0 < step ? Pn + 2 - n0 : step;
      if (inline_out) {
        // eight lanes = one table node, each takes every eighth unit in order and the eight partial sums meet by lane exchange
        // (fixed order).  The loop is bound by float64 issue, not by LDS: a thread per node left 14 of the 16 waves idle for 7 us.
        // The forms' rows are padded, so lanes in different intervals read different banks.
        const int q = tid & 7;
        for (int g0 = 0; g0 < nn; g0 += kBT / 8) {          // (n0 == 0: all nodes; node g is anchor g - 1)
          const int g = g0 + (tid >> 3);
          const bool live = g < nn;
          const double x = tnode(live ? g : 0);
          int s = 0;
          for (int i = q; i < P1; i += 8) s += k1[i] <= x ? 1 : 0;            // its interval: first-layer kinks <= x
          s += __shfl_xor(s, 1);
          s += __shfl_xor(s, 2);
          s += __shfl_xor(s, 4);
          const bool check = mark && g >= 1 && g <= P;
          int zero = 0;
          double acc[4] = {0.0, 0.0, 0.0, 0.0};
          const double* Ar = A + s * HS;
          const double* Br = B + s * HS;
#pragma unroll 4
          for (int j = q; j < H; j += 8) {
            double z = fma(Ar[j], x, Br[j]);
            const float wj = w1[j];                         // (no branch in here: the loads of the next units go out early)
            const double h = fma(static_cast<double>(wj), x, static_cast<double>(b1[j]));
            zero |= ((z == 0.0) | ((wj != 0.f) & (h == 0.0))) ? 1 : 0;
            z = z > 0.0 ? z : 0.0;
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (c < C) acc[c] = fma(static_cast<double>(wl[c * H + j]), z, acc[c]);
          }
#pragma unroll
          for (int m = 1; m < 8; m <<= 1) {
            zero |= __shfl_xor(zero, m);
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (c < C) acc[c] += __shfl_xor(acc[c], m);
          }
          if (live && q == 0) {
            if (zero && check) on_kink[g - 1] = 1;
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (c < C) V[static_cast<int64_t>(g) * C + c] = acc[c] + static_cast<double>(wl[4 * H + c]);
          }
        }
        __syncthreads();
        continue;
      }
      if (p.affine) {
        for (int ni = wave; ni < nn; ni += kBT / 64) {      // wave = node, lane = unit: relu(z) into the tile
          const int g = n0 + ni;
          const double x = tnode(g);
          const int s = __popcll(__ballot(lane < P1 && k1[lane] <= x));
          bool zero = false;
          if (lane < H) {
            double z = fma(A[s * HS + lane], x, B[s * HS + lane]);
            if (mark && g >= 1 && g <= P)
              zero = z == 0.0 || (w1[lane] != 0.f && fma(static_cast<double>(w1[lane]), x, static_cast<double>(b1[lane])) == 0.0);
            zt[ni * H + lane] = z > 0.0 ? z : 0.0;
          }
          if (mark && __ballot(zero) != 0 && lane == 0) on_kink[g - 1] = 1;
        }
        __syncthreads();
      } else {
        eval_nodes(tnode, n0, nn, H, p.L == 3, w1, b1, b2, W2, h1, zt, tid);
        if (mark) {
          for (int it = tid; it < nn * H; it += kBT) {
            const int g = n0 + it / H, j = it % H;
            if (g < 1 || g > P) continue;
            bool zero = w1[j] != 0.f && fma(static_cast<double>(w1[j]), bp[g - 1], static_cast<double>(b1[j])) == 0.0;
            if (p.L == 3) zero |= zt[it] == 0.0;
            if (zero) on_kink[g - 1] = 1;
          }
        }
      }
      const double* hid = h1;                               // L == 2: relu(layer 1) is the last hidden layer
      if (p.L == 3) {
        if (!p.affine) {
          if (mark) __syncthreads();
          for (int it = tid; it < nn * H; it += kBT) zt[it] = zt[it] > 0.0 ? zt[it] : 0.0;
          __syncthreads();
        }
        hid = zt;
      }
      if (p.wl_wide) {                                     // (uniform) the last layer's rows from LDS
        for (int it = tid; it < nn * C; it += kBT) {
          const int ni = it / C, c = it % C;
          double acc = static_cast<double>(wl_wide[C * (H + 1) + c]);
          const float* wr = wl_wide + c * (H + 1);
          const double* hr = hid + ni * H;
#pragma unroll 8
          for (int j = 0; j < H; ++j) acc = fma(static_cast<double>(wr[j]), hr[j], acc);
          V[static_cast<int64_t>(n0 + ni) * C + c] = acc;
        }
      } else {
        for (int it = tid; it < nn * C; it += kBT) {
          const int ni = it / C, c = it % C;
          double acc = p.b_last ? static_cast<double>(p.b_last[k * C + c]) : 0.0;
#pragma unroll 8
          for (int j = 0; j < H; ++j) acc = fma(static_cast<double>(Wl[c * H + j]), hid[ni * H + j], acc);
          V[static_cast<int64_t>(n0 + ni) * C + c] = acc;
        }
      }
      __syncthreads();
    }
    if (!mark) break;
    // behind the last of a run of coinciding anchors only, and only where the next anchor does not already end the piece there
    for (int i = tid; i < P; i += kBT) {
      if (!on_kink[i]) continue;
      const float up = nextafterf(static_cast<float>(bp[i]), INFINITY);
      if (i == P - 1 || static_cast<float>(bp[i + 1]) > up) extra[atomicAdd(&n_cand, 1)] = up;
    }
    __syncthreads();
    const int nf = n_cand;
    if (nf == 0) break;
    if (P + nf > p.cap) {                                   // no room: the host drops the tables (overflow) anyway
      if (tid == 0) over = 1;
      break;
    }
    for (int i = tid; i < nf; i += kBT) bp[P + i] = static_cast<double>(extra[i]);     // (P + nf <= cap <= kCap)
    __syncthreads();
    rank_merge(bp, P, bp + P, nf, tmp, tid);
    { double* t = bp; bp = tmp; tmp = t; }
    if (tid == 0) n_bp = P + nf;
  }
  __syncthreads();
  const int P = n_bp;
  const double t_first = P ? bp[0] : 0.0, t_last = P ? bp[P - 1] : 0.0;
  auto tnode = [&](int i) -> double {
    if (i == 0) return t_first - 1.0;
    if (i <= (P ? P : 1)) return P ? bp[i - 1] : 0.0;
    return t_last + 1.0;
  };
  __threadfence_block();
  __syncthreads();
  const int pieces = P + 1;
  float* A_out = p.anchor + static_cast<int64_t>(k) * (p.cap + 1);
  float* VL = p.val + static_cast<int64_t>(k) * (p.cap + 1) * C;
  float* SL = p.slope + static_cast<int64_t>(k) * (p.cap + 1) * C;
  for (int i = tid; i < pieces; i += kBT) {
    // piece i lies between table nodes i and i+1; it is anchored at its left kink, piece 0 at the first kink
    const int an = i == 0 ? 1 : i;
    A_out[i] = static_cast<float>(tnode(an));
    const double width = tnode(i + 1) - tnode(i);
    for (int c = 0; c < C; ++c) {
      VL[i * C + c] = static_cast<float>(V[static_cast<int64_t>(an) * C + c]);
      double s;
      if (P == 0) {
        s = V[2 * C + c] - V[1 * C + c];                    // affine: f(1) - f(0)
      } else {
        s = width > 0.0 ? (V[static_cast<int64_t>(i + 1) * C + c] - V[static_cast<int64_t>(i) * C + c]) / width : 0.0;
      }
      SL[i * C + c] = static_cast<float>(s);
    }
  }
  // (the overflow flag rides on the feature's piece count and is gathered by pwl_compact_kernel: an atomicOr here needed the
  // flag zeroed by a launch of its own before every build)
  if (tid == 0) p.pieces[k] = pieces | (over ? kOverBit : 0);
}

// Pack the per-feature padded tables back to back: off[k] = pieces[0] + ... + pieces[k-1].
struct CompactParams {
  const float* anchor_p;   // padded [F, cap+1]
  const float* val_p;      // padded [F, cap+1, C]
  const float* slope_p;
  const int32_t* pieces;   // per feature, | kOverBit where the feature ran out of room
  int F, C, cap;
  int32_t* off;            // [F+1]
  int32_t* overflow;       // [1]: 1 if any feature ran out of room, else 0 (written by the last feature's workgroup)
  float* anchor;           // compact [T], capacity F*(cap+1)
  float* val;
  float* slope;
  const float* index_range;   // optional: the look-up's bucket tables out of this pass (csrc/fpwl_bucket.hpp)
  uint16_t* index_table;
  float* index_key;
};

template <int LOGB>
__global__ __launch_bounds__(256) void pwl_compact_kernel(const CompactParams p) {
  __shared__ int red[256];
  const int k = blockIdx.x, tid = threadIdx.x;
  int s = 0, over = 0;
  for (int j = tid; j < k; j += 256) {
    const int v = p.pieces[j];
    s += v & ~kOverBit;
    over |= v & kOverBit;
  }
  red[tid] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) red[tid] += red[tid + st];
    __syncthreads();
  }
  const int base = red[0];
  const int own = p.pieces[k];
  const int n = own & ~kOverBit;
  const int any_over = __syncthreads_or(over | (own & kOverBit));
  if (tid == 0) {
    p.off[k] = base;
    if (k == p.F - 1) {
      p.off[p.F] = base + n;
      *p.overflow = any_over ? 1 : 0;
    }
  }
  const float* a = p.anchor_p + static_cast<int64_t>(k) * (p.cap + 1);
  const float* v = p.val_p + static_cast<int64_t>(k) * (p.cap + 1) * p.C;
  const float* sl = p.slope_p + static_cast<int64_t>(k) * (p.cap + 1) * p.C;
  for (int i = tid; i < n; i += 256) p.anchor[base + i] = a[i];
  for (int i = tid; i < n * p.C; i += 256) {
    p.val[static_cast<int64_t>(base) * p.C + i] = v[i];
    p.slope[static_cast<int64_t>(base) * p.C + i] = sl[i];
  }
  if constexpr (LOGB > 0) {            // the feature's bucket table, from its anchors where they lie (index_build_kernel otherwise)
    __syncthreads();
    gnan_index::build_bucket_index<LOGB>(k, a, n - 1, p.index_range, p.index_table, p.index_key, nullptr);
  }
}

size_t padded_floats(int F, int C, int cap) { return static_cast<size_t>(F) * (cap + 1) * (1 + 2 * static_cast<size_t>(C)); }

}  // namespace

extern "C" size_t gnan_pwl_build_scratch_bytes(int32_t F, int32_t C, int32_t cap) {
  // float64 network values at the table nodes | padded per-feature tables | pieces per feature
  return static_cast<size_t>(F) * (cap + 2) * C * sizeof(double) + padded_floats(F, C, cap) * sizeof(float) +
         static_cast<size_t>(F) * sizeof(int32_t);
}

extern "C" int gnan_pwl_build(const gnan_pwl_build_args* a, gnan_stream_t stream) {
  GNAN_REQUIRE(a != nullptr, "pwl_build: null args");
  GNAN_REQUIRE(a->F >= 1 && a->C >= 1 && a->H >= 1, "pwl_build: bad sizes");
  if (a->L != 2 && a->L != 3) return gnan::fail(GNAN_ERR_UNSUPPORTED, "pwl_build: kernel covers L in {2, 3} (got %d)", a->L);
  if (a->H > 128) return gnan::fail(GNAN_ERR_UNSUPPORTED, "pwl_build: hidden width %d > 128", a->H);
  GNAN_REQUIRE(a->cap >= 1 && a->cap <= kCap, "pwl_build: cap must be in [1, %d]", kCap);
  GNAN_REQUIRE(a->w_first && a->w_last && a->anchor && a->val && a->slope && a->off && a->overflow && a->scratch,
               "pwl_build: null pointer");
  if (a->L == 3) GNAN_REQUIRE(a->w_mid != nullptr, "pwl_build: L == 3 needs w_mid");
  GNAN_REQUIRE((a->index_table == nullptr && a->index_key == nullptr && a->index_range == nullptr) ||
                   (a->index_table && a->index_key && a->index_range &&
                    (a->index_buckets == 256 || a->index_buckets == 512 || a->index_buckets == 1024 || a->index_buckets == 2048)),
               "pwl_build: index_range / index_table / index_key go together, index_buckets in {256, 512, 1024, 2048}");
  if (a->scratch_bytes < gnan_pwl_build_scratch_bytes(a->F, a->C, a->cap))
    return gnan::fail(GNAN_ERR_WORKSPACE, "pwl_build: scratch too small");
  BuildParams p;
  p.w_first = a->w_first; p.b_first = a->b_first; p.w_mid = a->w_mid; p.b_mid = a->b_mid;
  p.w_last = a->w_last; p.b_last = a->b_last;
  p.F = a->F; p.L = a->L; p.H = a->H; p.C = a->C; p.cap = a->cap;
  p.overflow = a->overflow;
  p.scratch = static_cast<double*>(a->scratch);
  float* padded = reinterpret_cast<float*>(p.scratch + static_cast<size_t>(a->F) * (a->cap + 2) * a->C);
  p.anchor = padded;
  p.val = p.anchor + static_cast<size_t>(a->F) * (a->cap + 1);
  p.slope = p.val + static_cast<size_t>(a->F) * (a->cap + 1) * a->C;
  p.pieces = reinterpret_cast<int32_t*>(p.slope + static_cast<size_t>(a->F) * (a->cap + 1) * a->C);
  size_t lds = 3 * kCap * sizeof(double) + (7 * static_cast<size_t>(a->H) + 4 + (a->L == 3 ? static_cast<size_t>(a->H) * a->H : 0)) * sizeof(float);
  // 5..64 channels: the last layer's rows (padded) behind W2, if everything still fits 160 KB
  const size_t wide_floats = static_cast<size_t>(a->C) * (a->H + 1) + a->C;
  const size_t tiles_guess = (a->L == 3 && a->H <= 64)
                                 ? (2 * static_cast<size_t>(a->H + 1) * (a->H + 1) + 64 * static_cast<size_t>(a->H) + a->H) * sizeof(double) + a->H * sizeof(int)
                                 : 2 * static_cast<size_t>(a->H <= 64 ? 64 : 32) * a->H * sizeof(double);
  p.wl_wide = (a->C > 4 && a->C <= 64 && lds + wide_floats * sizeof(float) + 8 + tiles_guess + 64 <= 160 * 1024) ? 1 : 0;
  if (p.wl_wide) lds += wide_floats * sizeof(float);
  lds = (lds + 7) & ~static_cast<size_t>(7);
  p.hid_offset = static_cast<int>(lds);
  p.chunk = a->H <= 64 ? 64 : 32;
  p.affine = (a->L == 3 && a->H <= 64) ? 1 : 0;           // (H + 1) x H interval forms of 16 bytes: 67 KB at H = 64
  const size_t H_ = static_cast<size_t>(a->H);
  if (p.affine) lds += (2 * (H_ + 1) * (H_ + 1) + p.chunk * H_ + H_) * sizeof(double) + H_ * sizeof(int);
  else lds += 2 * static_cast<size_t>(p.chunk) * a->H * sizeof(double);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pwl_build_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return gnan::fail(GNAN_ERR_HIP, "pwl_build: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(pwl_build_kernel, dim3(a->F), dim3(kBT), lds, st, p);
  if (int rc = gnan::check_launch("pwl_build_kernel")) return rc;
  CompactParams c;
  c.anchor_p = p.anchor; c.val_p = p.val; c.slope_p = p.slope; c.pieces = p.pieces;
  c.F = a->F; c.C = a->C; c.cap = a->cap;
  c.off = a->off; c.overflow = a->overflow; c.anchor = a->anchor; c.val = a->val; c.slope = a->slope;
  c.index_range = a->index_range; c.index_table = a->index_table; c.index_key = a->index_key;
  const dim3 grid(a->F), block(256);
  if (a->index_table == nullptr) hipLaunchKernelGGL(pwl_compact_kernel<0>, grid, block, 0, st, c);
  else if (a->index_buckets == 256) hipLaunchKernelGGL(pwl_compact_kernel<8>, grid, block, 0, st, c);
  else if (a->index_buckets == 512) hipLaunchKernelGGL(pwl_compact_kernel<9>, grid, block, 0, st, c);
  else if (a->index_buckets == 1024) hipLaunchKernelGGL(pwl_compact_kernel<10>, grid, block, 0, st, c);
  else hipLaunchKernelGGL(pwl_compact_kernel<11>, grid, block, 0, st, c);
  return gnan::check_launch("pwl_compact_kernel");
}

namespace {
// Do the tables just built (off[F + 1] | overflow, as gnan_pwl_build left them on the device) fit a look-up that was sized
// for max_pieces per feature and max_group_pieces per group of fg features?  flag[0] = 1 if not (never cleared here).
__global__ __launch_bounds__(256) void pwl_check_fit_kernel(const int32_t* __restrict__ meta, int F, int fg, int max_pieces,
                                                            int max_group_pieces, float* __restrict__ flag) {
  __shared__ int bad[256];
  int b = 0;
  for (int k = threadIdx.x; k < F; k += 256) {
    if (meta[k + 1] - meta[k] > max_pieces) b = 1;
    if (k % fg == 0) {
      const int hi = k + fg < F ? k + fg : F;
      if (meta[hi] - meta[k] > max_group_pieces) b = 1;
    }
  }
  if (threadIdx.x == 0 && meta[F + 1] != 0) b = 1;          // the build itself ran out of room
  bad[threadIdx.x] = b;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (static_cast<int>(threadIdx.x) < st) bad[threadIdx.x] |= bad[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0 && bad[0]) flag[0] = 1.f;
}
}  // namespace

extern "C" int gnan_pwl_check_fit(const int32_t* meta, int32_t F, int32_t features_per_group, int32_t max_pieces,
                                  int32_t max_group_pieces, float* flag, gnan_stream_t stream) {
  GNAN_REQUIRE(meta != nullptr && flag != nullptr && F >= 1 && features_per_group >= 1, "pwl_check_fit: bad arguments");
  hipLaunchKernelGGL(pwl_check_fit_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), meta, F, features_per_group,
                     max_pieces, max_group_pieces, flag);
  return gnan::check_launch("pwl_check_fit_kernel");
}
