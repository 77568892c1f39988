// Matrix-core variant of the fused Helmholtz operator: 3D, P = 12 (p = 11),
// fp32, scalar fields, affine / multilinear elements (BASELINE config 5).
//
//   out = mask * scatter( lambda0 * B_loc(g) + lambda1 * A_loc(g) ),  g = gather(u)
//
// Same operator and reference call sites as helmholtz_kernel (sfem_helmholtz.h:
// core/interpolation.py:288-292 applied per element, core/fespace.py:471 its
// transpose, examples/poisson.py:141-154, navier_stokes.py:431).  At p = 11 the
// vector ALU, not HBM, bounds that kernel (profiles/r02_p11_f32_counters.json:
// the VALU pipe is ~79 % busy; plain v_fma_f32 runs at half the packed fp32
// rate), and the six 12x12 contractions are 40 % of its instructions.  Here they
// run on the matrix cores instead:
//
//   Y = D X  with  D (12 x 12, padded to 16 rows)  and  X (12 x 144)
//
// one v_mfma_f32_16x16x4_f32 per (16-column tile, 4-deep k step): 9 x 3 = 27
// MFMAs per contraction, 162 per element, exact fp32 arithmetic (the fma chain
// of the dense product, k ascending).  The contraction axis is chosen by how
// the B operand is read from the element's LDS copy X[a][i][j]:
//
//   axis 0: k = a, column c = (i, j)      address (4s + q) 144 + c
//   axis 1: k = i, column c = (a, j)      address a 144 + (4s + q) 12 + j
//   axis 2: k = j, column c = (a, i)      address c 12 + 4s + q
//
// (lane l: q = l >> 4 the k row of the B tile, m = l & 15 its column).  A result
// tile comes back with its column on the lane and rows 4q + r in 4 registers
// (MI355X guide, "Fragment layout"), i.e. at OTHER points than any one input
// lane holds, and differently for the three axes; the three gradients of a
// point therefore meet again in LDS: the forward MFMAs first (results in
// registers), then the results overwrite X and two more arrays at their
// natural positions, the point-wise geometry runs in place, and the
// transposed products go the same way back.
//
// One element = one workgroup of FOUR waves (3 x 6.75 KB of LDS, 7 elements =
// 28 waves per CU): the 27 tile chains of a product stage are dealt out to the
// waves (waves 0-2: tiles 0..6 of "their" axis, wave 3: tiles 7, 8 of all
// three), so the four SIMDs' matrix pipes work on one element at once and each
// wave's dependent chain is a quarter as long.  (A first version with one
// wave per element ran 162 MFMAs + 4 k other instructions in sequence on 7
// waves per CU: 1.9 ms against 0.99 ms for the vector-ALU kernel at 48^3.)
//
// MEASURED (MI355X, 48^3 elements, profiles/r02_mfma_notes.md): correct to fp32
// parity, 41 % fewer vector instructions per element (2462 vs 4203) and the
// matrix pipes 20 % busy -- but 1.25 ms against 1.0 ms for the vector-ALU kernel:
// the results of a product stage reach the next stage only through LDS (406 vs
// 288 LDS instructions per element) and six workgroup barriers per element
// leave the waves waiting (issue stalls 3x).  With the LDS accesses of each
// stage batched (all loads, then all stores): 1.19 ms affine, 1.31 vs 1.34 ms
// on multilinear elements (128-VGPR budget there).  Opt-in with SFEM_MFMA=1.
//
// Everything outside the MFMAs uses ONE thread-to-point mapping: wave w, lane l
// works on the points (a, c) with a = 3w .. 3w+2 and c = l + 64 r2 (r2 < 3,
// c < 144).  The weights of (i, j) = c are then three per-lane constants and
// w[a] comes from a 12-entry table: no index arithmetic per point.
#pragma once
#include "sfem_helmholtz.h"

#ifndef SFEM_MFMA_MINW
#define SFEM_MFMA_MINW 6
#endif
namespace sfem {

template <int P>
struct MfmaConsts {
  float d[P * P];   // D[row][col], row-major: (D x)[r] = sum_c D[r][c] x[c]
  float w[P];       // 1D quadrature weights
  float x[P];       // 1D nodes
};

typedef float mfma_f4 __attribute__((ext_vector_type(4)));

// B-operand address of axis AX for k step s, tile t (floats)
template <int AX>
__device__ __forceinline__ int mfma_b_addr(int q, int m, int s, int t) {
  constexpr int P = 12, PP = 144;
  if (AX == 0) return (4 * s + q) * PP + 16 * t + m;
  if (AX == 2) return (16 * t + m) * P + 4 * s + q;
  const int c = 16 * t + m;
  return (c / P) * PP + (4 * s + q) * P + (c % P);
}
// natural position of result row 4q + r of tile t
template <int AX>
__device__ __forceinline__ int mfma_o_addr(int q, int m, int r, int t) {
  constexpr int P = 12, PP = 144;
  if (AX == 0) return (4 * q + r) * PP + 16 * t + m;
  if (AX == 2) return (16 * t + m) * P + 4 * q + r;
  const int c = 16 * t + m;
  return (c / P) * PP + (4 * q + r) * P + (c % P);
}

// tiles T0 .. T1-1 of axis AX: products into `res`, NR = result slots used
template <int AX, int T0, int T1, int OFF>
__device__ __forceinline__ void mfma_tiles(const float (&aM)[3],
                                           const float* src, int q, int m,
                                           mfma_f4 (&res)[7]) {
  // all B operands first (one LDS round trip), then the MFMA chains
  float b[T1 - T0][3];
#pragma unroll
  for (int t = T0; t < T1; ++t)
#pragma unroll
    for (int s = 0; s < 3; ++s) b[t - T0][s] = src[mfma_b_addr<AX>(q, m, s, t)];
#pragma unroll
  for (int t = T0; t < T1; ++t) {
    mfma_f4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 3; ++s)
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(aM[s], b[t - T0][s], c, 0, 0,
                                               0);
    res[OFF + t - T0] = c;
  }
}
template <int AX, int T0, int T1, int OFF>
__device__ __forceinline__ void mfma_store(float* dst, int q, int m,
                                           const mfma_f4 (&res)[7]) {
  if (q < 3) {                               // result rows 4q + r < 12
#pragma unroll
    for (int t = T0; t < T1; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        dst[mfma_o_addr<AX>(q, m, r, t)] = res[OFF + t - T0][r];
  }
}

// One product stage: src arrays S0 / S1 / S2 (the same array three times for
// the forward stage) contracted with aM along axes 0 / 1 / 2, results left in
// `res`; then, after the workgroup has finished reading, written to D0/D1/D2.
__device__ __forceinline__ void mfma_stage(const float (&aM)[3], int wave,
                                           int q, int m, const float* S0,
                                           const float* S1, const float* S2,
                                           float* D0, float* D1, float* D2) {
  mfma_f4 res[7];
  if (wave == 0) mfma_tiles<0, 0, 7, 0>(aM, S0, q, m, res);
  else if (wave == 1) mfma_tiles<1, 0, 7, 0>(aM, S1, q, m, res);
  else if (wave == 2) mfma_tiles<2, 0, 7, 0>(aM, S2, q, m, res);
  else {
    mfma_tiles<0, 7, 9, 0>(aM, S0, q, m, res);
    mfma_tiles<1, 7, 9, 2>(aM, S1, q, m, res);
    mfma_tiles<2, 7, 9, 4>(aM, S2, q, m, res);
  }
  __syncthreads();     // every wave has read its operands
  if (wave == 0) mfma_store<0, 0, 7, 0>(D0, q, m, res);
  else if (wave == 1) mfma_store<1, 0, 7, 0>(D1, q, m, res);
  else if (wave == 2) mfma_store<2, 0, 7, 0>(D2, q, m, res);
  else {
    mfma_store<0, 7, 9, 0>(D0, q, m, res);
    mfma_store<1, 7, 9, 2>(D1, q, m, res);
    mfma_store<2, 7, 9, 4>(D2, q, m, res);
  }
  __syncthreads();
}

template <int GM, bool MASS>
__global__ void __launch_bounds__(256, (GM == GEO_AFFINE ? SFEM_MFMA_MINW : 4))
helmholtz_mfma_p12_kernel(MfmaConsts<12> cst, HelmholtzParams<float> prm) {
  constexpr int P = 12, PP = P * P, N = P * P * P;
  constexpr int KS = P / 4;            // 3 k steps
  constexpr int R2 = 3;                // columns per lane: c = lane + 64 r2
  constexpr int AW = 3;                // slices per wave: a = 3 wave + ai
  // 3 arrays + a 64-float sink: lanes without a point (c >= 144 in the r2 = 2
  // steps) read and write there, so the point loops carry no branches
  __shared__ float lds[3 * N + 64];
  float* A0 = lds;
  float* A1 = lds + N;
  float* A2 = lds + 2 * N;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, m = lane & 15;
  const int64_t work = blockIdx.x;
  const int64_t e = prm.elem_list ? (int64_t)prm.elem_list[work] : work;
  // the constants as memory: kernarg offset 0 (first kernel argument)
#if defined(__HIP_DEVICE_COMPILE__)
  const MfmaConsts<P>* km =
      (const MfmaConsts<P>*)__builtin_amdgcn_kernarg_segment_ptr();
#else
  const MfmaConsts<P>* km = &cst;
#endif

  // A operands: D padded to 16 rows (forward), D^T (transposed products)
  float aD[KS], aDt[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    aD[s] = m < P ? km->d[m * P + 4 * s + q] : 0.f;
    aDt[s] = m < P ? km->d[(4 * s + q) * P + m] : 0.f;
  }

  // ---- gather in the point mapping: slot(ai, r2) = (3 wave + ai) 144 + c
  const int a0 = AW * wave;
  // LDS position of point (ai, r2) relative to A0 / A1 / A2 (A2's view of the
  // sink is shifted by -2N, A1's by -N, so all three land inside it)
  int col[R2];
  bool has[R2];
#pragma unroll
  for (int r2 = 0; r2 < R2; ++r2) {
    has[r2] = lane + 64 * r2 < PP;
    col[r2] = has[r2] ? lane + 64 * r2 : -1;
  }
  uint32_t enc[AW][R2];
  float ua[AW][R2], wa[AW], xa[AW];
  {
    const int32_t* enc0 = prm.enc + e * N + a0 * PP;
#pragma unroll
    for (int ai = 0; ai < AW; ++ai) {
      wa[ai] = km->w[a0 + ai];                 // wave-uniform: scalar loads
      xa[ai] = km->x[a0 + ai];
#pragma unroll
      for (int r2 = 0; r2 < R2; ++r2) {
        const int c = lane + 64 * r2;
        enc[ai][r2] = c < PP ? (uint32_t)__builtin_nontemporal_load(
                                   &enc0[ai * PP + c])
                             : (uint32_t)SFEM_IDX_PAD;
      }
    }
  }
  const float* ug = prm.u;
  float* og = prm.out;
#pragma unroll
  for (int ai = 0; ai < AW; ++ai)
#pragma unroll
    for (int r2 = 0; r2 < R2; ++r2) {
      const uint32_t id = enc[ai][r2] & SFEM_IDX_MASK;
      ua[ai][r2] = id == SFEM_IDX_PAD ? 0.f : ug[id];
    }
#pragma unroll
  for (int ai = 0; ai < AW; ++ai)
#pragma unroll
    for (int r2 = 0; r2 < R2; ++r2) {
      const int c = lane + 64 * r2;
      if (c < PP) A0[(a0 + ai) * PP + c] = ua[ai][r2];
    }
  __syncthreads();

  // per-lane geometry constants of the columns c = (i, j)
  const float* Ae = prm.geo_elem + e * 24;     // wave-uniform: scalar loads
  float wij[R2], xi[R2], xj[R2];
#pragma unroll
  for (int r2 = 0; r2 < R2; ++r2) {
    const int c = lane + 64 * r2;
    const int i = c < PP ? c / P : 0, j = c < PP ? c % P : 0;
    wij[r2] = km->w[i] * km->w[j];
    xi[r2] = km->x[i];
    xj[r2] = km->x[j];
  }
  // W = w detJ and (affine) the constant G / w of this element
  float cg[6] = {0, 0, 0, 0, 0, 0}, cdet = 0.f;
  if (GM == GEO_AFFINE) {
    float r0[3], r1[3], r2v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      r0[c] = Ae[c]; r1[c] = Ae[3 + c]; r2v[c] = Ae[6 + c];
    }
    const float c0[3] = {r1[1] * r2v[2] - r1[2] * r2v[1],
                         r1[2] * r2v[0] - r1[0] * r2v[2],
                         r1[0] * r2v[1] - r1[1] * r2v[0]};
    const float c1[3] = {r2v[1] * r0[2] - r2v[2] * r0[1],
                         r2v[2] * r0[0] - r2v[0] * r0[2],
                         r2v[0] * r0[1] - r2v[1] * r0[0]};
    const float c2[3] = {r0[1] * r1[2] - r0[2] * r1[1],
                         r0[2] * r1[0] - r0[0] * r1[2],
                         r0[0] * r1[1] - r0[1] * r1[0]};
    cdet = r0[0] * c0[0] + r0[1] * c0[1] + r0[2] * c0[2];
    const float inv = 1.f / cdet;
    cg[0] = inv * (c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2]);
    cg[1] = inv * (c0[0] * c1[0] + c0[1] * c1[1] + c0[2] * c1[2]);
    cg[2] = inv * (c0[0] * c2[0] + c0[1] * c2[1] + c0[2] * c2[2]);
    cg[3] = inv * (c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2]);
    cg[4] = inv * (c1[0] * c2[0] + c1[1] * c2[1] + c1[2] * c2[2]);
    cg[5] = inv * (c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2]);
  }
  // Jacobian rows, cofactor rows and determinant of the multilinear map at
  // (r, s, t) = (x[a], x[i], x[j])
  auto jacobian = [&](float r, float s, float tt, float (&c0)[3],
                      float (&c1)[3], float (&c2)[3]) -> float {
    float R0[3], R1[3], R2v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float A1 = Ae[k], A2 = Ae[3 + k], A3 = Ae[6 + k], A4 = Ae[9 + k],
                  A5 = Ae[12 + k], A6 = Ae[15 + k], A7 = Ae[18 + k];
      R0[k] = A1 + A4 * s + (A6 + A7 * s) * tt;
      R1[k] = A2 + A5 * tt + r * (A4 + A7 * tt);
      R2v[k] = A3 + A5 * s + r * (A6 + A7 * s);
    }
    c0[0] = R1[1] * R2v[2] - R1[2] * R2v[1];
    c0[1] = R1[2] * R2v[0] - R1[0] * R2v[2];
    c0[2] = R1[0] * R2v[1] - R1[1] * R2v[0];
    c1[0] = R2v[1] * R0[2] - R2v[2] * R0[1];
    c1[1] = R2v[2] * R0[0] - R2v[0] * R0[2];
    c1[2] = R2v[0] * R0[1] - R2v[1] * R0[0];
    c2[0] = R0[1] * R1[2] - R0[2] * R1[1];
    c2[1] = R0[2] * R1[0] - R0[0] * R1[2];
    c2[2] = R0[0] * R1[1] - R0[1] * R1[0];
    return R0[0] * c0[0] + R0[1] * c0[1] + R0[2] * c0[2];
  };

  const bool has_stiff = prm.lambda1 != 0.f;
  if (has_stiff) {
    // ---- forward products: the gradients along the three axes
    mfma_stage(aD, wave, q, m, A0, A0, A0, A0, A1, A2);
    // ---- point-wise: (w0, w1, w2) = w detJ J^-1 J^-T (g0, g1, g2), in place.
    // All loads first, all stores last: one LDS round trip for the stage
    // instead of one per point (the compiler cannot hoist a load over an
    // earlier store into the same array).
    float h[AW][R2][3];
    int slot[AW][R2];
#pragma unroll
    for (int ai = 0; ai < AW; ++ai)
#pragma unroll
      for (int r2 = 0; r2 < R2; ++r2) {
        // idle lanes: the sink behind the third array (index 3N - kN + lane)
        slot[ai][r2] = has[r2] ? (a0 + ai) * PP + col[r2] : 3 * N + lane;
        h[ai][r2][0] = A0[slot[ai][r2]];
        h[ai][r2][1] = A0[slot[ai][r2] + (has[r2] ? N : 0)];
        h[ai][r2][2] = A0[slot[ai][r2] + (has[r2] ? 2 * N : 0)];
      }
#pragma unroll
    for (int ai = 0; ai < AW; ++ai)
#pragma unroll
      for (int r2 = 0; r2 < R2; ++r2) {
        const float h0 = h[ai][r2][0], h1 = h[ai][r2][1], h2 = h[ai][r2][2];
        const float wq = wa[ai] * wij[r2];
        float w0, w1, w2;
        if (GM == GEO_AFFINE) {
          w0 = wq * (cg[0] * h0 + cg[1] * h1 + cg[2] * h2);
          w1 = wq * (cg[1] * h0 + cg[3] * h1 + cg[4] * h2);
          w2 = wq * (cg[2] * h0 + cg[4] * h1 + cg[5] * h2);
        } else {
          // G g = (w / det) C (C^T g), C = rows of cofactors
          float c0[3], c1[3], c2[3];
          const float det = jacobian(xa[ai], xi[r2], xj[r2], c0, c1, c2);
          const float sc = fast_div(wq, det);
          float y[3];
#pragma unroll
          for (int k = 0; k < 3; ++k)
            y[k] = sc * (c0[k] * h0 + c1[k] * h1 + c2[k] * h2);
          w0 = c0[0] * y[0] + c0[1] * y[1] + c0[2] * y[2];
          w1 = c1[0] * y[0] + c1[1] * y[1] + c1[2] * y[2];
          w2 = c2[0] * y[0] + c2[1] * y[1] + c2[2] * y[2];
        }
        h[ai][r2][0] = w0; h[ai][r2][1] = w1; h[ai][r2][2] = w2;
      }
#pragma unroll
    for (int ai = 0; ai < AW; ++ai)
#pragma unroll
      for (int r2 = 0; r2 < R2; ++r2) {
        A0[slot[ai][r2]] = h[ai][r2][0];
        A0[slot[ai][r2] + (has[r2] ? N : 0)] = h[ai][r2][1];
        A0[slot[ai][r2] + (has[r2] ? 2 * N : 0)] = h[ai][r2][2];
      }
    __syncthreads();
    // ---- transposed products, back at their natural positions
    mfma_stage(aDt, wave, q, m, A0, A1, A2, A0, A1, A2);
  }

  // ---- sum, mass term, direct-stiffness summation (flags as helmholtz_kernel)
  double udot = 0.0;
  float vsum[AW][R2];
#pragma unroll
  for (int ai = 0; ai < AW; ++ai)
#pragma unroll
    for (int r2 = 0; r2 < R2; ++r2) {
      vsum[ai][r2] = 0.f;
      if (has_stiff) {
        const int sl = has[r2] ? (a0 + ai) * PP + col[r2] : 3 * N + lane;
        vsum[ai][r2] = A0[sl] + A0[sl + (has[r2] ? N : 0)] +
                       A0[sl + (has[r2] ? 2 * N : 0)];
      }
    }
#pragma unroll
  for (int ai = 0; ai < AW; ++ai)
#pragma unroll
    for (int r2 = 0; r2 < R2; ++r2) {
      const uint32_t ea = enc[ai][r2];
      const uint32_t id = ea & SFEM_IDX_MASK;
      if (has[r2] && id != SFEM_IDX_PAD) {
        float v = prm.lambda1 * vsum[ai][r2];
        if (MASS) {
          float W;
          if (GM == GEO_AFFINE) {
            W = wa[ai] * wij[r2] * cdet;
          } else {
            float c0[3], c1[3], c2[3];
            W = wa[ai] * wij[r2] *
                jacobian(xa[ai], xi[r2], xj[r2], c0, c1, c2);
          }
          v += prm.lambda0 * W * ua[ai][r2];
        }
        float* dst = og + id;
        const bool dirichlet = ea & SFEM_IDX_DIRICHLET;
        if (!dirichlet) udot += (double)v * (double)ua[ai][r2];
        if (ea & SFEM_IDX_SHARED) {
          if (!dirichlet) unsafeAtomicAdd(dst, v);
        } else {
          *dst = dirichlet ? 0.f : v;
        }
      }
    }
  if (prm.dot_out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) udot += __shfl_down(udot, off, 64);
    if (lane == 0)
      unsafeAtomicAdd(&prm.dot_out[(blockIdx.x * 4 + wave) &
                                   (SFEM_DOT_SLOTS - 1)], udot);
  }
}

// True if the matrix-core kernel covers this launch.
inline bool helmholtz_mfma_applies(const HelmholtzParams<float>& prm, int P,
                                   int ndim, bool gs) {
  // the kernel addresses u / out as dense scalar fields
  return gs && ndim == 3 && P == 12 && prm.ncomp == 1 &&
         prm.node_stride == 1 && prm.comp == 0 && !prm.colored &&
         (prm.geo_mode == GEO_AFFINE || prm.geo_mode == GEO_MULTILINEAR) &&
         prm.geo_elem && prm.weights_host && prm.nodes_host;
}

int launch_helmholtz_mfma_p12(const HelmholtzParams<float>& prm,
                              hipStream_t stream);

}  // namespace sfem
