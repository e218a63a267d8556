// ionode_grad_reduce.hpp -- weight gradients from the backward sweep's (d_l, h_l) record stream: a split-K fp32 MFMA GEMM.
//
//   dW_l[row][col] = sum over records, trajectories n:  d_l[row][n] * h_{l-1}[col][n]        l = 1..L   ("heavy" jobs)
//   db_l[row]      = sum d_l[row][n]                                                         l = 0..L
//   dW_0[row][c]   = sum d_0[row][n] * x_c[n]      (x = (V/100, a): the net's inputs)        ("light" job 0)
//   dwl[k]         = sum seed[n] * h_L[k][n],  dbl = sum seed[n]                             ("light" job L+1)
//
// A record is one MLP vector-Jacobian product of one 16-trajectory tile (ionode_grad.hpp); its tiles are stored in the
// operand layout of v_mfma_f32_16x16x4_f32 with the TRAJECTORY as contraction index: lane = 16*kk + m holds
// X[16*rt + m][4*c + kk] in component c, so a 1 KiB tile is both an A operand (rows of d) and a B operand (rows of h) of four
// MFMAs (c = 0..3) with one coalesced 16-byte load per lane.  Workgroup = (job, slab of records): a heavy job keeps a layer's
// NP x (column block) accumulator in registers (N = 200, round 5: two blocks of 7 and 6 column tiles, 24 tiles per wavefront, three
// workgroups per compute unit; N <= 112 the whole NP x NP; N = 500 four blocks), stages each record's NT + CB tiles through LDS once
// (double-buffered) and writes one partial per slab; the host sums the slabs in fp64 (deterministic, no atomics).  The two light
// jobs of a slab share one workgroup (four records in flight).  grad_reduce_slabs(): the slab count that makes one round of workgroups.
// Roofline: 2 * NP^2 * 16 FLOP per 2 * NT KiB of record -> 53 FLOP/B for N = 200: HBM- and MFMA-balanced at ~3 TB/s.
#pragma once

#include "ionode_grad.hpp"

namespace ionode {

// floats of one slab's partial: job 0 [NP][4] {db0, dW0[.][0], dW0[.][1], 0} | jobs 1..L: dW_l [NP][NP] + db_l [NP] | job L+1: dwl [NP] + {dbl,0,0,0}
__host__ __device__ constexpr size_t grad_partial_floats(int L, int NT) {
  return (size_t)4 * 16 * NT + (size_t)L * ((size_t)256 * NT * NT + 16 * NT) + 16 * NT + 4;
}

// Column blocks of a heavy job.  N = 500 (NT = 32: 1024 accumulator registers per wavefront) is cut into 4 jobs of 8 column tiles.
// N = 200 (NT = 13), round 5: TWO jobs of 7 and 6 column tiles (96 accumulator registers per wavefront instead of 172), so that THREE workgroups
// share a compute unit: one's staging, barrier and LDS reads run beside the others' MFMAs, the light workgroups sit beside heavy ones
// instead of holding a compute unit of their own, and 2 L + 1 workgroups per slab deal the records out finer (one round: 69 slabs on 768
// workgroup slots instead of 42 on 256; regression step 1.84 -> 1.71 ms, 46 slabs at two per unit: 1.74).  Both blocks stage the D_l tiles; the sums are the same chains in the same order: same bits.
__host__ __device__ constexpr int grad_reduce_cb(int NT) { return NT == 13 ? 7 : (NT < 13 ? NT : 8); }
__host__ __device__ constexpr int grad_reduce_ncb(int NT) { return (NT + grad_reduce_cb(NT) - 1) / grad_reduce_cb(NT); }
__host__ __device__ constexpr int grad_reduce_wg_per_cu(int NT) { return NT == 13 ? 3 : 1; }   // (168 registers, 40 KB of LDS: three fit)

template <int NT>
// unit_seed: the records' D tiles were produced with seed 1 (the factored two-phase sweep: the product is linear in the seed, a
// scalar per trajectory, which the walk writes into the record's scalar block afterwards); they are scaled by it while staged.
__global__ void __launch_bounds__(256, grad_reduce_wg_per_cu(NT)) ionode_grad_reduce_kernel(const float *__restrict__ records, int64_t n_records, int n_slabs,
                                                                 int L, float *__restrict__ partials, int unit_seed) {
  constexpr int NP = 16 * NT;
  constexpr int F = NT / 4;        // full row tiles per wavefront (rt = wave + 4i, every column tile)
  constexpr int R = NT - 4 * F;    // remainder row tiles, their column tiles dealt round-robin over the wavefronts
  // column block of a heavy job (see grad_reduce_cb): CB column tiles from cb0 on (the last block of N = 200 has one tile less: its
  // seventh is skipped, wave-uniformly); every block stages the D_l tiles and its own H_{l-1} tiles
  constexpr int CB = grad_reduce_cb(NT);
  constexpr int NCB = grad_reduce_ncb(NT);
  constexpr int RC = (CB + 3) / 4;   // column tiles of a remainder row tile per wavefront (block-local tiles ct with ct % 4 == wave)
  constexpr int RCL = (NT + 3) / 4;  // the light jobs' row tiles per wavefront
  constexpr int STE = (NT + CB) * 64;             // float4 elements staged per record: D_l tiles + this block's H_{l-1} tiles
  constexpr int STG = (STE + 255) / 256;          // float4 loads per thread per record
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f32x4 *buf = reinterpret_cast<f32x4 *>(smem);  // [2][STE]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NJOB = L * NCB + 1;                   // job 0 (both light jobs) | heavy jobs (layer, column block)
  const int jobx = blockIdx.x % NJOB;
  const int slab = blockIdx.x / NJOB;
  const int job = (jobx == 0) ? 0 : 1 + (jobx - 1) / NCB;   // 0 | layer 1..L
  const int cb0 = __builtin_amdgcn_readfirstlane((jobx == 0) ? 0 : ((jobx - 1) % NCB) * CB);   // first column tile of the block
  const int64_t r0 = n_records * slab / n_slabs, r1 = n_records * (slab + 1) / n_slabs;
  const int64_t RECF = grad_record_floats(L, NT);
  float *__restrict__ out = partials + (size_t)slab * grad_partial_floats(L, NT);
  const int m = lane & 15, kk = lane >> 4;

  if (job == 0) {
    // ---- the two light jobs (round 5: ONE workgroup runs both, four records in flight): one d (or h) tile set per record against
    // per-trajectory scalars, VALU, straight from global.  (Round 4: a workgroup per light job, one record at a time: a dependent load
    // chain of ~2 us per record -- as long as a heavy job; with both in one pipelined workgroup a slab costs L + 1 workgroups instead of
    // L + 2, i.e. 42 slabs instead of 36 in one round of workgroups for N = 200, L = 5.)  Summation order per slab: record order, as before.
    constexpr int NB = 4;
    for (int lj = 0; lj < 2; ++lj) {
      const bool first = lj == 0;
      float a0[RCL], a1[RCL], a2[RCL];
#pragma unroll
      for (int i = 0; i < RCL; ++i) a0[i] = a1[i] = a2[i] = 0.0f;
      float sg = 0.0f;
      for (int64_t rb = r0; rb < r1; rb += NB) {
        f32x4 t[NB][RCL];
        float s0[NB][4], s1[NB][4], sd[NB][4], sgl[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const int64_t rr = (rb + u < r1) ? rb + u : r1 - 1;   // (past the end: a valid record, its contribution is skipped below)
          const float *rec = records + rr * RECF;
          const f32x4 *tiles = reinterpret_cast<const f32x4 *>(rec) + (size_t)(first ? (L + 1) * NT : L * NT) * 64;  // D_0 | H_L
          const float *sc = rec + (size_t)2 * (L + 1) * NT * 256;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            s0[u][c] = first ? sc[4 * c + kk] : sc[32 + 4 * c + kk];  // x0 | seed
            s1[u][c] = first ? sc[16 + 4 * c + kk] : 0.0f;            // x1
            sd[u][c] = sc[32 + 4 * c + kk];
          }
          sgl[u] = (lane < 16) ? sc[32 + lane] : 0.0f;
#pragma unroll
          for (int i = 0; i < RCL; ++i) {
            const int rt = wave + 4 * i;
            t[u][i] = (rt < NT) ? tiles[rt * 64 + lane] : f32x4{0, 0, 0, 0};
          }
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          if (rb + u < r1) {
            if (!first && wave == 0 && lane < 16) sg += sgl[u];
#pragma unroll
            for (int i = 0; i < RCL; ++i) {
              const int rt = wave + 4 * i;
              if (rt < NT) {
                f32x4 tt = t[u][i];
                if (first && unit_seed) {
#pragma unroll
                  for (int c = 0; c < 4; ++c) tt[c] *= sd[u][c];   // D_0 of a unit-seed record
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                  a0[i] += first ? tt[c] : tt[c] * s0[u][c];
                  if (first) { a1[i] = fmaf(tt[c], s0[u][c], a1[i]); a2[i] = fmaf(tt[c], s1[u][c], a2[i]); }
                }
              }
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < RCL; ++i) {
        const int rt = wave + 4 * i;
        float v0 = a0[i], v1 = a1[i], v2 = a2[i];
        v0 += __shfl_xor(v0, 16); v0 += __shfl_xor(v0, 32);
        v1 += __shfl_xor(v1, 16); v1 += __shfl_xor(v1, 32);
        v2 += __shfl_xor(v2, 16); v2 += __shfl_xor(v2, 32);
        if (rt < NT && lane < 16) {
          if (first) {
            float *o = out + (size_t)(16 * rt + m) * 4;
            o[0] = v0; o[1] = v1; o[2] = v2; o[3] = 0.0f;
          } else {
            out[(size_t)4 * NP + (size_t)L * ((size_t)NP * NP + NP) + 16 * rt + m] = v0;
          }
        }
      }
      if (!first && wave == 0) {
        float tsum = (lane < 16) ? sg : 0.0f;
#pragma unroll
        for (int sft = 1; sft < 16; sft <<= 1) tsum += __shfl_xor(tsum, sft);
        if (lane == 0) {
          float *o = out + (size_t)4 * NP + (size_t)L * ((size_t)NP * NP + NP) + NP;
          o[0] = tsum; o[1] = o[2] = o[3] = 0.0f;
        }
      }
    }
    return;
  }

  // ---- heavy job l: dW_l += D_l . H_{l-1}^T over this slab's records ----
  const int l = job;
  f32x4 acc[F > 0 ? F : 1][CB], accr[R > 0 ? R : 1][RC], dba[F > 0 ? F : 1], dbr[R > 0 ? R : 1];
#pragma unroll
  for (int i = 0; i < F; ++i) {
    dba[i] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int ct = 0; ct < CB; ++ct) acc[i][ct] = f32x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    dbr[j] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < RC; ++u) accr[j][u] = f32x4{0, 0, 0, 0};
  }
  // staging: element e of the record's {D_l tiles, H_{l-1} tiles} (2*NT*64 float4), thread tid takes e = tid + 256*u
  auto src_of = [&](int64_t rr, int e) -> const f32x4 * {
    const f32x4 *base = reinterpret_cast<const f32x4 *>(records + rr * RECF);
    return (e < NT * 64) ? base + (size_t)((L + 1) + l) * NT * 64 + e                          // D_l
                         : base + ((size_t)(l - 1) * NT + cb0) * 64 + (e - NT * 64);            // H_{l-1}, this block's tiles
  };
  f32x4 stg[STG];
  // a unit-seed record's 16 seeds are STAGED with its tiles (one float per thread 0..15, into 64 bytes behind the tile buffers).  Loaded
  // from global memory in the iteration that uses them (rounds 3-5) their wait was `s_waitcnt vmcnt(0)` in front of the record's first
  // MFMA, which also waited for the NEXT record's staging loads -- every record paid a full memory round trip, with or without seeds.
  float *seedbuf = reinterpret_cast<float *>(buf + (size_t)2 * STE);   // [2][16]
  auto seed_src = [&](int64_t rr) -> const float * { return records + rr * RECF + (size_t)2 * (L + 1) * NT * 256 + 32 + (threadIdx.x & 15); };
  float sdg = 1.0f;
  if (r0 < r1) {
#pragma unroll
    for (int u = 0; u < STG; ++u) {
      const int e = threadIdx.x + 256 * u;
      if (e < STE) buf[e] = *src_of(r0, e);
    }
    if (unit_seed && threadIdx.x < 16) seedbuf[threadIdx.x] = *seed_src(r0);
  }
  __syncthreads();
  for (int64_t rr = r0; rr < r1; ++rr) {
    const int cur = (int)((rr - r0) & 1);
    const f32x4 *__restrict__ Db = buf + (size_t)cur * STE;
    const f32x4 *__restrict__ Hb = Db + NT * 64;
    const bool more = rr + 1 < r1;
    if (more) {
#pragma unroll
      for (int u = 0; u < STG; ++u) {
        const int e = threadIdx.x + 256 * u;
        if (e < STE) stg[u] = *src_of(rr + 1, e);
      }
      if (unit_seed && threadIdx.x < 16) sdg = *seed_src(rr + 1);
    }
    f32x4 sd = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
    if (unit_seed) {
#pragma unroll
      for (int c = 0; c < 4; ++c) sd[c] = seedbuf[cur * 16 + 4 * c + kk];   // this lane's component c is trajectory 4 c + kk
    }
    f32x4 af[F > 0 ? F : 1], ar[R > 0 ? R : 1];
#pragma unroll
    for (int i = 0; i < F; ++i) {
      af[i] = Db[(wave + 4 * i) * 64 + lane];
      if (unit_seed) af[i] = af[i] * sd;
      dba[i] += af[i];
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      ar[j] = Db[(4 * F + j) * 64 + lane];
      if (unit_seed) ar[j] = ar[j] * sd;
      dbr[j] += ar[j];
    }
#pragma unroll
    for (int ct = 0; ct < CB; ++ct) {
      if (NT % CB != 0 && ct == CB - 1 && cb0 + ct >= NT) break;   // (the short last block lacks its last tile; wave-uniform)
      const f32x4 b = Hb[ct * 64 + lane];
      // trajectory group c outer, row tile inner: consecutive MFMAs on different accumulators (issue 32 cycles, result 40)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < F; ++i) acc[i][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][c], b[c], acc[i][ct], 0, 0, 0);
      if (R > 0 && (ct & 3) == wave) {  // wave-uniform: this wavefront's column tiles of the remainder row tiles
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int j = 0; j < R; ++j) accr[j][ct / 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[j][c], b[c], accr[j][ct / 4], 0, 0, 0);
      }
    }
    if (more) {
      f32x4 *nb = buf + (size_t)(cur ^ 1) * STE;
#pragma unroll
      for (int u = 0; u < STG; ++u) {
        const int e = threadIdx.x + 256 * u;
        if (e < STE) nb[e] = stg[u];
      }
      if (unit_seed && threadIdx.x < 16) seedbuf[(cur ^ 1) * 16 + threadIdx.x] = sdg;
    }
    __syncthreads();
  }
  // ---- write the partial: accumulator register r of lane (q = lane >> 4, n' = lane & 15) is dW[16*rt + 4q + r][16*ct + n'] ----
  float *__restrict__ W = out + (size_t)4 * NP + (size_t)(l - 1) * ((size_t)NP * NP + NP);
  float *__restrict__ bvec = W + (size_t)NP * NP;
#pragma unroll
  for (int i = 0; i < F; ++i) {
    const int rt = wave + 4 * i;
#pragma unroll
    for (int ct = 0; ct < CB; ++ct) {
      if (NT % CB != 0 && ct == CB - 1 && cb0 + ct >= NT) break;
#pragma unroll
      for (int r = 0; r < 4; ++r) W[(size_t)(16 * rt + 4 * kk + r) * NP + 16 * (cb0 + ct) + m] = acc[i][ct][r];
    }
    float s = (dba[i][0] + dba[i][1]) + (dba[i][2] + dba[i][3]);
    s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
    if (lane < 16 && cb0 == 0) bvec[16 * rt + m] = s;   // the bias gradient once per layer (column block 0)
  }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int rt = 4 * F + j;
#pragma unroll
    for (int u = 0; u < RC; ++u) {
      const int ct = 4 * u + wave;   // block-local column tile
      if (ct < CB && cb0 + ct < NT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) W[(size_t)(16 * rt + 4 * kk + r) * NP + 16 * (cb0 + ct) + m] = accr[j][u][r];
      }
    }
    float s = (dbr[j][0] + dbr[j][1]) + (dbr[j][2] + dbr[j][3]);
    s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
    if (wave == 0 && lane < 16 && cb0 == 0) bvec[16 * rt + m] = s;
  }
}

// slabs that make ONE round of workgroups on `cus` compute units (L x NCB heavy workgroups + one light workgroup per slab); at least
// four records per slab
inline int grad_reduce_slabs(int L, int NT, int cus, int64_t n_records) {
  const int64_t slots = (int64_t)cus * grad_reduce_wg_per_cu(NT), per_slab = (int64_t)L * grad_reduce_ncb(NT) + 1;
  int64_t n = slots / per_slab;
  if (n > n_records / 4) n = n_records / 4;
  return n < 1 ? 1 : (int)n;
}

inline hipError_t launch_grad_reduce(int L, int NT, const float *records, int64_t n_records, int n_slabs, float *partials,
                                     hipStream_t s, int unit_seed = 0) {
  const int CB = grad_reduce_cb(NT), NCB = grad_reduce_ncb(NT);
  const unsigned grid = (unsigned)(n_slabs * (L * NCB + 1));
  const size_t lds = (size_t)2 * (NT + CB) * 64 * 16 + 128;   // tile buffers + the staged seeds
  switch (NT) {
    case 1: hipLaunchKernelGGL(ionode_grad_reduce_kernel<1>, dim3(grid), dim3(256), lds, s, records, n_records, n_slabs, L, partials, unit_seed); break;
    case 7: hipLaunchKernelGGL(ionode_grad_reduce_kernel<7>, dim3(grid), dim3(256), lds, s, records, n_records, n_slabs, L, partials, unit_seed); break;
    case 13: hipLaunchKernelGGL(ionode_grad_reduce_kernel<13>, dim3(grid), dim3(256), lds, s, records, n_records, n_slabs, L, partials, unit_seed); break;
    case 32: {
      auto kern = ionode_grad_reduce_kernel<32>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, records, n_records, n_slabs, L, partials, unit_seed);
      break;
    }
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace ionode
