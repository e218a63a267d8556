// ionode_capi.hip -- the C ABI of libionode.so (include/ionode.h): argument checking, choice of
// kernel instantiation and launch geometry, and the host-side re-layout of an nn.Sequential
// state dict into the MFMA fragment order the kernels stream.
#include <cstdio>
#include <cstring>
#include <vector>

#include "ionode_launch.hpp"

#ifndef IONODE_TILE32_FROM
#define IONODE_TILE32_FROM 8192  // N = 200: two 16-trajectory tiles per compute unit
#endif
#ifndef IONODE_TILE4_UPTO
#define IONODE_TILE4_UPTO 1024  // N = 200: up to this many trajectories, 4 per tile = at most one tile per compute unit (16-tiles would use <= 64 of the 256 CUs)
#endif
#ifndef IONODE_TILE1_UPTO
#define IONODE_TILE1_UPTO 256   // N = 200: up to this many trajectories, ONE per tile (MlpRow1: a lane owns a row) = at most one tile per compute unit
#endif
#ifndef IONODE_TINY64_MFMA
#define IONODE_TINY64_MFMA 0   // 1: N = 10 keeps the MFMA form at 64 trajectories per wavefront (A/B)
#endif
#ifndef IONODE_TINY64_FROM
#define IONODE_TINY64_FROM 32769  // round 4, final build (per-lane packed net, even placement; 20 001 samples): 16 per wavefront 12.0-12.1 ms from 16 384 to 32 768 (two wavefronts per SIMD), 19.6 at 49 152, 23.1 at 65 536; 64 per wavefront 13.3-13.7 ms from 8 192 to 65 536
#endif

namespace {

thread_local char g_err[256] = "";
thread_local const char *g_last_kernel = "";  // variant name of this thread's last successful ionode_dopri5 launch

void set_err(const char *fmt, const char *detail = "") { snprintf(g_err, sizeof g_err, fmt, detail); }

inline int np_of(int N) { return 16 * ((N + 15) / 16); }

struct Plan {
  const ionode::Variant *v = nullptr;
  unsigned grid = 0;
  unsigned block = 0;
  size_t lds = 0;
  size_t lw_bytes = 0;   // lane-wise kernels: LDS bytes of one wavefront's region (the workgroup reserves four)
};


// Lane-wise kernels: four one-wavefront tiles per workgroup (ionode_device.hpp IONODE_LW_TILES_PER_WG), tile t on XCD t % 8.  The
// workgroup count is a multiple of 8 so that (workgroup, wavefront) -> tile is onto; empty tiles leave at once.
// EVEN PLACEMENT: the hardware places whole workgroups, and a four-wavefront workgroup occupies one slot on each SIMD of its compute
// unit -- so the workgroups that fit a CU are the wavefronts per SIMD.  A launch of fewer tiles than the kernel's natural residency
// reserves more LDS per workgroup (a multiple of the 1280-byte granule), capping the workgroups per CU at ceil(workgroups / CUs):
// without the cap the dispatcher stacks a small launch three deep on some SIMDs and leaves others idle.
void plan_lane_wise(Plan *pl, size_t tiles, size_t per_wave_bytes) {
  const size_t cap = 160 * 1024, gran = 1280, T = IONODE_LW_TILES_PER_WG;
  pl->lw_bytes = (per_wave_bytes + 15) & ~(size_t)15;
  pl->grid = (unsigned)(8 * ((tiles + 8 * T - 1) / (8 * T)));
  pl->block = (unsigned)(64 * T);
  size_t lds = T * pl->lw_bytes;
#ifndef IONODE_EVEN_PLACEMENT
#define IONODE_EVEN_PLACEMENT 1
#endif
  if (IONODE_EVEN_PLACEMENT) {
    // compute units of the current device (256 on an unpartitioned MI355X).  The padding assumes the 8-XCD round-robin of the whole
    // chip: on a partitioned device (CPX / DPX) or without a device (the plan is also computed on hosts without a GPU) it is skipped
    static const int ncu_dev = [] {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
      return n;
    }();
    const size_t ncu = 256;
    const size_t per_cu = (pl->grid + ncu - 1) / ncu;
    const size_t natural = cap / (((lds + gran - 1) / gran) * gran);
    if ((ncu_dev == 0 || ncu_dev == (int)ncu) && per_cu >= 1 && per_cu < natural) {
      const size_t pad = (cap / per_cu) / gran * gran;
      if (pad > lds) lds = pad;
    }
  }
  pl->lds = lds;
}

// Batch size from which the dispatcher takes the one-trajectory-per-lane (64 per wavefront) kernel of a model; 0: the model has none.
// Exported as ionode_lane_wise_from() so that host code (capi.py: protocol-major launch order) does not keep a copy of the numbers.
int lane_wise_from(int model, int mlp_width) {
  if (model == IONODE_MODEL_HH2) return 49152;
  if (model == IONODE_MODEL_MARKOV6) return 24576;
  if ((model == IONODE_MODEL_NNF || model == IONODE_MODEL_NND) && mlp_width >= 1 && mlp_width <= 16) return IONODE_TINY64_FROM;
  return 0;
}

// Closed-form kernels are registered with NT == 0 and RT = trajectories per wavefront (0 -> 64); rt < 0: any RT.
const ionode::Variant *find_variant(int model, int f32, int G, int NT, int rt = -1, int tail = 0, int pd = -1) {
  using namespace ionode;
  typedef const Variant *(*TabFn)(int *);
  static const TabFn tabs[] = {variants_closed, variants_nnf_f64, variants_nnf_f32, variants_nnd_f64, variants_nnd_f32};
  for (TabFn tf : tabs) {
    int n = 0;
    const Variant *t = tf(&n);
    for (int i = 0; i < n; ++i)
      if (t[i].model == model && t[i].f32 == f32 && (G == 0 || t[i].G == G) && t[i].NT == NT && (rt < 0 || t[i].RT == rt) && t[i].tail == tail && (pd < 0 ? !(t[i].RT == 64 && t[i].PD > 1) : t[i].PD == pd))
        return &t[i];
  }
  return nullptr;
}

int make_plan(const ionode_desc *d, Plan *pl, bool want_current = false, bool explicit_grid = false) {
  if (!d) { set_err("null descriptor"); return IONODE_ERR_ARG; }
#ifdef IONODE_STAMPS
  const bool has_step_log = false;   // diagnostic build: the step log carries the phase stamps and does not exclude the lean variants
#else
  const bool has_step_log = d->step_log != nullptr;
#endif
  const bool mlp = d->model == IONODE_MODEL_NNF || d->model == IONODE_MODEL_NND;
  const int D = d->model == IONODE_MODEL_MARKOV6 ? 6 : 2;
  if (d->model < 0 || d->model > 3) { set_err("unknown model"); return IONODE_ERR_ARG; }
  if (d->n_state != D) { set_err("n_state does not match model"); return IONODE_ERR_ARG; }
  if (d->n_traj < 1 || d->n_out < 1 || d->n_prot < 1 || d->prot_n < 2) { set_err("empty batch / grid / protocol"); return IONODE_ERR_ARG; }
  if (d->n_params < (D == 6 ? 12 : 8)) { set_err("n_params too small for model"); return IONODE_ERR_ARG; }
  if (!(d->rtol > 0) || !(d->atol >= 0) || !(d->prot_dt > 0)) { set_err("rtol/atol/prot_dt must be positive"); return IONODE_ERR_ARG; }
  const int f32 = d->state_f32 ? 1 : 0;
  if (!mlp) {
    // small batches: 16 trajectories per wavefront (lanes replicated 4x, 4x the wavefronts) while the launch has fewer wavefronts than
    // the chip has SIMDs to spread them over; measured crossovers (round 4, tools/gpu/r4_t4b.sh, 20 001 samples): 2-state 32 768: 6.3 ms at
    // 16 per wavefront / 9.7 at 64, 65 536: 11.6 / 10.1; 6-state 16 384: 9.1 / 13.8, 32 768: 17.5 / 15.0.  tile_waves = 64 / 16 forces a choice.
    const int tpw64_from = lane_wise_from(d->model, 0);
    const int tpw = (d->tile_waves == 64 || d->tile_waves == 16) ? d->tile_waves : (d->n_traj >= tpw64_from ? 64 : 16);
    // The specialised variants are compiled under a CONTRACT (ionode_device.hpp, top of the kernel): uniform protocol grid, no step log,
    // no checkpoints -- anything else takes the general variant (TAIL slot 0).
    const bool lean_ok = !explicit_grid && !has_step_log && !d->ckpt;
    //   1: the lean variant -- states only on a VERIFIED uniform output grid, no current trace / objective
    //   2: current trace / fused objective with the protocol-at-outputs table given (hint path)
    const int tail = !lean_ok ? 0
                      : (want_current && d->v_at_outputs && d->t_eval_dt_hint > 0.0 && d->n_out > 1) ? 2
                      : ((d->t_eval_exact && d->t_eval_dt_hint > 0.0 && d->n_out > 1 && !want_current) ? 1 : 0);
    // (Rounds 2-3 kept two builds of the 2-state kernels -- 2 and 3 wavefronts per SIMD -- and switched at 2048 wavefronts; since round 4
    // one build per variant: the lean one fits four per SIMD, the others three, without a spill: ionode_device.hpp IONODE_WAVES_PER_SIMD.)
    // 6-state model: two wavefronts per SIMD (lean), otherwise one.
    pl->v = find_variant(d->model, f32, 1, 0, tpw == 64 ? 0 : 16, tail);
    plan_lane_wise(pl, (size_t)((d->n_traj + tpw - 1) / tpw), (size_t)ionode::LwLds::bytes(D, tail));
  } else {
    if (d->mlp_width < 1 || d->mlp_layers < 0) { set_err("bad MLP shape"); return IONODE_ERR_ARG; }
    if (d->mlp_width <= 16 && d->mlp_layers > 10) { set_err("N <= 16 kernels keep at most 10 hidden layers resident"); return IONODE_ERR_UNSUPPORTED; }
    const int NP = np_of(d->mlp_width), NT = NP / 16;
    if (d->tile_waves != 0 && d->tile_waves != 1 && d->tile_waves != 4 && !(d->tile_waves == 64 && NT == 1) && !((d->tile_waves == 8 || d->tile_waves == 2 || d->tile_waves == 16) && NT == 13)) {
      set_err("tile_waves must be 0, 1 or 4 for MLP models (64: the N <= 16 kernel at 64 trajectories per wavefront; 8 / 2 / 16: the N = 200 kernel with 32 / 4 / 1 trajectories per tile)");
      return IONODE_ERR_UNSUPPORTED;
    }
    // N <= 16 (architectures s03-s05): from IONODE_TINY64_FROM trajectories on, one trajectory per lane (64 per wavefront, four
    // MFMA column tiles per evaluation) instead of 16 per wavefront with the scalar integrator work replicated over 4 lane groups
    // (several weight images: the automatic choice takes the 64-per-wavefront kernel only when an image's trajectories fill whole
    // 64-lane tiles -- a population of nets padded to 16 / 32 / 48 trajectories per candidate stays on the 16-per-wavefront kernel;
    // an explicit tile_waves = 64 with such a population is still an argument error, below)
    const bool img64 = d->traj_per_image <= 0 || d->traj_per_image % 64 == 0;
    const bool t64 = NT == 1 && (d->tile_waves == 64 || (d->tile_waves == 0 && d->n_traj >= IONODE_TINY64_FROM && img64));
    // (the lean variant's contract as for the closed-form kernels: verified uniform output grid, no current / objective)
    const int t64lean = (t64 && d->t_eval_exact && d->t_eval_dt_hint > 0.0 && d->n_out > 1 && !want_current && !explicit_grid && !has_step_log && !d->ckpt) ? 1 : 0;
    // N = 200: from two 16-trajectory tiles per compute unit on (8192 trajectories), 32-trajectory tiles -- two column sets per weight
    // fragment, the scalar integrator work replicated twice instead of four times (tile_waves = 8 forces it, 4 forces the 16-tile).
    // Needs a hidden layer (asm stream) and weight images that cover whole 32-trajectory tiles.
    const bool t32 = !t64 && NT == 13 && d->mlp_layers >= 1 && (d->traj_per_image <= 0 || d->traj_per_image % 32 == 0) &&
                     (d->tile_waves == 8 || (d->tile_waves == 0 && d->n_traj >= IONODE_TILE32_FROM));
    // N = 10 (architectures s03-s05) at one trajectory per lane: the per-lane vector-ALU net (MlpLane), unless IONODE_TINY64_MFMA
    const bool vnet = t64 && d->mlp_width == 10 && !IONODE_TINY64_MFMA;
    // N = 200, small batches and single calls: 4 trajectories per tile (MlpTile4; tile_waves = 2 forces it, 4 / 8 exclude it)
    // N = 200, single calls and the smallest batches: ONE trajectory per tile (MlpRow1; tile_waves = 16 forces it, 2 / 4 / 8 exclude it)
    const bool t1 = !t64 && !t32 && NT == 13 && d->mlp_layers >= 1 && d->mlp_layers <= 15 &&
                    (d->tile_waves == 16 || (d->tile_waves == 0 && d->n_traj <= IONODE_TILE1_UPTO));
    const bool t1deep = t1 && d->mlp_layers > ionode::MlpRow1::max_layers();   // no room for LDS-resident steps: every step streamed
    const bool t4 = !t64 && !t32 && !t1 && NT == 13 && d->mlp_layers >= 1 && (d->traj_per_image <= 0 || d->traj_per_image % 4 == 0) &&
                    (d->tile_waves == 2 || (d->tile_waves == 0 && d->n_traj <= IONODE_TILE4_UPTO));
    // N = 200 tiles: the lean variant when its contract holds (ionode_device.hpp LEANM)
    const bool leanm = !t64 && (NT == 13 || NT == 7 || NT == 32) && d->mlp_layers >= 1 && !explicit_grid && !has_step_log && !d->ckpt && d->t_eval_exact && d->t_eval_dt_hint > 0.0 && d->n_out > 1;
    pl->v = t64 ? find_variant(d->model, f32, 1, NT, 64, t64lean, vnet ? 10 : 1)
                : find_variant(d->model, f32, ((d->tile_waves == 8 || d->tile_waves == 2 || d->tile_waves == 16) ? 4 : d->tile_waves), NT, NT == 1 ? 1 : -1, (t32 ? 4 : 0) | (leanm ? 8 : 0) | (t4 ? 16 : 0) | (t1 ? 32 : 0) | (t1deep ? 64 : 0));
    // any other width up to 512 (table-s1.py:145-153 builds nets of any (n_layers, n_nodes)): the run-time-width tile (MlpGen)
    bool gen = false;
    if (!pl->v && NT >= 2 && NT <= ionode::MlpGen::NT_MAX && (d->tile_waves == 0 || d->tile_waves == 4)) {
      const bool leang = d->mlp_layers >= 1 && !explicit_grid && !has_step_log && !d->ckpt && d->t_eval_exact && d->t_eval_dt_hint > 0.0 && d->n_out > 1;
      pl->v = find_variant(d->model, f32, 4, 0, 1, leang ? 8 : 0);
      gen = pl->v != nullptr;
      if (gen && ionode::MlpGen::lds_bytes(d->mlp_layers, NT) > 160 * 1024) {
        set_err("this (layers, width) needs more than 160 KB of LDS for its biases and activations");
        return IONODE_ERR_UNSUPPORTED;
      }
    }
    if (!pl->v) {
      set_err("MLP width outside the compiled kernel variants: 1 <= N <= 512 (tuned tiles for N = 10, 100, 200, 500 -- architectures "
              "s00-s11 -- and the run-time-width tile for every other N; tile_waves must be 0 or 4 for the latter)");
      return IONODE_ERR_UNSUPPORTED;
    }
    pl->grid = t64 ? (unsigned)((d->n_traj + 63) / 64) : (t32 ? (unsigned)((d->n_traj + 31) / 32) : (t1 ? (unsigned)d->n_traj : (t4 ? (unsigned)((d->n_traj + 3) / 4) : (unsigned)((d->n_traj + 15) / 16))));
    pl->block = 64u * pl->v->G;
    const int Gv = pl->v->G, Rv = NT - Gv * (NT / Gv);  // remainder row tiles: K-split partial sums in LDS
    pl->lds = ((size_t)2 * (NT + Gv - 1) * 64 + (size_t)2 * Rv * Gv * 64 + NP) * 16 + ((size_t)d->mlp_layers * NP + NP + 4) * 4;
    // the asm tile (N = 200): + scratch slot (+ the input exchange of the two-column-set tile), MlpTile::lds_total
    if (gen) pl->lds = ionode::MlpGen::lds_bytes(d->mlp_layers, NT);
    else if (t1) pl->lds = t1deep ? ionode::MlpRow1Deep::lds_bytes(d->mlp_layers) : ionode::MlpRow1::lds_bytes(d->mlp_layers);
    else if (t4) pl->lds = ionode::MlpTile4::lds_bytes(d->mlp_layers);
    else if (Gv == 4 && NT == 13) pl->lds = t32 ? ionode::MlpTile<4, 4, 13, 13, 4>::lds_total(d->mlp_layers) : ionode::MlpTile<4, 4, 13, 13, 0>::lds_total(d->mlp_layers);
    if (t64) plan_lane_wise(pl, (size_t)((d->n_traj + 63) / 64), (vnet ? (size_t)0 : ((pl->lds + 15) & ~(size_t)15)) + (size_t)ionode::LwLds::bytes(2, t64lean));  // MlpTile region + the lane-wise region
  }
  if (!pl->v) { set_err("no kernel variant compiled for this descriptor"); return IONODE_ERR_UNSUPPORTED; }
  if (mlp && d->traj_per_image > 0) {
    // several weight images: a tile reads ONE image (first trajectory / traj_per_image), so an image's trajectories must fill whole
    // tiles.  Tile size from the VARIANT (a lane-wise workgroup is 4 x 64 lanes: the block size says nothing about it).
    const int tile = (pl->v->RT == 64) ? 64 : ((pl->v->tail & 4) && pl->v->G == 4 ? 32 : ((pl->v->tail & 32) ? 1 : ((pl->v->tail & 16) ? 4 : 16)));
    if (d->traj_per_image % tile != 0 || d->mlp_image_stride < (int64_t)ionode_mlp_packed_floats(d->mlp_layers, d->mlp_width)) {
      set_err("traj_per_image must be a multiple of the tile size (16; 64 with tile_waves = 64; 32 with tile_waves = 8) and mlp_image_stride at least one packed image");
      return IONODE_ERR_ARG;
    }
  }
  return IONODE_OK;
}

}  // namespace

namespace ionode {
// Pre-pass of the current / objective epilogue: V(t_k) for every protocol at every requested output time, evaluated ONCE per
// protocol with the same protocol_v() the integrator uses (so the epilogue's values do not change), instead of once per
// trajectory per sample (~35 fp64 vector instructions each: as much as the dense-output polynomial itself).
__global__ void __launch_bounds__(256) ionode_protocol_at_outputs_kernel(const KArgs a, double *__restrict__ v_out) {
  const long long n = (long long)a.P * a.Nt;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int p = (int)(e / a.Nt), k = (int)(e - (long long)p * a.Nt);
    double v;
    protocol_v(a, a.prot_v + (size_t)p * a.Np, a.t_eval[k], v);
    v_out[e] = v;
  }
}
}  // namespace ionode

extern "C" {

int32_t ionode_abi_version(void) { return IONODE_ABI_VERSION; }

const char *ionode_last_error(void) { return g_err; }

// (G, RT) of the kernel variant that serves width N: the fragment stream is laid out per wavefront.
static bool tile_shape(int N, int *G, int *RT) {
  const ionode::Variant *v = find_variant(IONODE_MODEL_NNF, 0, 0, np_of(N) / 16);
  if (!v) return false;
  *G = v->G; *RT = v->RT;
  return true;
}

// 1 KiB fragments per wavefront per hidden layer: F per step, + R on the steps s % G == 0 (K-slices of the remainder tiles)
static size_t frags_per_wave(int NT, int G) {
  const int F = NT / G, R = NT - G * F, NOWN = (NT + G - 1) / G;
  return (size_t)NT * F + (size_t)NOWN * R;
}

// widths without a tuned tile: the image of the run-time-width tile (ionode_device.hpp MlpGen)
static bool generic_width(int N) {
  int G, RT;
  const int NT = np_of(N) / 16;
  return N >= 1 && !tile_shape(N, &G, &RT) && NT >= 2 && NT <= ionode::MlpGen::NT_MAX;
}

size_t ionode_mlp_packed_floats(int32_t L, int32_t N) {
  int G, RT;
  if (L >= 0 && generic_width(N)) return ionode::MlpGen::image_floats(L, np_of(N) / 16);
  if (L < 0 || N < 1 || !tile_shape(N, &G, &RT)) return 0;
  const size_t NP = (size_t)np_of(N), NT = NP / 16;
  // N <= 16: + the scalar section of the per-lane net (row PAIRS: layer 0, then per hidden layer the weights in the canonical k order, bias, pad)
  const size_t npair = (size_t)(N + 1) / 2, pb = (size_t)((2 * (N + 1) + 3) & ~3);
  const size_t scalar = (NT == 1) ? npair * 8 + (size_t)L * npair * pb : 0;
  // N = 200: + the section of the 4-trajectory tile (MlpTile4): its own fragment order and bias float4s
  const size_t tile4 = (NT == 13) ? (size_t)L * ionode::MlpTile4::layer_floats() : 0;
  // ... and behind it the section of the one-trajectory tile (MlpRow1)
  const size_t row1 = (NT == 13) ? (size_t)L * ionode::MlpRow1::layer_floats() : 0;
  return 4 * NP + (size_t)L * ((size_t)G * frags_per_wave((int)NT, G) * 256 + NP) + NP + 4 + scalar + tile4 + row1;
}

int ionode_mlp_pack(const float *w, int32_t L, int32_t N, float *out) {
  int G, RT;
  if (!w || !out || L < 0 || N < 1) { set_err("ionode_mlp_pack: bad argument"); return IONODE_ERR_ARG; }
  if (generic_width(N)) {
    // MlpGen: [NP][4]{b0, w00, w01, 0} | L x ([rt][kt][lane = 16 q + m] float4 over r of W[16 rt + m][16 kt + 4 q + r], then bias[NP]) | wl[NP], bl
    const int NP = np_of(N), NT = NP / 16;
    memset(out, 0, ionode_mlp_packed_floats(L, N) * sizeof(float));
    const float *W0 = w, *b0 = w + (size_t)N * 2;
    for (int r = 0; r < N; ++r) { out[4 * r + 0] = b0[r]; out[4 * r + 1] = W0[2 * r + 0]; out[4 * r + 2] = W0[2 * r + 1]; }
    const float *src = b0 + N;
    float *dst = out + 4 * (size_t)NP;
    for (int l = 0; l < L; ++l) {
      const float *W = src, *b = src + (size_t)N * N;
      for (int rt = 0; rt < NT; ++rt)
        for (int kt = 0; kt < NT; ++kt)
          for (int lane = 0; lane < 64; ++lane) {
            const int m = lane & 15, q = lane >> 4, row = 16 * rt + m;
            float *f = dst + (((size_t)rt * NT + kt) * 64 + lane) * 4;
            for (int r = 0; r < 4; ++r) {
              const int k = 16 * kt + 4 * q + r;
              f[r] = (row < N && k < N) ? W[(size_t)row * N + k] : 0.0f;
            }
          }
      float *bias = dst + (size_t)NT * NT * 256;
      for (int r = 0; r < N; ++r) bias[r] = b[r];
      src += (size_t)N * N + N;
      dst += ionode::MlpGen::layer_floats(NT);
    }
    for (int k = 0; k < N; ++k) dst[k] = src[k];
    dst[NP] = src[N];
    return IONODE_OK;
  }
  if (!tile_shape(N, &G, &RT)) { set_err("ionode_mlp_pack: MLP width outside the supported range (1 <= N <= 512)"); return IONODE_ERR_UNSUPPORTED; }
  const int NP = np_of(N), NT = NP / 16;
  memset(out, 0, ionode_mlp_packed_floats(L, N) * sizeof(float));
  // layer 0: rows {b0, w00, w01, 0}
  const float *W0 = w, *b0 = w + (size_t)N * 2;
  for (int r = 0; r < N; ++r) {
    out[4 * r + 0] = b0[r];
    out[4 * r + 1] = W0[2 * r + 0];
    out[4 * r + 2] = W0[2 * r + 1];
  }
  const float *src = b0 + N;
  float *dst = out + 4 * (size_t)NP;
  const int F = NT / G, Rm = NT - G * F;
  const size_t FR = frags_per_wave(NT, G);
  const size_t frag_floats = (size_t)G * FR * 256;
  for (int l = 0; l < L; ++l) {
    const float *W = src, *b = src + (size_t)N * N;
    // A operand of v_mfma_f32_16x16x4_f32: lane = 16q + m supplies row 16*rt + m, k = 16*kt + 4*q + r.
    // Stream order: wavefront wv | step s, k-tile kt = (s + wv) mod NT | fragments | lane.  F fragments hold the full
    // row tiles wv + i*G k-step-major (element e = r*F + i -> fragment e/4, component e%4); steps with s % G == 0 add
    // one fragment per remainder tile G*F + j (component = k-step r), zero when the step wraps (s + wv >= NT).
    for (int wv = 0; wv < G; ++wv) {
      size_t pos = 0;  // fragment index inside this wavefront's layer stream
      for (int st = 0; st < NT; ++st) {
        const int kt = (st + wv) % NT;
        for (int lane = 0; lane < 64; ++lane) {
          const int m = lane & 15, q = lane >> 4;
          float *f0 = dst + (((size_t)wv * FR + pos) * 64 + lane) * 4;
          // the asm tile's short form of the half-padded k-tile 12 (N <= 200, ionode_device.hpp IONODE_KT12_SHORT): MFMA r = 0 takes
          // k = 192, 196, 193, 197 from the lane groups q = 0..3, MFMA r = 1 takes 194, 198, 195, 199; r = 2, 3 are not executed
          const bool short12 = IONODE_KT12_SHORT && NT == 13 && G == 4 && N <= 200 && kt == 12;
          auto kof = [&](int r) { return short12 ? (r < 2 ? 192 + 4 * (q & 1) + (q >> 1) + 2 * r : N) : 16 * kt + 4 * q + r; };
          for (int e = 0; e < 4 * F; ++e) {
            const int r = e / F, i = e % F, rt = wv + i * G;
            const int row = 16 * rt + m, k = kof(r);
            f0[(size_t)(e / 4) * 256 + e % 4] = (row < N && k < N) ? W[(size_t)row * N + k] : 0.0f;
          }
          if (Rm > 0 && st % G == 0)
            for (int j = 0; j < Rm; ++j)
              for (int r = 0; r < 4; ++r) {
                const int row = 16 * (G * F + j) + m, k = kof(r);
                f0[(size_t)(F + j) * 256 + r] = (st + wv < NT && row < N && k < N) ? W[(size_t)row * N + k] : 0.0f;
              }
        }
        pos += F + ((Rm > 0 && st % G == 0) ? Rm : 0);
      }
    }
    float *bias = dst + frag_floats;
    for (int r = 0; r < N; ++r) bias[r] = b[r];
    src += (size_t)N * N + N;
    dst += frag_floats + NP;
  }
  for (int k = 0; k < N; ++k) dst[k] = src[k];
  dst[NP] = src[N];
  if (NT == 13) {
    // section of the 4-trajectory tile (ionode_device.hpp MlpTile4), round-5 lane layout.  Per layer: wavefronts 0..2: [w][step s][q][lane = 4 b + i]
    // float4 over r of W[row][16 kt + 4 q + r], block b = 4 g + u: row 16 (4 w + g) + 4 u + i, kt = (s + g) mod 13; wavefront 3 (partial chains of
    // the remainder rows): [step j][q][lane] float4 over r of W[192 + 4 u + i][16 (c + 4 j) + 4 q + r] with c = g, -0.0f where c + 4 j > 12; then per
    // (wavefront, lane) the accumulator start float4 {bias of rows i = 0..3 of the lane's block} (chains c > 0: 0)
    float *t4 = dst + NP + 4;
    const float *lsrc = b0 + N;
    for (int l = 0; l < L; ++l) {
      const float *W = lsrc, *b = lsrc + (size_t)N * N;
      float *lay = t4 + (size_t)l * ionode::MlpTile4::layer_floats();
      for (int wv = 0; wv < 3; ++wv)
        for (int st = 0; st < 13; ++st)
          for (int q = 0; q < 4; ++q)
            for (int lane = 0; lane < 64; ++lane) {
              const int i = lane & 3, bb = lane >> 2, g = bb >> 2, u = bb & 3;
              float *f = lay + ((((size_t)wv * 13 + st) * 4 + q) * 64 + lane) * 4;
              const int row = 16 * (4 * wv + g) + 4 * u + i, kt = (st + g) % 13;
              for (int r = 0; r < 4; ++r) {
                const int k = 16 * kt + 4 * q + r;
                f[r] = (row < N && k < N) ? W[(size_t)row * N + k] : 0.0f;
              }
            }
      float *rem = lay + (size_t)3 * 13 * 4 * 256;
      for (int jj = 0; jj < 4; ++jj)
        for (int q = 0; q < 4; ++q)
          for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 3, bb = lane >> 2, c = bb >> 2, u = bb & 3;
            float *f = rem + (((size_t)jj * 4 + q) * 64 + lane) * 4;
            const int row = 192 + 4 * u + i, kt = c + 4 * jj;
            for (int r = 0; r < 4; ++r) {
              const int k = 16 * kt + 4 * q + r;
              f[r] = (kt >= 13) ? -0.0f : ((row < N && k < N) ? W[(size_t)row * N + k] : 0.0f);
            }
          }
      float *bias = rem + (size_t)4 * 4 * 256;
      for (int wv = 0; wv < 4; ++wv)
        for (int lane = 0; lane < 64; ++lane) {
          const int bb = lane >> 2, g = bb >> 2, u = bb & 3;
          for (int i = 0; i < 4; ++i) {
            const int row = (wv < 3) ? 16 * (4 * wv + g) + 4 * u + i : 192 + 4 * u + i;
            bias[((size_t)wv * 64 + lane) * 4 + i] = (row < N && (wv < 3 || g == 0)) ? b[row] : 0.0f;
          }
        }
      lsrc += (size_t)N * N + N;
    }
    // section of the one-trajectory tile (ionode_device.hpp MlpRow1), behind the 4-trajectory tile's.  Per layer: wavefronts 0..2: [w][step s]
    // [r][lane] float4 over q of W[64 w + lane][16 ((s + lane / 16) mod 13) + 4 q + r]; wavefront 3 (partial chains of the remainder rows):
    // [step j][r][lane = 16 c + i] float4 over q of W[192 + i][16 (c + 4 j) + 4 q + r], -0.0f where c + 4 j > 12; then per (wavefront, lane) the
    // accumulator start (the row's bias; chains c > 0: 0)
    float *r1 = t4 + (size_t)L * ionode::MlpTile4::layer_floats();
    lsrc = b0 + N;
    for (int l = 0; l < L; ++l) {
      const float *W = lsrc, *b = lsrc + (size_t)N * N;
      float *lay = r1 + (size_t)l * ionode::MlpRow1::layer_floats();
      for (int wv = 0; wv < 3; ++wv)
        for (int st = 0; st < 13; ++st)
          for (int r = 0; r < 4; ++r)
            for (int lane = 0; lane < 64; ++lane) {
              float *f = lay + ((((size_t)wv * 13 + st) * 4 + r) * 64 + lane) * 4;
              const int row = 64 * wv + lane, kt = (st + (lane >> 4)) % 13;
              for (int q = 0; q < 4; ++q) {
                const int k = 16 * kt + 4 * q + r;
                f[q] = (row < N && k < N) ? W[(size_t)row * N + k] : 0.0f;
              }
            }
      float *rem = lay + (size_t)3 * 13 * 4 * 256;
      for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 4; ++r)
          for (int lane = 0; lane < 64; ++lane) {
            float *f = rem + (((size_t)j * 4 + r) * 64 + lane) * 4;
            const int row = 192 + (lane & 15), kt = (lane >> 4) + 4 * j;
            for (int q = 0; q < 4; ++q) {
              const int k = 16 * kt + 4 * q + r;
              f[q] = (kt >= 13) ? -0.0f : ((row < N && k < N) ? W[(size_t)row * N + k] : 0.0f);
            }
          }
      float *bias = rem + (size_t)4 * 4 * 256;
      for (int wv = 0; wv < 4; ++wv)
        for (int lane = 0; lane < 64; ++lane) {
          const int row = (wv < 3) ? 64 * wv + lane : 192 + (lane & 15);
          bias[wv * 64 + lane] = (row < N && (wv < 3 || lane < 16)) ? b[row] : 0.0f;
        }
      lsrc += (size_t)N * N + N;
    }
  }
  if (NT == 1) {
    // scalar section (ionode_device.hpp MlpLane), in row PAIRS (2 m, 2 m + 1) -- the two halves of a v_pk_fma_f32; an odd N's last
    // pair has a zero second row.  Layer 0: {b0, b0'} {w00, w00'} {w01, w01'} {0, 0}.  Hidden layer l, pair m: {W[2m][k], W[2m+1][k]}
    // for k = 4 q + r < N in the order r-major / q-minor (the order in which the 16 x 16 x 4 MFMA tile accumulates them), then the
    // biases as bias + 0.0f (a -0 bias becomes +0: see MlpLane), pad to a multiple of 4 floats.
    const int NPAIR = (N + 1) / 2, PB = (2 * (N + 1) + 3) & ~3;
    float *sc = dst + NP + 4;
    for (int i = 0; i < NPAIR * 8 + L * NPAIR * PB; ++i) sc[i] = 0.0f;
    for (int m = 0; m < NPAIR; ++m)
      for (int e = 0; e < 2; ++e) {
        const int j = 2 * m + e;
        if (j >= N) continue;
        sc[m * 8 + 0 + e] = b0[j];
        sc[m * 8 + 2 + e] = W0[2 * j + 0];
        sc[m * 8 + 4 + e] = W0[2 * j + 1];
      }
    float *sh = sc + NPAIR * 8;
    const float *lsrc = b0 + N;
    for (int l = 0; l < L; ++l) {
      const float *W = lsrc, *b = lsrc + (size_t)N * N;
      for (int m = 0; m < NPAIR; ++m)
        for (int e = 0; e < 2; ++e) {
          const int j = 2 * m + e;
          if (j >= N) continue;
          float *blk = sh + ((size_t)l * NPAIR + m) * PB;
          int pos = 0;
          for (int r = 0; r < 4; ++r)
            for (int q = 0; q < 4; ++q)
              if (4 * q + r < N) blk[2 * (pos++) + e] = W[(size_t)j * N + 4 * q + r];
          blk[2 * N + e] = (N < 16) ? b[j] + 0.0f : b[j];
        }
      lsrc += (size_t)N * N + N;
    }
  }
  return IONODE_OK;
}

int ionode_launch_geometry(const ionode_desc *d, int32_t out[4]) {
  Plan pl;
  const int rc = make_plan(d, &pl);
  if (rc != IONODE_OK) return rc;
  out[0] = (int32_t)pl.grid;
  out[1] = (int32_t)pl.block;
  out[2] = (int32_t)pl.lds;
  out[3] = pl.v->G;
  return IONODE_OK;
}

const char *ionode_kernel_name(const ionode_desc *d) {
  Plan pl;
  // the descriptor does not say whether i_out will be passed: a table or a fused objective implies the epilogue
  if (make_plan(d, &pl, d && (d->sse_out != nullptr || d->v_at_outputs != nullptr)) != IONODE_OK) return "";
  return pl.v->name;
}

const char *ionode_last_kernel_name(void) { return g_last_kernel; }

int32_t ionode_lane_wise_from(int32_t model, int32_t mlp_width) { return lane_wise_from(model, mlp_width); }

int ionode_dopri5(const ionode_desc *d, const float *mlp_packed, const double *params, const double *prot_v,
                  const double *prot_t, const int32_t *prot_of_traj, const void *y0, const double *t_eval,
                  void *y_out, double *i_out, int32_t *status, int64_t *stats, void *stream) {
  Plan pl;
  const int rc = make_plan(d, &pl, i_out != nullptr || d->sse_out != nullptr, prot_t != nullptr);
  if (rc != IONODE_OK) return rc;
  const bool mlp = d->model == IONODE_MODEL_NNF || d->model == IONODE_MODEL_NND;
  if (!params || !prot_v || !y0 || !t_eval || (!y_out && !d->sse_out) || !status || (mlp && !mlp_packed)) {
    set_err("ionode_dopri5: required buffer is NULL");
    return IONODE_ERR_ARG;
  }
  if (d->sse_out && (!d->sse_ref || !(d->t_eval_dt_hint > 0.0) || d->n_out < 2)) {
    set_err("fused objective: sse_out needs sse_ref and the output-grid hint (t_eval_dt_hint > 0)");
    return IONODE_ERR_ARG;
  }
  ionode::KArgs a;
  memset(&a, 0, sizeof a);
  a.mlp = mlp_packed; a.params = params; a.prot_v = prot_v; a.prot_t = prot_t; a.prot_of_traj = prot_of_traj;
  a.y0 = y0; a.t_eval = t_eval; a.y_out = y_out; a.i_out = i_out; a.status = status; a.stats = stats;
  a.B = d->n_traj; a.Nt = d->n_out; a.P = d->n_prot; a.Np = d->prot_n; a.n_params = d->n_params;
  if (mlp) { a.L = d->mlp_layers; a.N = d->mlp_width; a.NP = np_of(d->mlp_width); a.NT = a.NP / 16; }
  a.max_steps = d->max_steps > 0 ? d->max_steps : (int64_t)2147483647;  // torchdiffeq max_num_steps default 2**31 - 1
  a.max_total = d->max_total_steps > 0 ? d->max_total_steps
                                       : (d->max_total_steps == 0 ? (int64_t)IONODE_DEFAULT_MAX_TOTAL_STEPS : INT64_MAX);
  a.ckpt = d->ckpt; a.ckpt_cap = d->ckpt ? d->ckpt_cap : 0;
  if (d->ckpt && d->ckpt_cap < 1) { set_err("ckpt given with ckpt_cap < 1"); return IONODE_ERR_ARG; }
  a.prot_rdt = 1.0 / d->prot_dt;  // correctly rounded: the kernels divide by prot_dt through div_by()
  a.dt_max = d->max_step > 0.0 ? d->max_step : __builtin_inf();
  a.prot_t0 = d->prot_t0; a.prot_dt = d->prot_dt; a.v_oob = d->v_oob; a.rtol = d->rtol; a.atol = d->atol;
  a.obs_g = d->obs_g; a.obs_e = d->obs_e; a.obs_open = d->obs_open_state_only;
  a.step_log = d->step_log; a.step_log_cap = d->step_log ? d->step_log_cap : 0;
  a.sse_ref = d->sse_ref; a.sse_out = d->sse_out; a.v_tab = d->v_at_outputs;
  if (mlp && d->traj_per_image > 0) { a.mlp_stride = d->mlp_image_stride; a.traj_per_img = d->traj_per_image; }  // checked in make_plan
  if (d->launch_order) {
    if (a.traj_per_img > 0) { set_err("launch_order cannot be combined with traj_per_image (tiles of an image must stay together)"); return IONODE_ERR_ARG; }
    a.order = d->launch_order;
  }
  a.lw_bytes = (int32_t)pl.lw_bytes;
  a.te_t0 = d->t_eval_t0_hint; a.te_dt = (d->t_eval_dt_hint > 0.0 && d->n_out > 1) ? d->t_eval_dt_hint : 0.0;
  a.te_rdt = a.te_dt > 0.0 ? 1.0 / a.te_dt : 0.0;
  a.te_exact = (a.te_dt > 0.0 && d->t_eval_exact) ? 1 : 0;
  const hipError_t e = pl.v->fn(a, pl.grid, pl.lds, reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) { set_err("kernel launch failed: %s", hipGetErrorString(e)); return IONODE_ERR_LAUNCH; }
  g_last_kernel = pl.v->name;
  return IONODE_OK;
}

int ionode_protocol_at_outputs(const ionode_desc *d, const double *prot_v, const double *prot_t, const double *t_eval,
                               double *v_out, void *stream) {
  if (!d || !prot_v || !t_eval || !v_out) { set_err("ionode_protocol_at_outputs: required buffer is NULL"); return IONODE_ERR_ARG; }
  if (d->n_out < 1 || d->n_prot < 1 || d->prot_n < 2 || !(d->prot_dt > 0)) { set_err("ionode_protocol_at_outputs: empty grid / protocol"); return IONODE_ERR_ARG; }
  ionode::KArgs a;
  memset(&a, 0, sizeof a);
  a.prot_v = prot_v; a.prot_t = prot_t; a.t_eval = t_eval; a.Nt = d->n_out; a.P = d->n_prot; a.Np = d->prot_n;
  a.prot_t0 = d->prot_t0; a.prot_dt = d->prot_dt; a.prot_rdt = 1.0 / d->prot_dt; a.v_oob = d->v_oob;
  const long long n = (long long)a.P * a.Nt;
  const unsigned grid = (unsigned)((n + 255) / 256 > 65536 ? 65536 : (n + 255) / 256);
  hipLaunchKernelGGL(ionode::ionode_protocol_at_outputs_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, v_out);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_err("kernel launch failed: %s", hipGetErrorString(e)); return IONODE_ERR_LAUNCH; }
  return IONODE_OK;
}

}  // extern "C"
