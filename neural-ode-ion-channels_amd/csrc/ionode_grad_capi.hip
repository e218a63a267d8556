// ionode_grad_capi.hip -- C ABI of the gradient path (include/ionode.h, "gradients through the solve"): the grad image
// packer, the backward-sweep launcher and the weight-gradient reduction launcher.
#include <cstdio>
#include <cstring>

#include "ionode_grad_launch.hpp"
#include "ionode_grad_reduce.hpp"

namespace {

thread_local char g_gerr[256] = "";
void gerr(const char *m) { snprintf(g_gerr, sizeof g_gerr, "%s", m); }

inline int np_of(int N) { return 16 * ((N + 15) / 16); }

using ionode::SweepFn;
using ionode::pick_sweep;
using ionode::launch_sweep;

// the widths of architectures/s00-s11.py: N = 10, 100, 200, 500
SweepFn find_sweep(int model, int f32, int NT) {
  switch (NT) {
    case 1: return pick_sweep<1>(model, f32);
    case 7: return pick_sweep<7>(model, f32);
    case 13: return pick_sweep<13>(model, f32);
    case 32: return ionode::pick_sweep32(model, f32);   // inst_grad32.hip
    default: return nullptr;
  }
}

}  // namespace

extern "C" {

const char *ionode_grad_last_error(void) { return g_gerr; }

size_t ionode_grad_image_floats(int32_t L, int32_t N) {
  if (L < 1 || N < 1) return 0;
  return ionode::grad_img_floats(L, np_of(N) / 16);
}

size_t ionode_grad_record_floats(int32_t L, int32_t N) {
  if (L < 1 || N < 1) return 0;
  return (size_t)ionode::grad_record_floats(L, np_of(N) / 16);
}

int ionode_grad_pack(const float *w, int32_t L, int32_t N, float *out) {
  if (!w || !out || L < 1 || N < 1) { gerr("ionode_grad_pack: bad argument"); return IONODE_ERR_ARG; }
  const int NP = np_of(N), NT = NP / 16;
  memset(out, 0, ionode::grad_img_floats(L, NT) * sizeof(float));
  const float *W0 = w, *b0 = w + (size_t)N * 2;
  for (int r = 0; r < N; ++r) {
    out[4 * r + 0] = b0[r];
    out[4 * r + 1] = W0[2 * r + 0];
    out[4 * r + 2] = W0[2 * r + 1];
  }
  const float *src = b0 + N;
  float *bias = out + ionode::grad_img_bias(NT);
  float *fw = out + ionode::grad_img_fwd(L, NT), *bw = out + ionode::grad_img_bwd(L, NT);
  for (int l = 0; l < L; ++l) {
    const float *W = src, *b = src + (size_t)N * N;
    for (int r = 0; r < N; ++r) bias[(size_t)l * NP + r] = b[r];
    // A operand of v_mfma_f32_16x16x4_f32, k-step r: lane = 16*kq + m supplies A[m][kq]; with the forward kernel's
    // k-permutation that is row 16*rt + m, contraction index 16*kt + 4*kq + r.  Transposed section: W^T.
    for (int rt = 0; rt < NT; ++rt)
      for (int kt = 0; kt < NT; ++kt)
        for (int lane = 0; lane < 64; ++lane) {
          const int m = lane & 15, kq = lane >> 4;
          const size_t f = ((((size_t)l * NT + rt) * NT + kt) * 64 + lane) * 4;
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * rt + m, k = 16 * kt + 4 * kq + r;
            const bool in = row < N && k < N;
            fw[f + r] = in ? W[(size_t)row * N + k] : 0.0f;
            bw[f + r] = in ? W[(size_t)k * N + row] : 0.0f;
          }
        }
    src += (size_t)N * N + N;
  }
  float *wl = out + ionode::grad_img_wl(L, NT);
  for (int k = 0; k < N; ++k) wl[k] = src[k];
  wl[NP] = src[N];
  return IONODE_OK;
}

// mode 0: one-phase sweep; 1: phase A (forward recompute of every (tile, step), records + sign words); 2: phase B (the walk)
static int backward_impl(int mode, const ionode_desc *d, int32_t it_begin, int32_t it_end, int32_t n_iter, const float *grad_image,
                         const double *params, const double *prot_v, const double *prot_t, const int32_t *prot_of_traj,
                         const double *t_eval, const int32_t *n_accepted, const void *grad_y, double *state,
                         float *records, double *packets, double *grad_params, double *grad_y0, void *stream) {
  if (!d) { gerr("null descriptor"); return IONODE_ERR_ARG; }
  if (mode != 0 && (!packets || (d->model != IONODE_MODEL_NNF && d->model != IONODE_MODEL_NND))) {
    gerr("two-phase sweep: NN-f / NN-d only, `packets` required"); return IONODE_ERR_ARG;
  }
  if (mode == 1) {   // phase A carries no adjoint state: stand-ins so that the shared checks pass (never dereferenced)
    static double dummy;
    state = &dummy; grad_params = &dummy; grad_y0 = &dummy;
  }
  const bool m6 = d->model == IONODE_MODEL_MARKOV6;
  const bool hh2 = d->model == IONODE_MODEL_HH2 || m6;  // closed-form models: no MLP image, no records
  if (d->model < 0 || d->model > 3) { gerr("backward sweep: unknown model"); return IONODE_ERR_UNSUPPORTED; }
  if (d->n_state != (m6 ? 6 : 2) || d->n_traj < 1 || d->n_out < 1 || d->n_prot < 1 || d->prot_n < 2 || d->n_params < (m6 ? 12 : 8) || !(d->prot_dt > 0)) {
    gerr("inconsistent descriptor"); return IONODE_ERR_ARG;
  }
  if ((!grad_image && !hh2) || !params || !prot_v || !t_eval || !n_accepted || !grad_y || !state || !grad_params || !grad_y0 || !d->ckpt || d->ckpt_cap < 1) {
    gerr("ionode_dopri5_backward: required buffer is NULL (ckpt / ckpt_cap come from the descriptor)"); return IONODE_ERR_ARG;
  }
  if (it_begin < 0 || it_end <= it_begin || it_end > n_iter) { gerr("bad iteration range"); return IONODE_ERR_ARG; }
  if (mode == 1 && (int64_t)it_end - it_begin > (int64_t)65535 * ionode::GRAD_RECOMPUTE_IB) {
    gerr("ionode_dopri5_backward_recompute: at most 65535 x 4 iterations per launch (HIP's grid.y limit): split the range");
    return IONODE_ERR_ARG;
  }
  if (d->traj_per_image > 0) { gerr("backward sweep: one weight set per launch (traj_per_image must be 0)"); return IONODE_ERR_UNSUPPORTED; }
  if (!hh2 && (d->mlp_layers < 1 || d->mlp_width < 1)) { gerr("bad MLP shape"); return IONODE_ERR_ARG; }
  const int NP = hh2 ? 16 : np_of(d->mlp_width), NT = NP / 16, L = hh2 ? 0 : d->mlp_layers;
  SweepFn fn = m6 ? (d->state_f32 ? &launch_sweep<IONODE_MODEL_MARKOV6, float, 1> : &launch_sweep<IONODE_MODEL_MARKOV6, double, 1>)
               : hh2 ? (d->state_f32 ? &launch_sweep<IONODE_MODEL_HH2, float, 1> : &launch_sweep<IONODE_MODEL_HH2, double, 1>)
                     : find_sweep(d->model, d->state_f32 ? 1 : 0, NT);
  const size_t lds = hh2 ? (size_t)16 * 5 * (m6 ? 6 : 2) * 8 : ionode::grad_lds_bytes(L, NT);
  if (!fn || lds > 160 * 1024 || L > 15) {
    gerr("backward sweep: (L, N) outside the compiled variants (N pads to 16, 112, 208 or 512; at most 15 hidden layers)");
    return IONODE_ERR_UNSUPPORTED;
  }
  ionode::GArgs a;
  memset(&a, 0, sizeof a);
  a.k.params = params; a.k.prot_v = prot_v; a.k.prot_t = prot_t; a.k.prot_of_traj = prot_of_traj; a.k.t_eval = t_eval;
  a.k.B = d->n_traj; a.k.Nt = d->n_out; a.k.P = d->n_prot; a.k.Np = d->prot_n; a.k.n_params = d->n_params;
  a.k.L = L; a.k.N = d->mlp_width; a.k.NP = NP; a.k.NT = NT;
  a.k.prot_t0 = d->prot_t0; a.k.prot_dt = d->prot_dt; a.k.prot_rdt = 1.0 / d->prot_dt; a.k.v_oob = d->v_oob;
  a.img = grad_image; a.ckpt = d->ckpt; a.ckpt_cap = d->ckpt_cap; a.nacc = n_accepted; a.grad_y = grad_y; a.state = state;
  a.records = hh2 ? nullptr : records; a.grad_params = grad_params; a.grad_y0 = grad_y0;
  a.it_begin = it_begin; a.it_end = it_end; a.n_iter = n_iter;
  a.record_floats = ionode::grad_record_floats(L, NT);
  a.packets = mode != 0 ? packets : nullptr;
  a.phase = mode;
  if (mode != 0 && ionode::grad_lds_bytes(L, NT) + 16 + 16 * ionode::GRAD_PACKET * 8 > 160 * 1024) { gerr("two-phase sweep: LDS"); return IONODE_ERR_UNSUPPORTED; }
  fn(a, (unsigned)((d->n_traj + 15) / 16), lds, reinterpret_cast<hipStream_t>(stream));
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { gerr(hipGetErrorString(e)); return IONODE_ERR_LAUNCH; }
  return IONODE_OK;
}

int ionode_dopri5_backward(const ionode_desc *d, int32_t it_begin, int32_t it_end, int32_t n_iter, const float *grad_image,
                           const double *params, const double *prot_v, const double *prot_t, const int32_t *prot_of_traj,
                           const double *t_eval, const int32_t *n_accepted, const void *grad_y, double *state,
                           float *records, double *grad_params, double *grad_y0, void *stream) {
  return backward_impl(0, d, it_begin, it_end, n_iter, grad_image, params, prot_v, prot_t, prot_of_traj, t_eval, n_accepted, grad_y,
                       state, records, nullptr, grad_params, grad_y0, stream);
}

size_t ionode_grad_packet_doubles(void) { return (size_t)16 * ionode::GRAD_PACKET; }

int ionode_dopri5_backward_recompute(const ionode_desc *d, int32_t it_begin, int32_t it_end, int32_t n_iter, const float *grad_image,
                                     const double *params, const double *prot_v, const double *prot_t, const int32_t *prot_of_traj,
                                     const double *t_eval, const int32_t *n_accepted, const void *grad_y, float *records,
                                     double *packets, void *stream) {
  return backward_impl(1, d, it_begin, it_end, n_iter, grad_image, params, prot_v, prot_t, prot_of_traj, t_eval, n_accepted, grad_y,
                       nullptr, records, packets, nullptr, nullptr, stream);
}

int ionode_dopri5_backward_sweep(const ionode_desc *d, int32_t it_begin, int32_t it_end, int32_t n_iter, const float *grad_image,
                                 const double *params, const double *prot_v, const double *prot_t, const int32_t *prot_of_traj,
                                 const double *t_eval, const int32_t *n_accepted, const void *grad_y, double *state,
                                 float *records, const double *packets, double *grad_params, double *grad_y0,
                                 void *stream) {
  return backward_impl(2, d, it_begin, it_end, n_iter, grad_image, params, prot_v, prot_t, prot_of_traj, t_eval, n_accepted, grad_y,
                       state, records, const_cast<double *>(packets), grad_params, grad_y0, stream);
}

static int reduce_impl(int32_t L, int32_t N, const float *records, int64_t n_records, int32_t n_slabs, float *partials, void *stream,
                       int unit_seed) {
  if (!records || !partials || n_records < 1 || n_slabs < 1 || L < 1 || N < 1) { gerr("ionode_grad_reduce: bad argument"); return IONODE_ERR_ARG; }
  const int NT = np_of(N) / 16;
  const hipError_t e = ionode::launch_grad_reduce(L, NT, records, n_records, n_slabs, partials, reinterpret_cast<hipStream_t>(stream), unit_seed);
  if (e == hipErrorInvalidValue) { gerr("ionode_grad_reduce: width outside the compiled variants"); return IONODE_ERR_UNSUPPORTED; }
  if (e != hipSuccess) { gerr(hipGetErrorString(e)); return IONODE_ERR_LAUNCH; }
  return IONODE_OK;
}

int32_t ionode_grad_reduce_slabs(int32_t L, int32_t N, int64_t n_records) {
  if (L < 1 || N < 1 || n_records < 1) return 1;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) {
    (void)hipGetLastError();
    cus = 256;   // (no device: the plan of an unpartitioned MI355X)
  }
  return ionode::grad_reduce_slabs(L, np_of(N) / 16, cus, n_records);
}

int ionode_grad_reduce(int32_t L, int32_t N, const float *records, int64_t n_records, int32_t n_slabs, float *partials,
                       void *stream) {
  return reduce_impl(L, N, records, n_records, n_slabs, partials, stream, 0);
}

int ionode_grad_reduce_unit(int32_t L, int32_t N, const float *records, int64_t n_records, int32_t n_slabs, float *partials,
                            void *stream) {
  return reduce_impl(L, N, records, n_records, n_slabs, partials, stream, 1);
}

int ionode_regress_step(int32_t L, int32_t N, const float *grad_image, const float *x, const float *offset, const float *y,
                        int32_t n_rows, float netscale, float *records, double *loss_partials, int32_t n_workgroups,
                        void *stream) {
  if (!grad_image || !x || !y || !records || !loss_partials || n_rows < 1 || n_workgroups < 1 || L < 1 || N < 1) {
    gerr("ionode_regress_step: bad argument"); return IONODE_ERR_ARG;
  }
  const int NT = np_of(N) / 16;
  if (ionode::grad_lds_bytes(L, NT) > 160 * 1024 || L > 15) { gerr("ionode_regress_step: (L, N) outside the compiled variants (at most 15 hidden layers)"); return IONODE_ERR_UNSUPPORTED; }
  ionode::RArgs a;
  memset(&a, 0, sizeof a);
  a.img = grad_image; a.x = x; a.y = y; a.offset = offset; a.records = records; a.loss_part = loss_partials;
  a.M = n_rows; a.L = L; a.N = N; a.NT = NT; a.record_floats = ionode::grad_record_floats(L, NT); a.netscale = netscale;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (NT) {
    case 1: ionode::launch_regress<1>(a, (unsigned)n_workgroups, s); break;
    case 7: ionode::launch_regress<7>(a, (unsigned)n_workgroups, s); break;
    case 13: ionode::launch_regress<13>(a, (unsigned)n_workgroups, s); break;
    case 32: ionode::launch_regress32(a, (unsigned)n_workgroups, s); break;
    default: gerr("ionode_regress_step: width outside the compiled variants (N pads to 16, 112, 208 or 512)"); return IONODE_ERR_UNSUPPORTED;
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { gerr(hipGetErrorString(e)); return IONODE_ERR_LAUNCH; }
  return IONODE_OK;
}

int ionode_adam_step(int32_t n_params, int32_t n_slabs, int32_t L, int32_t N, const float *partials, const int32_t *padmap,
                     float *weights, float *exp_avg, float *exp_avg_sq, float lr, float beta1, float beta2, float eps,
                     int32_t step, float *grad_out, int32_t apply, void *stream) {
  if (!partials || !padmap || n_params < 1 || n_slabs < 1 || step < 1 || (apply && (!weights || !exp_avg || !exp_avg_sq))) {
    gerr("ionode_adam_step: bad argument"); return IONODE_ERR_ARG;
  }
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
  const size_t partf = ionode::grad_partial_floats(L, np_of(N) / 16);
  hipLaunchKernelGGL(ionode::ionode_adam_kernel, dim3((n_params + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     n_params, n_slabs, partf, partials, padmap, weights, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, bc1, bc2_sqrt,
                     grad_out, apply);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { gerr(hipGetErrorString(e)); return IONODE_ERR_LAUNCH; }
  return IONODE_OK;
}

int ionode_image_refresh(int32_t L, int32_t N, const int32_t *image_map, const float *weights, float *grad_image, void *stream) {
  if (!image_map || !weights || !grad_image || L < 1 || N < 1) { gerr("ionode_image_refresh: bad argument"); return IONODE_ERR_ARG; }
  const size_t n = ionode::grad_img_floats(L, np_of(N) / 16);
  hipLaunchKernelGGL(ionode::ionode_image_refresh_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), n, image_map, weights, grad_image);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { gerr(hipGetErrorString(e)); return IONODE_ERR_LAUNCH; }
  return IONODE_OK;
}

size_t ionode_grad_partial_floats(int32_t L, int32_t N) {
  if (L < 1 || N < 1) return 0;
  return ionode::grad_partial_floats(L, np_of(N) / 16);
}

}  // extern "C"
