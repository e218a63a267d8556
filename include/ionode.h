/*
 * ionode.h -- C ABI of libionode.so: the MI355X-native batched neural-ODE integrator for
 * ion-channel gating models (adaptive Dormand-Prince RK45 + dense output, HIP, gfx950).
 *
 * What it replaces.  The reference (chonlei/neural-ode-ion-channels) has no FFI: its hot path is the
 * Python call
 *     odeint(func, y0, t)                      train-s1.py:322,327  train-d0.py:428  train-r1.py:273 ...
 * into torchdiffeq==0.2.1 (requirements.txt:1), which calls back func.forward(t, y) six times per
 * step (train-s1.py:231-247 NN-f, train-d2.py:257-272 NN-d, train-s1.py:161-177 HH,
 * train-d1.py:165-187 6-state).  The entry points below are what a binding for that call binds:
 * ONE launch integrates a whole batch of independent trajectories, with the RHS fused into the
 * kernel.  Caller owns every buffer; the library allocates nothing and keeps no state.
 *
 *   reference interface                                    entry point here
 *   -----------------------------------------------------  --------------------------------------
 *   odeint(func, y0, t, rtol, atol, method='dopri5')       ionode_dopri5()
 *   func.net state_dict  (net.{0,2,..}.weight/.bias)       ionode_mlp_packed_floats()/ionode_mlp_pack()
 *   func.set_fixed_form_voltage_protocol(t, v)             prot_v / prot_t / prot_t0 / prot_dt arguments
 *   pred_y[:,0,0]*pred_y[:,0,1]*(func._v(t)+86)            optional fused epilogue -> i_out
 *       (train-s1.py:328, train-r1.py:274, train-d1.py:299)
 *   AssertionError 'underflow in dt' / 'non-finite' /      per-trajectory status[] codes
 *       'max_num_steps exceeded' (torchdiffeq)
 *
 * All pointers passed to ionode_dopri5 are DEVICE pointers (HBM), except the descriptor.
 */
#ifndef IONODE_H
#define IONODE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IONODE_ABI_VERSION 9

/* RHS families (func.forward variants of the reference) */
#define IONODE_MODEL_HH2 0     /* 2-state Hodgkin-Huxley: Lambda, train-s1.py:134-177; candidate ODEFunc train-d0.py:321-374 */
#define IONODE_MODEL_MARKOV6 1 /* 6-state Markov: Lambda, train-d1.py:134-187 */
#define IONODE_MODEL_NNF 2     /* NN-f: da/dt = MLP(V/100, a)/1000, train-s1.py:181-247 */
#define IONODE_MODEL_NND 3     /* NN-d: da/dt = HH + MLP(V/100, a)/1000, train-d2.py:191-272 */

/* per-trajectory status (torchdiffeq's assertion messages) */
#define IONODE_STATUS_OK 0
#define IONODE_STATUS_DT_UNDERFLOW 1 /* 'underflow in dt' */
#define IONODE_STATUS_NONFINITE 2    /* 'non-finite values in state `y`' */
#define IONODE_STATUS_MAX_STEPS 3    /* 'max_num_steps exceeded' */

/* return codes of the entry points */
#define IONODE_OK 0
#define IONODE_ERR_ARG (-1)      /* inconsistent descriptor / NULL where a buffer is required */
#define IONODE_ERR_UNSUPPORTED (-2) /* shape outside what the kernels are built for */
#define IONODE_ERR_LAUNCH (-3)   /* hipLaunchKernel failed; see ionode_last_error() */

typedef struct ionode_desc {
  int32_t model;       /* IONODE_MODEL_* */
  int32_t state_f32;   /* 1: solver state (y, k, dense output) in fp32 = the reference scripts' y0.dtype;
                          0: fp64 state (BASELINE configs 2-3).  Time/step control is fp64 either way. */
  int32_t n_state;     /* D: 2 (HH2, NNF, NND) or 6 (MARKOV6) */
  int32_t n_out;       /* len(t) */
  int32_t n_traj;      /* B */
  int32_t n_prot;      /* P voltage protocols of prot_n samples each */
  int32_t prot_n;      /* samples per protocol */
  int32_t mlp_layers;  /* L: hidden N x N Linear layers (architectures/sNN.py: n_layers) */
  int32_t mlp_width;   /* N: nodes per layer (architectures/sNN.py: n_nodes) */
  int32_t n_params;    /* doubles per trajectory in params[]: >= 8 (p1..p8), >= 12 for MARKOV6 */
  int64_t max_steps;   /* torchdiffeq's max_num_steps: step attempts allowed between two emitted outputs (its _advance
                          resets the counter for every output time); 0 = torchdiffeq's default 2**31 - 1 */
  double prot_t0;      /* uniform protocol grid t_i = prot_t0 + i*prot_dt (used when prot_t == NULL) */
  double prot_dt;
  double v_oob;        /* voltage substituted outside the protocol's time range: -80 (train-s1.py:237) */
  double rtol, atol;   /* torchdiffeq defaults 1e-7 / 1e-9; the reference never overrides them */
  double obs_g;        /* observation epilogue i = g * gate * (V(t_k) - obs_e); gate = y0*y1 or y[D-1] */
  double obs_e;
  int32_t obs_open_state_only; /* 1: gate = last state (6-state O, train-d1.py:299) */
  int32_t tile_waves;  /* tuning, 0 = auto.  MLP models: wavefronts cooperating on one 16-trajectory tile (1, 4); N <= 16 nets
                          also 64 = one trajectory per lane, 64 per wavefront (auto from ionode_lane_wise_from() trajectories); N = 200 also 8 / 2 / 16 = 32 / 4 / 1
                          trajectories per tile (auto: 1 up to 256 trajectories, 4 up to 1024, 16 up to 8191, 32 beyond).  Closed-form models: trajectories per wavefront (64 or 16; auto = 16 below ionode_lane_wise_from()) */
  double *step_log;    /* optional DEVICE buffer [step_log_cap][4] fp64: (t0, dt, error ratio, accepted) of every
                          step attempt of trajectory 0 -- the per-step trace parity tests compare; NULL = off */
  int64_t step_log_cap;
  double t_eval_t0_hint; /* optional hint t_eval[k] ~ t0_hint + k*dt_hint (dt_hint <= 0: none).  Only a starting guess for */
  double t_eval_dt_hint; /* the output cursor, verified against t_eval in the kernel: a wrong hint costs time, not results */
  int64_t max_total_steps; /* runaway bound: step attempts allowed over the whole solve of one trajectory (a tile runs until
                          its slowest trajectory ends; the reference bounds a solve with a 600 s SIGALRM, train-d0.py:309-318).
                          0 = library default IONODE_DEFAULT_MAX_TOTAL_STEPS; < 0 = unbounded.  Exceeding it gives
                          IONODE_STATUS_MAX_STEPS like max_steps. */
  double *ckpt;        /* optional DEVICE buffer [n_traj][ckpt_cap][4 + 8*n_state] fp64, one record per ACCEPTED step:
                          {t0, dt, first output index of the step, outputs emitted, y[D], k1..k7[D]} -- what the backward
                          sweep (ionode_dopri5_backward) replays.  Steps beyond ckpt_cap are not recorded (the caller
                          compares stats[b][0] with ckpt_cap and retries with a larger buffer).  NULL = off */
  int32_t ckpt_cap;
  int32_t t_eval_exact; /* 1: the caller has VERIFIED t_eval[k] == t_eval_t0_hint + (double)k * t_eval_dt_hint bit for bit (fp64
                          multiply, then add) for every k.  The closed-form kernels then form output times arithmetically and
                          take their store-friendly emission path; results are identical either way.  0 = not verified */
  const double *sse_ref; /* fused objective (PINTS SumOfSquaresError over Model.simulate, train-d0.py:415-439, :508-540), optional:
                          DEVICE [n_prot][n_out] reference currents.  With sse_out set, every trajectory's
                          sum_k (i_k - sse_ref[protocol][k])^2 (i as in the i_out epilogue) is accumulated in the kernel and
                          written to sse_out[b] (inf where the solve failed -- the reference's time-limit rule); y_out and i_out
                          may then be NULL, so that no trace leaves the chip: BASELINE configs[3]'s 65 536 candidates x 32
                          sweeps x 1e5 samples would be 4.5 TB of traces.  Needs the output-grid hint. */
  double *sse_out;     /* DEVICE [n_traj] fp64 or NULL */
  double max_step;     /* EXTENSION (torchdiffeq 0.2.1's dopri5 has no such option; 0 = off = reference behaviour): cap on the
                          step size in ms.  At an equilibrium dopri5 grows dt until h*lambda leaves its stability region
                          (the error estimate of a state AT equilibrium is ~0); the forward solve copes through rejections,
                          but the reverse-mode derivative of such accepted-but-unstable steps multiplies adjoints by
                          |R(h*lambda)| >> 1 per step.  max_step < 3.3 / lambda_max keeps the backward sweep bounded. */
  const double *v_at_outputs; /* optional (NULL = off), closed-form models with i_out or sse_out: DEVICE [n_prot][n_out] protocol
                          voltage at the output times, filled by ionode_protocol_at_outputs() for the same protocols and t_eval.
                          The current / objective epilogue then loads V(t_k) instead of re-deriving it per trajectory per sample
                          (same values: the pre-pass runs the integrator's own lookup).  Pays when n_prot << n_traj. */
  int64_t mlp_image_stride; /* NN models, several weight sets in one launch (an ensemble of trained nets, a population of random
                          initialisations): floats between consecutive packed images in `mlp_packed` (>= ionode_mlp_packed_floats) ... */
  int32_t traj_per_image;   /* ... and the number of CONSECUTIVE trajectories that share one image: trajectory b uses image
                          b / traj_per_image.  A multiple of 16 (of 64 with tile_waves = 64).  0 = one image for every trajectory.
                          Forward path only (the backward sweep differentiates one weight set). */
  const int32_t *launch_order; /* optional (NULL = index order), DEVICE [n_traj]: a permutation of 0..n_traj-1.  Launch slot s
                          integrates trajectory launch_order[s]; every input and output stays at the trajectory's OWN index, so the
                          results are those of index order, bit for bit -- only which trajectories share a tile / a wavefront, and
                          in which sequence the tiles start, changes.  What the reference's callers would sort by: predicted cost
                          (tiles become homogeneous, the launch a longest-first list schedule) or protocol (the lanes of a
                          wavefront interpolate one protocol).  Not with traj_per_image.  Since ABI 6. */
} ionode_desc;

#define IONODE_DEFAULT_MAX_TOTAL_STEPS 1000000

/* Number of floats of the device-side weight image for an (L, N) MLP  Linear(2,N) + L x Linear(N,N) + Linear(N,1). */
size_t ionode_mlp_packed_floats(int32_t mlp_layers, int32_t mlp_width);

/* HOST -> HOST.  Re-lays a state dict (flat fp32, order net.0.weight [N][2], net.0.bias [N],
 * {net.2i.weight [N][N], net.2i.bias [N]} x L, net.last.weight [1][N], net.last.bias [1]; row-major as
 * torch stores nn.Linear) into the MFMA-fragment order the kernels stream.  The caller uploads `packed`.
 * The image is an opaque, derived artefact of THIS library build (its sections and their order follow the kernels:
 * fragment streams, the 4-trajectory tile's section, the per-lane net's scalar row pairs): pack after loading the
 * library, do not persist images across library versions. */
int ionode_mlp_pack(const float *state_dict_flat, int32_t mlp_layers, int32_t mlp_width, float *packed);

/*
 * Integrate d->n_traj independent trajectories on the current device, asynchronously on `stream`.
 *
 *   mlp_packed    device, ionode_mlp_packed_floats() floats (NULL for HH2 / MARKOV6)
 *   params        device, [B][n_params] fp64: p1..p8 (NNF reads p5..p8) or p1..p12
 *   prot_v        device, [P][prot_n] fp64 mV
 *   prot_t        device, [prot_n] fp64 ms shared by all protocols, or NULL for the uniform grid
 *   prot_of_traj  device, [B] int32 protocol index per trajectory, or NULL (trajectory b uses b % P)
 *   y0            device, [B][D] in the state dtype
 *   t_eval        device, [n_out] fp64, strictly increasing; t_eval[0] is the initial time
 *   y_out         device, [B][n_out][D] in the state dtype (may be NULL when d->sse_out is set)
 *   i_out         device, [B][n_out] fp64 current trace, or NULL
 *   status        device, [B] int32 IONODE_STATUS_*
 *   stats         device, [B][4] int64 {accepted steps, rejected steps, RHS evaluations, status}, or NULL
 *   stream        hipStream_t (NULL = default stream)
 *
 * Returns IONODE_OK once the kernel is enqueued; per-trajectory failures are reported in status[]
 * (their remaining outputs are NaN), never as a return code.
 */
int ionode_dopri5(const ionode_desc *d, const float *mlp_packed, const double *params, const double *prot_v,
                  const double *prot_t, const int32_t *prot_of_traj, const void *y0, const double *t_eval,
                  void *y_out, double *i_out, int32_t *status, int64_t *stats, void *stream);

/* Pre-pass for ionode_desc.v_at_outputs: v_out[p][k] = V_p(t_eval[k]) for the d->n_prot protocols at the d->n_out output
 * times, by the integrator's own lookup (linear interpolation, -80 mV / v_oob outside the protocol: train-s1.py:218-237).
 * Reads d->n_out, n_prot, prot_n, prot_t0, prot_dt, v_oob.  Asynchronous on `stream`. */
int ionode_protocol_at_outputs(const ionode_desc *d, const double *prot_v, const double *prot_t, const double *t_eval,
                               double *v_out, void *stream);

/* Launch geometry the dispatcher would use for `d` (for tests / bench reporting): grid, block, LDS bytes, tile_waves. */
int ionode_launch_geometry(const ionode_desc *d, int32_t out[4]);

/* Name of the kernel instantiation ionode_dopri5 would launch for `d` (matches rocprofv3 kernel-trace).  The descriptor does
 * not say whether an i_out buffer will be passed; a fused objective or a v_at_outputs table implies the epilogue variant. */
const char *ionode_kernel_name(const ionode_desc *d);

/* Name of the instantiation this thread's last successful ionode_dopri5 call launched ("" before the first). */
const char *ionode_last_kernel_name(void);

/* Batch size from which ionode_dopri5 (tile_waves = 0) takes the one-trajectory-per-lane kernel (64 trajectories per wavefront) of
 * `model` (mlp_width: the net's width for the NN models, ignored otherwise); 0 if the model has no such kernel.  Callers that sort
 * trajectories by protocol for those kernels (launch_order) read the crossover here instead of keeping a copy.  Since ABI 8. */
int32_t ionode_lane_wise_from(int32_t model, int32_t mlp_width);

const char *ionode_last_error(void);
int32_t ionode_abi_version(void);

/* ---------------------------------------------------------------------------------------------------------------------
 * Gradients through the solve (BASELINE.json configs[4]).  Reference interface: `from torchdiffeq import odeint_adjoint as
 * odeint` (train-s1.py:29-32) -- the reference only switches this import and never differentiates (SURVEY.md finding 3),
 * so what is computed is defined here: the exact reverse-mode derivative of the discretisation the forward launch executed,
 * with its accepted steps (t0, dt) as constants; rejected attempts do not contribute.  NN-f / NN-d for the widths of
 * architectures s00-s11 (N = 10, 100, 200, 500) with at most 15 hidden layers, and the closed-form HH 2-state and 6-state
 * models (no grad_image, no records: gradients with respect to the rate parameters and y0 only; D = 6 and 12 parameters
 * for the 6-state model).
 *
 * Flow (all pointers DEVICE unless noted):
 *   1. forward:  ionode_dopri5() with d->ckpt / d->ckpt_cap set; n_accepted[b] = stats[b][0] for status[b] == 0, else 0
 *   2. sweep:    ionode_dopri5_backward() over iterations [0, n_iter), n_iter = max(n_accepted) + 1, in one launch or in
 *                chunks [it_begin, it_end) (the adjoint state is carried in `state`); each tile-evaluation appends one record
 *                of ionode_grad_record_floats() floats to `records` ([tiles][it_end - it_begin][6] records per chunk)
 *   3. reduce:   ionode_grad_reduce() contracts a chunk's records into n_slabs partial weight gradients; the caller sums
 *                the partials (and the chunks) and un-pads them (layout: ionode_grad_partial_floats)
 * ------------------------------------------------------------------------------------------------------------------- */

/* HOST -> HOST: flat state dict (ionode_mlp_pack's input order) -> grad image (forward AND transposed MFMA fragments). */
size_t ionode_grad_image_floats(int32_t mlp_layers, int32_t mlp_width);
int ionode_grad_pack(const float *state_dict_flat, int32_t mlp_layers, int32_t mlp_width, float *image);

/* floats of one record (one MLP vector-Jacobian product of one 16-trajectory tile): h_0..h_L, d_0..d_L tiles + 64 scalars */
size_t ionode_grad_record_floats(int32_t mlp_layers, int32_t mlp_width);

/*
 * Backward sweep, asynchronous on `stream`.  d: the forward launch's descriptor (model, state dtype, sizes, protocol grid,
 * v_oob, ckpt, ckpt_cap).
 *   (D = n_state, NPAR = 8, or 12 for the 6-state model)
 *   grad_image   device, ionode_grad_image_floats() floats (NULL for the closed-form models)
 *   n_accepted   device, [B] int32: accepted steps to replay per trajectory (0 = contributes nothing)
 *   grad_y       device, [B][n_out][D] dL/dy_out in the state dtype
 *   state        device, [B][2 D + NPAR] fp64 scratch carried between chunk launches (need not be initialised for it_begin == 0)
 *   records      device, [ceil(B/16)][it_end - it_begin][6][record floats] fp32, or NULL to skip the weight-gradient stream
 *   grad_params  device, [B][NPAR] fp64 dL/dp      (written by the launch with it_end == n_iter)
 *   grad_y0      device, [B][D] fp64 dL/dy0
 */
int ionode_dopri5_backward(const ionode_desc *d, int32_t it_begin, int32_t it_end, int32_t n_iter, const float *grad_image,
                           const double *params, const double *prot_v, const double *prot_t, const int32_t *prot_of_traj,
                           const double *t_eval, const int32_t *n_accepted, const void *grad_y, double *state,
                           float *records, double *grad_params, double *grad_y0, void *stream);

/*
 * Two-phase form of the same sweep (NN-f / NN-d).  A stage's vector-Jacobian product is LINEAR in its seed, and the seed is a scalar
 * per trajectory: product(seed) = seed * product(1).  Everything else the product needs -- the stage inputs, protocol voltages, rate
 * exponentials -- and the reduction of the step's output gradients depend on the step's checkpoint only.  So
 *   ionode_dopri5_backward_recompute()  runs, for EVERY (tile, step) of the chunk at once (the whole chip instead of one workgroup per
 *       16-trajectory tile), the unit-seed product of all six stages: records with unit-seed D tiles, and per tile and step
 *       ionode_grad_packet_doubles() doubles of scalars (`packets`: [ceil(B/16)][it_end - it_begin][doubles]);
 *   ionode_dopri5_backward_sweep()      walks the steps with the adjoint algebra alone (one wavefront per tile, no MLP work) and
 *       writes each evaluation's seed into its record;
 *   ionode_grad_reduce_unit()           contracts such records, scaling the D tiles by the seeds while staging them.
 * Same gradients as ionode_dopri5_backward() up to fp32 rounding (the seed multiplies at the end of the product instead of at its
 * start).  Call order per chunk: recompute, sweep, reduce_unit; recompute of chunk k + 1 may run beside the sweep of chunk k.
 */
size_t ionode_grad_packet_doubles(void);
int ionode_dopri5_backward_recompute(const ionode_desc *d, int32_t it_begin, int32_t it_end, int32_t n_iter, const float *grad_image,
                                     const double *params, const double *prot_v, const double *prot_t, const int32_t *prot_of_traj,
                                     const double *t_eval, const int32_t *n_accepted, const void *grad_y, float *records,
                                     double *packets, void *stream);
int ionode_dopri5_backward_sweep(const ionode_desc *d, int32_t it_begin, int32_t it_end, int32_t n_iter, const float *grad_image,
                                 const double *params, const double *prot_v, const double *prot_t, const int32_t *prot_of_traj,
                                 const double *t_eval, const int32_t *n_accepted, const void *grad_y, double *state,
                                 float *records, const double *packets, double *grad_params, double *grad_y0,
                                 void *stream);
int ionode_grad_reduce_unit(int32_t mlp_layers, int32_t mlp_width, const float *records, int64_t n_records, int32_t n_slabs,
                            float *partials, void *stream);

/* floats of one slab's partial gradient: [NP][4]{db0, dW0[.][0], dW0[.][1], 0} | L x (dW_l [NP][NP] + db_l [NP]) | dwl [NP] +
 * {dbl, 0, 0, 0}, NP = 16 * ceil(N / 16), rows/columns >= N are padding */
size_t ionode_grad_partial_floats(int32_t mlp_layers, int32_t mlp_width);

/* The slab count that makes one round of workgroups of ionode_grad_reduce() on the current device (256 compute units when there is
 * none): a slab costs mlp_layers x (column blocks of the width) heavy workgroups + one light one, N = 200 runs two workgroups per
 * compute unit.  Any n_slabs >= 1 is valid; this one is the fastest.  Since ABI 9. */
int32_t ionode_grad_reduce_slabs(int32_t mlp_layers, int32_t mlp_width, int64_t n_records);

/* partials[n_slabs][ionode_grad_partial_floats()] = per-slab sums over records [n_records * s / n_slabs, ...); asynchronous */
int ionode_grad_reduce(int32_t mlp_layers, int32_t mlp_width, const float *records, int64_t n_records, int32_t n_slabs,
                       float *partials, void *stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * MLP state-space regression step -- the reference's actual training loop, train-s1.py:891-909 /
 * train-d2.py:901-915:  p = net(x_av.float()) / netscale [+ model_dadt];  loss = MSELoss(reduction='sum')(p, y.float());
 * loss.backward(); Adam(lr 1e-3).step(); StepLR.step().  One iteration = ionode_regress_step -> ionode_grad_reduce ->
 * ionode_adam_step -> ionode_image_refresh, all asynchronous on one stream, no host synchronisation.
 * ------------------------------------------------------------------------------------------------------------------- */

/* Forward + backward of the net for every 16-row tile of x (fp32 MFMA, activations LDS-resident): one record per tile into
 * `records` ([ceil(n_rows/16)][ionode_grad_record_floats()]) and sum((p - y)^2) partials into loss_partials[n_workgroups]
 * (fp64).  x [n_rows][2], y [n_rows], offset [n_rows] or NULL: device fp32.  n_workgroups = persistent grid size (<= tiles). */
int ionode_regress_step(int32_t mlp_layers, int32_t mlp_width, const float *grad_image, const float *x, const float *offset,
                        const float *y, int32_t n_rows, float netscale, float *records, double *loss_partials,
                        int32_t n_workgroups, void *stream);

/* g[i] = sum over slabs of partials[s][padmap[i]] (flat state-dict order), then torch.optim.Adam's update (no amsgrad, no
 * weight decay), element-wise in fp32.  `step` counts from 1.  grad_out (optional) receives g; apply = 0 only gathers g. */
int ionode_adam_step(int32_t n_params, int32_t n_slabs, int32_t mlp_layers, int32_t mlp_width, const float *partials,
                     const int32_t *padmap, float *weights, float *exp_avg, float *exp_avg_sq, float lr, float beta1,
                     float beta2, float eps, int32_t step, float *grad_out, int32_t apply, void *stream);

/* grad_image[k] = weights[image_map[k] - 1] (image_map[k] == 0: padding).  image_map = ionode_grad_pack() of the values
 * 1, 2, ..., n_params (exact in fp32), converted to int32 by the caller. */
int ionode_image_refresh(int32_t mlp_layers, int32_t mlp_width, const int32_t *image_map, const float *weights,
                         float *grad_image, void *stream);

const char *ionode_grad_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* IONODE_H */
