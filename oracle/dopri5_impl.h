/*
 * oracle/dopri5_impl.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Type-generic body of the CPU oracle.  Included twice by dopri5_oracle.c with
 *   REAL = float   SFX(x) = x##_f32   (reference-compatible: solver state in y0.dtype = fp32)
 *   REAL = double  SFX(x) = x##_f64   (fp64 state: BASELINE.json configs 2-3)
 *
 * It restates, operation by operation, what the reference executes on this path:
 *   - the right-hand sides defined in the reference scripts
 *       HH 2-state   Lambda.forward            train-s1.py:161-177
 *       6-state      Lambda.forward            train-d1.py:165-187 (= train-d2.py:165-187)
 *       NN-f         ODEFunc.forward           train-s1.py:231-247
 *       NN-d         ODEFunc.forward           train-d2.py:247-272
 *       protocol     interp1d + -80 mV rule    train-s1.py:218-229, :234-237
 *   - the dopri5 integrator those scripts call, torchdiffeq==0.2.1 (requirements.txt:1).
 *     That package is NOT in /root/reference and cannot be obtained offline, so its
 *     published algorithm is restated from SURVEY.md Appendix A: Dormand-Prince/Shampine
 *     tableau, initial-step heuristic, RMS error norm, accept iff ratio <= 1,
 *     safety 0.9 / ifactor 10 / dfactor 0.2, Perturb.PREV at the alpha = 1 stages,
 *     4th-order dense output, no clipping of steps to output times.
 *
 * Arithmetic rules (the "canonical arithmetic" the HIP kernels are held to):
 *   - every torch elementwise op is one IEEE operation in the dtype torch would use;
 *     no contraction (the file is compiled with -ffp-contract=off); sums run in index order;
 *   - a 0-dim fp64 tensor combined with a dimensioned fp32 tensor is first rounded to fp32
 *     (torch type promotion), which is why rtol/atol are REAL below;
 *   - the MLP is fp32 always (`.float()`, train-s1.py:245).  Its accumulation order is not
 *     observable in the reference (it is whatever the CPU BLAS did); the oracle fixes it as
 *     an fmaf chain seeded with the bias, over k in the order
 *         for s in 0..NT-1, tile = (s + rowtile % 4) mod NT: for r in 0..3: for q in 0..3: k = 16*tile + 4*q + r
 *     with the width zero-padded to NP = 16*NT, NT = ceil(N/16) (no rotation when NT = 1).  Rows of the last
 *     NT mod 4 row tiles are instead four chains over the k-tiles tile % 4 == w (chain 0 seeded with the bias),
 *     folded (p0 + p1) + (p2 + p3); the final Linear(N, 1) is four chains (one per q, seeded with 0)
 *     combined as ((p0 + p1) + (p2 + p3)) + bias   (DESIGN.md "canonical MLP order").
 */

#ifndef REAL
#error "include from dopri5_oracle.c"
#endif

/* ---- small helpers --------------------------------------------------- */

static inline REAL SFX(r_abs)(REAL x) { return x < 0 ? -x : x; }
static inline REAL SFX(r_max)(REAL a, REAL b) { return a > b ? a : b; } /* torch.max(a,b), finite inputs */

/* tensor.pow(2).mean().sqrt() in REAL, index order (torchdiffeq _rms_norm) */
static inline REAL SFX(rms)(const REAL *x, int n) {
  REAL s = x[0] * x[0];
  for (int i = 1; i < n; ++i) s = s + x[i] * x[i];
  s = s / (REAL)n;
  return R_SQRT(s);
}

/* ---- right-hand sides -------------------------------------------------- */

/* func(t, y) as _PerturbFunc hands it over: t already cast to y.dtype. */
static void SFX(rhs)(const ctx_t *c, REAL t, const REAL *y, REAL *f) {
  double v;
  const int inrange = protocol_v(c, (double)t, &v);
  const double *p = c->p;

#if IS_F32
  if (!inrange) {
    /* train-s1.py:236-237: v = torch.tensor([-80]) is int64, so `p * v` is a float32 tensor
     * and exp runs in fp32; every product below is then fp32 x fp32. */
    const float vf = (float)c->v_oob;
    if (c->model == MODEL_MARKOV6) {
      const float a1 = (float)p[0] * det_expf((float)p[1] * vf);
      const float b1 = (float)p[2] * det_expf((float)(-p[3]) * vf);
      const float bh = (float)p[4] * det_expf((float)p[5] * vf);
      const float ah = (float)p[6] * det_expf((float)(-p[7]) * vf);
      const float a2 = (float)p[8] * det_expf((float)p[9] * vf);
      const float b2 = (float)p[10] * det_expf((float)(-p[11]) * vf);
      const float c1 = y[0], c2 = y[1], i_ = y[2], ic1 = y[3], ic2 = y[4], o = y[5];
      f[0] = a1 * c2 + ah * ic1 + b2 * o - (b1 + bh + a2) * c1;
      f[1] = b1 * c1 + ah * ic2 - (a1 + bh) * c2;
      f[2] = a2 * ic1 + bh * o - (b2 + ah) * i_;
      f[3] = a1 * ic2 + bh * c1 + b2 * i_ - (b1 + ah + a2) * ic1;
      f[4] = b1 * ic1 + bh * c2 - (ah + a1) * ic2;
      f[5] = a2 * c1 + ah * i_ - (b2 + bh) * o;
      return;
    }
    const float a = y[0], r = y[1];
    const float k3 = (float)p[4] * det_expf((float)p[5] * vf);
    const float k4 = (float)p[6] * det_expf((float)(-p[7]) * vf);
    const float drdt = -k3 * r + k4 * (1.0f - r);
    float dadt = 0.0f;
    if (c->model == MODEL_HH2 || c->model == MODEL_NND) {
      const float k1 = (float)p[0] * det_expf((float)p[1] * vf);
      const float k2 = (float)p[2] * det_expf((float)(-p[3]) * vf);
      dadt = k1 * (1.0f - a) - k2 * a;
    }
    if (c->model == MODEL_NNF || c->model == MODEL_NND) {
      const float nv = vf / 100.0f; /* int64 tensor / fp32 vrange -> fp32 */
      const float net = mlp_eval(c, nv, a) / 1000.0f;
      dadt = (c->model == MODEL_NND) ? dadt + net : net;
    }
    f[0] = dadt;
    f[1] = drdt;
    return;
  }
#else
  (void)inrange; /* fp64 state: out-of-range voltage is v_oob through the same fp64 formulas */
#endif

  if (c->model == MODEL_MARKOV6) {
    /* train-d1.py:173-185; rates are fp64 (1,) tensors, states 0-dim y.dtype tensors */
    const double a1 = p[0] * det_exp(p[1] * v);
    const double b1 = p[2] * det_exp(-p[3] * v);
    const double bh = p[4] * det_exp(p[5] * v);
    const double ah = p[6] * det_exp(-p[7] * v);
    const double a2 = p[8] * det_exp(p[9] * v);
    const double b2 = p[10] * det_exp(-p[11] * v);
    const double c1 = y[0], c2 = y[1], i_ = y[2], ic1 = y[3], ic2 = y[4], o = y[5];
    f[0] = (REAL)(a1 * c2 + ah * ic1 + b2 * o - (b1 + bh + a2) * c1);
    f[1] = (REAL)(b1 * c1 + ah * ic2 - (a1 + bh) * c2);
    f[2] = (REAL)(a2 * ic1 + bh * o - (b2 + ah) * i_);
    f[3] = (REAL)(a1 * ic2 + bh * c1 + b2 * i_ - (b1 + ah + a2) * ic1);
    f[4] = (REAL)(b1 * ic1 + bh * c2 - (ah + a1) * ic2);
    f[5] = (REAL)(a2 * c1 + ah * i_ - (b2 + bh) * o);
    return;
  }

  const REAL a = y[0], r = y[1];
  /* `1. - a`, `self.unity - r`: computed in y.dtype before meeting the fp64 rate */
  const REAL one_m_a = (REAL)1 - a;
  const REAL one_m_r = (REAL)1 - r;
  const double k3 = p[4] * det_exp(p[5] * v);
  const double k4 = p[6] * det_exp(-p[7] * v);
  const double drdt = -k3 * (double)r + k4 * (double)one_m_r;
  double dadt = 0.0;
  if (c->model == MODEL_HH2 || c->model == MODEL_NND) {
    const double k1 = p[0] * det_exp(p[1] * v);
    const double k2 = p[2] * det_exp(-p[3] * v);
    dadt = k1 * (double)one_m_a - k2 * (double)a;
  }
  if (c->model == MODEL_NNF || c->model == MODEL_NND) {
    const double nv = v / 100.0;                              /* v / self.vrange */
    const float net = mlp_eval(c, (float)nv, (float)a) / 1000.0f; /* .float(); / self.netscale */
    dadt = (c->model == MODEL_NND) ? dadt + (double)net : (double)net;
  }
  f[0] = (REAL)dadt; /* rounded into k (y0.dtype) */
  f[1] = (REAL)drdt;
}

/* ---- dopri5 ------------------------------------------------------------ */

typedef struct {
  REAL y1[MAXD], f1[MAXD];
  double t0, t1, dt;
  REAL ic[5][MAXD]; /* e, d, c, b, a */
} SFX(rkstate);

static double SFX(select_initial_step)(const ctx_t *c, double t0d, const REAL *y0, const REAL *f0,
                                       REAL rtol, REAL atol, int64_t *nfe) {
  const int D = c->D;
  const REAL t0 = (REAL)t0d;
  REAL scale[MAXD], tmp[MAXD], y1[MAXD], f1[MAXD];
  for (int d = 0; d < D; ++d) scale[d] = atol + SFX(r_abs)(y0[d]) * rtol;
  for (int d = 0; d < D; ++d) tmp[d] = y0[d] / scale[d];
  const REAL d0 = SFX(rms)(tmp, D);
  for (int d = 0; d < D; ++d) tmp[d] = f0[d] / scale[d];
  const REAL d1 = SFX(rms)(tmp, D);
  REAL h0;
  if (d0 < (REAL)1e-5 || d1 < (REAL)1e-5) h0 = (REAL)1e-6;
  else h0 = (REAL)0.01 * d0 / d1;
  for (int d = 0; d < D; ++d) y1[d] = y0[d] + h0 * f0[d];
  SFX(rhs)(c, t0 + h0, y1, f1);
  ++*nfe;
  for (int d = 0; d < D; ++d) tmp[d] = (f1[d] - f0[d]) / scale[d];
  const REAL d2 = SFX(rms)(tmp, D) / h0;
  REAL h1;
  if (d1 <= (REAL)1e-15 && d2 <= (REAL)1e-15) h1 = SFX(r_max)((REAL)1e-6, h0 * (REAL)1e-3);
  else h1 = (REAL)det_root5((double)((REAL)0.01 / (d1 > d2 ? d1 : d2))); /* ** (1 / (order + 1)) */
  const REAL h = ((REAL)100 * h0 < h1) ? (REAL)100 * h0 : h1;
  return (double)h;
}

/* k (D x 7) . coeffs(n) in REAL, index order, mul then add */
static inline REAL SFX(kdot)(const REAL k[MAXD][7], int d, const REAL *bd, int n) {
  REAL s = k[d][0] * bd[0];
  for (int j = 1; j < n; ++j) s = s + k[d][j] * bd[j];
  return s;
}

/* one _adaptive_step; returns 0 ok, else status code */
static int SFX(adaptive_step)(const ctx_t *c, SFX(rkstate) *s, REAL rtol, REAL atol, int64_t *nfe,
                              int *accepted, double *ratio_out) {
  const int D = c->D;
  const double t0 = s->t1, dt = s->dt;
  const double t1 = t0 + dt;
  if (!(t0 + dt > t0)) return STATUS_DT_UNDERFLOW;
  for (int d = 0; d < D; ++d)
    if (!isfinite((double)s->y1[d])) return STATUS_NONFINITE;

  const REAL *y0 = s->y1, *f0 = s->f1;
  const REAL t0s = (REAL)t0, dts = (REAL)dt, t1s = (REAL)t1;
  REAL k[MAXD][7], yi[MAXD], fi[MAXD], bd[7];
  for (int d = 0; d < D; ++d) k[d][0] = f0[d];
  for (int i = 0; i < 6; ++i) {
    REAL ti;
    if (TAB_ALPHA[i] == 1.0) ti = R_NEXTAFTER(t1s, t1s - (REAL)1); /* Perturb.PREV */
    else ti = t0s + (REAL)TAB_ALPHA[i] * dts;
    for (int j = 0; j <= i; ++j) bd[j] = (REAL)TAB_BETA[i][j] * dts;
    for (int d = 0; d < D; ++d) yi[d] = y0[d] + SFX(kdot)(k, d, bd, i + 1);
    SFX(rhs)(c, ti, yi, fi);
    ++*nfe;
    for (int d = 0; d < D; ++d) k[d][i + 1] = fi[d];
  }
  /* c_sol == beta[5] + [0]  =>  y1 = y_5 (FSAL) */
  REAL y1[MAXD], err[MAXD], tmp[MAXD];
  for (int j = 0; j < 7; ++j) bd[j] = dts * (REAL)TAB_CERR[j];
  for (int d = 0; d < D; ++d) {
    y1[d] = yi[d];
    err[d] = SFX(kdot)(k, d, bd, 7);
  }
  for (int d = 0; d < D; ++d) {
    const REAL tol = atol + rtol * SFX(r_max)(SFX(r_abs)(y0[d]), SFX(r_abs)(y1[d]));
    tmp[d] = err[d] / tol;
  }
  const REAL ratio = SFX(r_abs)(SFX(rms)(tmp, D));
  const int acc = ratio <= (REAL)1;
  *accepted = acc;
  *ratio_out = (double)ratio;

  /* _optimal_step_size, fp64 */
  double dt_next;
  if (ratio == (REAL)0) dt_next = dt * 10.0;
  else {
    const double dfactor = (ratio < (REAL)1) ? 1.0 : 0.2;
    const double er = (double)ratio;
    double fac = 0.9 / det_root5(er);
    if (!(fac > dfactor)) fac = dfactor; /* torch.max(x, dfactor) */
    if (!(fac < 10.0)) fac = 10.0;       /* torch.min(ifactor, .) */
    if (isnan(er)) fac = NAN;
    dt_next = dt * fac;
  }

  if (acc) {
    /* _interp_fit */
    REAL ymid[MAXD];
    for (int j = 0; j < 7; ++j) bd[j] = dts * (REAL)TAB_CMID[j];
    for (int d = 0; d < D; ++d) ymid[d] = y0[d] + SFX(kdot)(k, d, bd, 7);
    for (int d = 0; d < D; ++d) {
      const REAL F0 = k[d][0], F1 = k[d][6], Y0 = y0[d], Y1 = y1[d], YM = ymid[d];
      const REAL A = ((REAL)2 * dts) * (F1 - F0) - (REAL)8 * (Y1 + Y0) + (REAL)16 * YM;
      const REAL B = dts * ((REAL)5 * F0 - (REAL)3 * F1) + (REAL)18 * Y0 + (REAL)14 * Y1 - (REAL)32 * YM;
      const REAL C = dts * (F1 - (REAL)4 * F0) - (REAL)11 * Y0 - (REAL)5 * Y1 + (REAL)16 * YM;
      const REAL Dc = dts * F0;
      s->ic[0][d] = Y0; s->ic[1][d] = Dc; s->ic[2][d] = C; s->ic[3][d] = B; s->ic[4][d] = A;
    }
    for (int d = 0; d < D; ++d) { s->y1[d] = y1[d]; s->f1[d] = k[d][6]; }
    s->t0 = t0;
    s->t1 = t1;
  } else {
    s->t0 = t0;
    s->t1 = t0;
  }
  s->dt = dt_next;
  return STATUS_OK;
}

static int SFX(solve)(const ctx_t *c, const double *y0d, const double *t_eval, int n_out, double rtol_d,
                      double atol_d, int64_t max_steps, int64_t max_total, double max_step, double *y_out, int64_t *stats, double *step_log,
                      int64_t step_log_cap) {
  const int D = c->D;
  const REAL rtol = (REAL)rtol_d, atol = (REAL)atol_d;
  int64_t nfe = 0, nacc = 0, nrej = 0;
  SFX(rkstate) s;
  REAL y0[MAXD];
  for (int d = 0; d < D; ++d) y0[d] = (REAL)y0d[d];
  SFX(rhs)(c, (REAL)t_eval[0], y0, s.f1);
  ++nfe;
  for (int d = 0; d < D; ++d) s.y1[d] = y0[d];
  s.dt = SFX(select_initial_step)(c, t_eval[0], y0, s.f1, rtol, atol, &nfe);
  const double dt_max = max_step > 0.0 ? max_step : INFINITY; /* product extension; off in every reference-parity test */
  if (s.dt > dt_max) s.dt = dt_max;
  s.t0 = s.t1 = t_eval[0];
  for (int q = 0; q < 5; ++q)
    for (int d = 0; d < D; ++d) s.ic[q][d] = y0[d];
  for (int d = 0; d < D; ++d) y_out[d] = (double)y0[d];

  int status = STATUS_OK;
  int i = 1;
  for (; i < n_out; ++i) {
    const double next_t = t_eval[i];
    int64_t n_steps = 0; /* torchdiffeq _advance(next_t): the max_num_steps counter restarts for every output time */
    while (next_t > s.t1) {
      if (n_steps >= max_steps || nacc + nrej >= max_total) { status = STATUS_MAX_STEPS; break; }
      ++n_steps;
      int acc = 0;
      const double t_before = s.t1, dt_before = s.dt;
      double ratio_d = 0.0;
      status = SFX(adaptive_step)(c, &s, rtol, atol, &nfe, &acc, &ratio_d);
      if (status != STATUS_OK) break;
      if (s.dt > dt_max) s.dt = dt_max;
      if (step_log && nacc + nrej < step_log_cap) {
        double *row = step_log + 4 * (nacc + nrej);
        row[0] = t_before; row[1] = dt_before; row[2] = ratio_d; row[3] = (double)acc;
      }
      if (acc) ++nacc; else ++nrej;
    }
    if (status != STATUS_OK) break;
    /* _interp_evaluate: x in fp64 then cast; running powers */
    const REAL x = (REAL)((next_t - s.t0) / (s.t1 - s.t0));
    for (int d = 0; d < D; ++d) {
      REAL total = s.ic[0][d] + x * s.ic[1][d];
      REAL xp = x;
      for (int q = 2; q < 5; ++q) {
        xp = xp * x;
        total = total + xp * s.ic[q][d];
      }
      y_out[(size_t)i * D + d] = (double)total;
    }
  }
  for (; i < n_out; ++i)
    for (int d = 0; d < D; ++d) y_out[(size_t)i * D + d] = NAN; /* failed trajectories: rest is NaN */
  if (stats) { stats[0] = nacc; stats[1] = nrej; stats[2] = nfe; stats[3] = status; }
  return status;
}
