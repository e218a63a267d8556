#!/usr/bin/env python3
"""Generator of the hand-scheduled hidden-layer stream of the 16-trajectory MLP tile (gfx950 inline asm).

    python3 tools/gen_mlp_asm.py --nt 13 --out neural-ode-ion-channels_amd/csrc/mlp_asm_nt13.inc

What it emits (consumed by MlpTileAsm in csrc/ionode_mlp_asm.hpp): two C string macros,

    IONODE_MLPASM_INIT_<NT>    prime the weight ring with hidden layer 0 (kernel start)
    IONODE_MLPASM_LAYERS_<NT>  the whole hidden stack of one stage evaluation: for l = 0 .. L-1 one pass of
                               NT k-tile steps of v_mfma_f32_16x16x4_f32 with the layer boundary software-pipelined

plus register-map constants.  The arithmetic is the canonical accumulation order of ionode_device.hpp (MlpTile::eval):
every accumulator's chain visits the same (k-tile, k-step) sequence, so the bits do not change; what changes is the order
in which INDEPENDENT accumulators are interleaved, where the epilogue sits, and who allocates the registers:

  * fixed register assignment: weight ring in AGPRs a[0 : 12*NT + 4*NOWN) (loaded by buffer_load straight into AGPRs,
    read by the MFMAs as srcA), accumulators / bias / B operands / temporaries in v[VB : VB+72); hipcc sees one opaque
    statement with clobbers and keeps its own values elsewhere -- no AGPR<->VGPR copies, no spills in the stream;
  * the LAST step of a layer finishes the wavefront's own tile (accumulator 0) and accumulator 1 first; their
    LeakyReLU + ds_write run in the shadow of the remaining MFMAs of that step;
  * the FIRST step of the next layer starts on accumulators 0 and 1 (B operand = own tile, from registers, C operand =
    bias prefetched two steps earlier) while accumulator 2 and the remainder tile's partial sums of the previous layer
    are post-processed and stored; the workgroup barrier sits in the middle of that step and the first LDS-read B
    operand is issued right behind it, under the step's second half;
  * remainder-tile fold, bias prefetch, address bookkeeping: all inside MFMA gaps (<= 4 VALU or 1 LDS + 2 VALU per gap);
  * s_waitcnt counts are derived by simulation of the in-order vmcnt / lgkmcnt queues (checked: the emitted iteration
    is a fixed point).

Hazards (probed with hipcc on gfx950, tools/README.md): MFMA result -> VALU / LDS-store read: 10 wait states (here: at least
two younger MFMAs issued in between = 64 cycles, or explicit s_nop); VALU write -> MFMA operand: 2 wait states;
dependent MFMA on the same accumulator: interlocked.
"""
import argparse

G = 4  # wavefronts per tile


class Gen:
    def __init__(self, nt, pd=7, vb=180, sb=84, stamps=False):
        assert nt % G == 1 or True
        self.NT = nt
        self.F = nt // G
        self.R = nt - G * self.F
        assert self.R == 1 and self.F == 3, "written for N = 200 (NT = 13): three full row tiles + one K-split remainder tile per wavefront"
        self.NOWN = (nt + G - 1) // G
        self.FRAGS = nt * self.F + self.NOWN * self.R
        self.PD = pd            # ring depth in k-tile steps (slot of step u: u mod PD); NT <= 2 PD
        assert pd <= nt <= 2 * pd
        self.RD = self.NOWN if pd == nt else 2   # ring depth of the remainder fragments (owned steps 0, G, 2G, ...)
        assert self.NOWN % self.RD == 0
        self.ring_regs = 12 * pd + 4 * self.RD
        self.VB = vb
        self.SB = sb
        v = lambda k: vb + k
        # --- fixed VGPR map ---
        self.ACC = [v(0), v(4), v(8)]
        self.ACCR = v(12)
        self.T = [v(16), v(20), v(24)]
        self.TR = v(28)
        self.B = [v(32), v(36)]
        self.HO = v(40)
        self.X = v(44)
        self.Y = v(48)
        names = ["HW_IN", "HW_OUT", "FW_IN", "FW_OUT", "PL_IN", "PL_OUT", "PW_IN", "PW_OUT",
                 "BIAS_A", "BIAS_R", "VOFF", "DUMMY", "DUP", "LO_TILE", "LO_PART", "FOLDW", "W0A", "W0R", "WLA", "PART"]
        self.A = {n: v(52 + i) for i, n in enumerate(names)}
        self.n_vgpr = 76
        assert 52 + len(names) <= self.n_vgpr and vb + self.n_vgpr <= 256
        # --- fixed SGPR map ---
        s = lambda k: sb + k
        self.S_L = s(0)       # layer counter
        self.S_NL = s(1)      # number of layers
        self.S_LNEXT = s(2)   # byte offset of the layer the refills stream from
        self.S_T = s(3)       # scratch offset of one load
        self.S_LB = s(4)      # bytes per layer
        self.S_H0 = s(5)      # byte offset of hidden layer 0
        self.S_W0 = (s(6), s(7))    # all-ones iff wave == 0
        self.S_W3 = (s(8), s(9))    # all-ones iff wave < 3
        self.S_C01 = s(10)    # 0.01f
        self.S_LCUR = s(11)   # byte offset of the current layer
        self.n_sgpr = 12
        self.stamps = stamps   # diagnostic build: s_memtime deltas summed per position in the lanes of AGPR a[ring_regs]
        if stamps:
            self.S_NOW, self.S_LAST, self.S_DT = (s(12), s(13)), s(14), s(15)
            self.n_sgpr = 16
        self.lines = []
        self.reset_counters()

    # ---------------- in-order queue simulation ----------------
    def reset_counters(self):
        self.ds_seq = 0
        self.ds_done = 0
        self.vm_seq = 0
        self.vm_done = 0
        self.ring_id = {}   # fragment slot -> vm id of its latest load
        self.tag_id = {}    # named LDS read -> ds id

    def emit(self, s):
        self.lines.append(s)

    def ds_issue(self, tag=None):
        self.ds_seq += 1
        if tag:
            self.tag_id[tag] = self.ds_seq
        return self.ds_seq

    def wait_ds(self, ident):
        if ident <= self.ds_done:
            return
        n = self.ds_seq - ident
        assert n <= 15, "lgkmcnt field"
        self.emit("s_waitcnt lgkmcnt(%d)" % n)
        self.ds_done = ident

    def wait_ds_all(self):
        if self.ds_done < self.ds_seq:
            self.emit("s_waitcnt lgkmcnt(0)")
            self.ds_done = self.ds_seq

    def wait_vm(self, ident):
        if ident <= self.vm_done:
            return
        n = self.vm_seq - ident
        assert n <= 63, "vmcnt field"
        self.emit("s_waitcnt vmcnt(%d)" % n)
        self.vm_done = ident

    # ---------------- register helpers ----------------
    @staticmethod
    def vr(b, n=4):
        return "v[%d:%d]" % (b, b + n - 1)

    def slot_base(self, u, k):
        """first AGPR of fragment k (0..2 full, 3 remainder) of step u"""
        if k < 3:
            return 12 * (u % self.PD) + 4 * k
        return 12 * self.PD + 4 * ((u // G) % self.RD)

    def areg(self, u, e):
        if e < 12:
            return self.slot_base(u, e // 4) + e % 4
        return self.slot_base(u, 3) + (e - 12)

    def refill_target(self, u, k):
        """which fragment goes into the slot that step u's fragment k leaves: ('cur' | 'next' layer, step)"""
        NT, PD = self.NT, self.PD
        if k < 3:
            if u + PD < NT:
                return "cur", u + PD
            if u >= PD:
                return "next", u - PD
            return "next", u            # slot used once per pass
        o, no, rd = u // G, self.NOWN, self.RD
        if o + rd < no:
            return "cur", (o + rd) * G
        return "next", (o + rd - no) * G

    def frag_index(self, u, k):
        """index of fragment k (0..2 full, 3 remainder) of step u in a wavefront's layer stream (ionode_mlp_pack order)"""
        base = u * self.F + self.R * ((u + G - 1) // G)
        return base + k

    # ---------------- instruction emitters ----------------
    def mfma(self, acc, a, b, c=None):
        c = acc if c is None else c
        self.emit("v_mfma_f32_16x16x4_f32 %s, a%d, v%d, %s" % (self.vr(acc), a, b, self.vr(c)))

    def refill(self, u, k):
        where, ut = self.refill_target(u, k)
        a0 = self.slot_base(u, k)
        assert a0 == self.slot_base(ut, k)
        self.emit("s_add_u32 s%d, s%d, %d" % (self.S_T, self.S_LNEXT if where == "next" else self.S_LCUR, self.frag_index(ut, k) * 1024))
        self.emit("buffer_load_dwordx4 a[%d:%d], v%d, %%[rsrc], s%d offen" % (a0, a0 + 3, self.A["VOFF"], self.S_T))
        self.vm_seq += 1
        self.ring_id[(ut, k)] = self.vm_seq

    def ds_read(self, dst, addr, off=0, tag=None):
        self.emit("ds_read_b128 %s, v%d%s" % (self.vr(dst), self.A[addr], (" offset:%d" % off) if off else ""))
        return self.ds_issue(tag)

    def ds_write(self, addr, src, off=0):
        self.emit("ds_write_b128 v%d, %s%s" % (self.A[addr], self.vr(src), (" offset:%d" % off) if off else ""))
        return self.ds_issue()

    def lrelu(self, dst, src, r, tmp):
        """dst[r] = max(src[r], 0.01 * src[r])  (nn.LeakyReLU(0.01): fmaxf(x, x * 0.01f))"""
        self.emit("v_mul_f32_e32 v%d, s%d, v%d" % (tmp, self.S_C01, src + r))
        self.emit("v_max_f32_e32 v%d, v%d, v%d" % (dst + r, src + r, tmp))

    def lrelu_tile(self, dst, src, tmp):
        """the four registers of a tile: independent multiplies first, then the maxes"""
        for r in range(4):
            self.emit("v_mul_f32_e32 v%d, s%d, v%d" % (tmp + r, self.S_C01, src + r))
        for r in range(4):
            self.emit("v_max_f32_e32 v%d, v%d, v%d" % (dst + r, src + r, tmp + r))

    def stamp(self, idx):
        """diagnostic: add the cycles since the previous stamp to lane idx of a[ring_regs] (drains the LDS queue: lgkmcnt)"""
        if not self.stamps:
            return
        e, tmp, acc = self.emit, self.VB + self.n_vgpr - 1, self.ring_regs
        e("s_memtime s[%d:%d]" % self.S_NOW)
        e("s_waitcnt lgkmcnt(0)")
        self.ds_done = self.ds_seq
        e("s_sub_u32 s%d, s%d, s%d" % (self.S_DT, self.S_NOW[0], self.S_LAST))
        e("s_mov_b32 s%d, s%d" % (self.S_LAST, self.S_NOW[0]))
        e("v_accvgpr_read_b32 v%d, a%d" % (tmp, acc))
        e("s_nop 4")
        e("v_readlane_b32 s%d, v%d, %d" % (self.S_NOW[0], tmp, idx))
        e("s_nop 4")
        e("s_add_u32 s%d, s%d, s%d" % (self.S_NOW[0], self.S_NOW[0], self.S_DT))
        e("s_nop 4")
        e("v_writelane_b32 v%d, s%d, %d" % (tmp, self.S_NOW[0], idx))
        e("s_nop 1")
        e("v_accvgpr_write_b32 a%d, v%d" % (acc, tmp))

    # ---------------- one layer pass ----------------
    def step_mfmas(self, u):
        """ordered MFMA list of step u: (acc register, ring element e, k-step r, wave-0-only)"""
        NT = self.NT
        own = (u % G == 0)
        w0only = own and not (u + G - 1 < NT)
        ops = []
        if u == 0:
            for r in range(4):
                ops += [(0, 3 * r + 0, r, False), (1, 3 * r + 1, r, False)]
            for r in range(4):
                ops += [(2, 3 * r + 2, r, False), (3, 12 + r, r, False)]
        elif u == NT - 1:
            for r in range(4):
                ops += [(0, 3 * r + 0, r, False), (1, 3 * r + 1, r, False)]
            for r in range(4):
                ops.append((2, 3 * r + 2, r, False))
                if own:
                    ops.append((3, 12 + r, r, w0only))
        else:
            for r in range(4):
                for i in range(3):
                    ops.append((i, 3 * r + i, r, False))
                if own:
                    ops.append((3, 12 + r, r, w0only))
        return ops

    def layer(self):
        NT, A = self.NT, self.A
        accs = self.ACC + [self.ACCR]
        X, Y, HO = self.X, self.Y, self.HO
        T4 = self.T + [self.TR]
        tmp = Y  # scratch VGPRs of the LeakyReLU multiplies: Y[0..3]
        for u in range(NT):
            ops = self.step_mfmas(u)
            # last reader of each fragment of this step (only MFMAs that every wavefront executes count; the wave-0-only
            # remainder MFMAs of the last owned step come last in their fragment anyway)
            last_use = {}
            for pos, (ai, e, r, w0) in enumerate(ops):
                last_use[e // 4 if e < 12 else 3] = pos
            side = {}  # position -> list of callables issued behind that MFMA

            def at(pos, fn):
                side.setdefault(pos, []).append(fn)

            bsrc = HO if u == 0 else self.B[u & 1]
            # ---- LDS read of the next step's B operand ----
            if u == 0:
                pass  # issued behind the barrier (below)
            elif u + 1 < NT:
                self.ds_read(self.B[(u + 1) & 1], "HW_IN", (u + 1) * 1024, tag="B%d" % (u + 1))

            # ---- shadow work of this step ----
            if u == 0:
                # leftover of the previous layer: accumulator 2 -> activation tile, remainder partial sums -> Ps
                # (measured, tools/ubench/mfma_valu.hip: VALU work does NOT hide under v_mfma_f32_16x16x4_f32 -- the f32 MFMA
                # and the VALU share the SIMD's pipe, each VALU instruction costs its 4+ cycles and the first one behind an
                # MFMA ~10 more -- so side work is issued in few large clusters, not spread over the gaps)
                at(2, lambda: [self.lrelu_tile(X, accs[2], tmp), self.ds_write("LO_TILE", X), self.ds_write("LO_PART", self.ACCR)])

                def barrier():
                    self.stamp(13)   # first half of step 0
                    self.wait_ds_all()
                    self.emit("s_barrier")
                    self.stamp(14)   # the barrier
                    self.ds_read(self.B[1], "HW_IN", 1024, tag="B1")
                at(7, barrier)
            if u == 1:
                # fold of the remainder tile: h = lrelu((p0 + p1) + (p2 + p3)) -> slot NT-1 of the input buffer (own copy)
                at(0, lambda: [self.ds_read(T4[k], "PL_IN", k * 1024, tag="P%d" % k) for k in range(4)])

                def fold():
                    self.wait_ds(self.tag_id["P3"])
                    for r in range(4):
                        self.emit("v_add_f32_e32 v%d, v%d, v%d" % (X + r, T4[0] + r, T4[1] + r))
                    for r in range(4):
                        self.emit("v_add_f32_e32 v%d, v%d, v%d" % (Y + r, T4[2] + r, T4[3] + r))
                    for r in range(4):
                        self.emit("v_add_f32_e32 v%d, v%d, v%d" % (X + r, X + r, Y + r))
                    self.lrelu_tile(X, X, Y)
                    self.ds_write("FOLDW", X)
                at(5, fold)
            if u == NT - 3:
                # bias of the next layer -> T (C operands of its first MFMAs)
                at(0, lambda: [self.emit("v_add_u32_e32 v%d, %d, v%d" % (A["BIAS_A"], 16 * NT * 4, A["BIAS_A"])),
                               self.emit("v_add_u32_e32 v%d, %d, v%d" % (A["BIAS_R"], 16 * NT * 4, A["BIAS_R"])),
                               self.ds_read(T4[0], "BIAS_A", 0, tag="T0"), self.ds_read(T4[1], "BIAS_A", 256, tag="T1"),
                               self.ds_read(T4[2], "BIAS_A", 512, tag="T2"), self.ds_read(T4[3], "BIAS_R", 0, tag="T3")])
            if u == NT - 2:
                def mask_tr():
                    self.wait_ds(self.tag_id["T3"])
                    for r in range(4):  # partial sum 0 carries the bias, the others start from +0
                        self.emit("v_cndmask_b32_e64 v%d, 0, v%d, s[%d:%d]" % (self.TR + r, self.TR + r, self.S_W0[0], self.S_W0[1]))
                at(3, mask_tr)
                # duplicate slot of the own tile (tiles 0..2 are stored twice: rotated reads at immediate offsets)
                at(3, lambda: [self.emit("v_add_u32_e32 v%d, %d, v%d" % (A["DUP"], NT * 1024, A["HW_OUT"])),
                               self.emit("v_cndmask_b32_e64 v%d, v%d, v%d, s[%d:%d]" % (A["DUP"], A["DUMMY"], A["DUP"], self.S_W3[0], self.S_W3[1]))])
            if u == NT - 1:
                # own tile (accumulator 0) and accumulator 1 are complete after position 7
                at(10, lambda: [self.lrelu_tile(HO, accs[0], tmp), self.ds_write("HW_OUT", HO), self.ds_write("DUP", HO),
                               self.lrelu_tile(X, accs[1], tmp), self.ds_write("HW_OUT", X, 4096)])

            # ---- refills: right behind the last MFMA that reads the fragment ----
            for k, pos in last_use.items():
                at(pos, (lambda k=k: self.refill(u, k)))

            # ---- emission ----
            if u >= 1:
                self.wait_ds(self.tag_id["B%d" % u])
            if u == NT - 1:
                # positions 8.. of the last step differ between wave 0 (K-slice owner) and the others: two bodies
                self.emit_ops(u, ops[:8], 0, side, bsrc, accs, T4)
                tail = ops[8:]
                lbl = "%="
                self.emit("s_cmp_eq_u64 s[%d:%d], 0" % self.S_W0)
                self.emit("s_cbranch_scc1 .Lnw0_" + lbl)
                st = (self.ds_seq, self.ds_done, self.vm_seq, self.vm_done, dict(self.ring_id), dict(self.tag_id))
                self.emit_ops(u, tail, 8, side, bsrc, accs, T4, dense=True)
                self.emit("s_branch .Ljoin_" + lbl)
                end0 = (self.ds_seq, self.ds_done, self.vm_seq, self.vm_done)
                self.emit(".Lnw0_" + lbl + ":")
                (self.ds_seq, self.ds_done, self.vm_seq, self.vm_done, self.ring_id, self.tag_id) = st
                self.emit_ops(u, [o for o in tail if not o[3]], 8, side, bsrc, accs, T4, dense=True, renumber=tail)
                assert (end0[0], end0[2]) == (self.ds_seq, self.vm_seq), "both bodies must issue the same memory operations"
                self.ds_done, self.vm_done = min(self.ds_done, end0[1]), min(self.vm_done, end0[3])  # what both guarantee
                self.emit(".Ljoin_" + lbl + ":")
            else:
                self.emit_ops(u, ops, 0, side, bsrc, accs, T4)
            self.stamp(u)

    def emit_ops(self, u, ops, pos0, side, bsrc, accs, T4, dense=False, renumber=None):
        """MFMAs of a step from position pos0 on, each followed by its side work.  `renumber`: the full op list whose
        positions the side table refers to (body without the wave-0-only MFMAs: side work of a skipped position is issued
        behind the previous emitted MFMA)."""
        full = renumber if renumber is not None else ops
        pending = []
        idx = 0
        for k, op in enumerate(full):
            pos = pos0 + k
            present = op in ops if renumber is not None else True
            if present:
                ai, e, r, w0 = op
                key = (u, e // 4 if e < 12 else 3)
                self.wait_vm(self.ring_id[key])
                c = None
                if u == 0 and r == 0:
                    c = T4[ai]
                    self.wait_ds(self.tag_id["T%d" % ai])
                self.mfma(accs[ai], self.areg(u, e), bsrc + r, c)
            for fn in side.get(pos, []):
                fn()

    # ---------------- whole statements ----------------
    def gen_layers(self):
        A = self.A
        self.lines = []
        e = self.emit
        lbl = "%="
        # ---- entry ----
        e("s_waitcnt lgkmcnt(0)")
        ins = ["HW_IN", "HW_OUT", "FW_IN", "FW_OUT", "PL_IN", "PL_OUT", "PW_IN", "PW_OUT", "BIAS_A", "BIAS_R", "VOFF", "DUMMY",
               "W0A", "W0R", "WLA"]
        for n in ins:
            e("v_mov_b32_e32 v%d, %%[%s]" % (A[n], n.lower()))
        e("s_mov_b32 s%d, 0" % self.S_L)
        e("s_mov_b32 s%d, %%[nl]" % self.S_NL)
        e("s_mov_b32 s%d, %%[lbytes]" % self.S_LB)
        e("s_mov_b32 s%d, %%[hid0]" % self.S_H0)
        e("s_mov_b32 s%d, 0x3c23d70a" % self.S_C01)
        e("s_cmp_eq_u32 %[wave], 0")
        e("s_cselect_b64 s[%d:%d], -1, 0" % self.S_W0)
        e("s_cmp_lt_u32 %[wave], 3")
        e("s_cselect_b64 s[%d:%d], -1, 0" % self.S_W3)
        for n in ("LO_TILE", "LO_PART", "FOLDW"):
            e("v_mov_b32_e32 v%d, v%d" % (A[n], A["DUMMY"]))
        e("v_add_u32_e32 v%d, %d, v%d" % (A["DUP"], self.NT * 1024, A["HW_IN"]))
        e("v_cndmask_b32_e64 v%d, v%d, v%d, s[%d:%d]" % (A["DUP"], A["DUMMY"], A["DUP"], self.S_W3[0], self.S_W3[1]))
        T4 = self.T + [self.TR]
        # ---- layer 0: Linear(2, N) + LeakyReLU on the VALU, h = lrelu(fmaf(w1, x1, fmaf(w0, x0, b))); rows {b, w0, w1, 0} in LDS.
        # Row tiles wave, wave + 4, wave + 8 (own tile -> HO and the B operand of step 0) and the remainder tile (every
        # wavefront writes it: identical bits).  Two 16-register row buffers: the next tile's rows are in flight meanwhile.
        self.reset_counters()
        RB = [T4, [self.B[0], self.B[1], self.X, self.Y]]
        tiles = [("W0A", 0, self.HO, [("HW_IN", 0), ("DUP", 0)]), ("W0A", 1024, self.ACC[0], [("HW_IN", 4096)]),
                 ("W0A", 2048, self.ACC[1], [("HW_IN", 8192)]), ("W0R", 0, self.ACC[2], [("FW_IN", 0)])]
        rd = {}

        def l0_reads(ti):
            areg, off, _, _ = tiles[ti]
            rd[ti] = [self.ds_read(RB[ti & 1][r], areg, off + 16 * r) for r in range(4)]
        l0_reads(0)
        l0_reads(1)
        for ti, (areg, off, dst, stores) in enumerate(tiles):
            rows = RB[ti & 1]
            self.wait_ds(rd[ti][3])
            for r in range(4):
                e("v_fma_f32 v%d, v%d, %%[x0], v%d" % (rows[r], rows[r] + 1, rows[r]))
            for r in range(4):
                e("v_fma_f32 v%d, v%d, %%[x1], v%d" % (rows[r], rows[r] + 2, rows[r]))
            for r in range(4):
                e("v_mul_f32_e32 v%d, s%d, v%d" % (rows[r] + 3, self.S_C01, rows[r]))
            for r in range(4):
                e("v_max_f32_e32 v%d, v%d, v%d" % (dst + r, rows[r], rows[r] + 3))
            if ti + 2 < len(tiles):
                l0_reads(ti + 2)
            for (areg2, off2) in stores:
                self.ds_write(areg2, dst, off2)
        # bias of hidden layer 0 -> C operands of its first MFMAs
        e("ds_read_b128 %s, v%d" % (self.vr(T4[0]), A["BIAS_A"]))
        e("ds_read_b128 %s, v%d offset:256" % (self.vr(T4[1]), A["BIAS_A"]))
        e("ds_read_b128 %s, v%d offset:512" % (self.vr(T4[2]), A["BIAS_A"]))
        e("ds_read_b128 %s, v%d" % (self.vr(T4[3]), A["BIAS_R"]))
        e("s_waitcnt lgkmcnt(0)")
        for r in range(4):
            e("v_cndmask_b32_e64 v%d, 0, v%d, s[%d:%d]" % (self.TR + r, self.TR + r, self.S_W0[0], self.S_W0[1]))
        e("s_mov_b32 s%d, s%d" % (self.S_LNEXT, self.S_H0))  # refill source of layer l: layer l + 1, or layer 0 after the last
        self.stamp(15)   # everything outside the layer loop
        e(".Lloop_" + lbl + ":")
        # at the loop top s_lnext is the offset of layer l; the refills of layer l stream layer (l + 1 < L ? l + 1 : 0)
        e("s_mov_b32 s%d, s%d" % (self.S_LCUR, self.S_LNEXT))
        e("s_add_u32 s%d, s%d, s%d" % (self.S_LNEXT, self.S_LNEXT, self.S_LB))
        e("s_add_u32 s%d, s%d, 1" % (self.S_T, self.S_L))
        e("s_cmp_lt_u32 s%d, s%d" % (self.S_T, self.S_NL))
        e("s_cselect_b32 s%d, s%d, s%d" % (self.S_LNEXT, self.S_LNEXT, self.S_H0))
        head = len(self.lines)
        # ---- the layer body: simulate to the fixed point of the wait counts, emit the fixed point ----
        self.reset_counters()
        # ring as primed by init / left by the previous pass: loads in refill order of a pass
        body = None
        for it in range(3):
            start = len(self.lines)
            if it == 0:
                for (u, k) in self.refill_order():
                    where, ut = self.refill_target(u, k)
                    if where == "next":
                        self.vm_seq += 1
                        self.ring_id[(ut, k)] = self.vm_seq
                for k in range(4):
                    self.ds_issue("T%d" % k)
                self.ds_done = self.ds_seq
            self.layer()
            self.bookkeeping()
            text = self.lines[start:]
            del self.lines[start:]
            if it >= 1:
                if body is not None:
                    assert body == text, "wait counts did not reach a fixed point"
                body = text
        self.lines += body
        # ---- loop control ----
        e("s_add_u32 s%d, s%d, 1" % (self.S_L, self.S_L))
        e("s_cmp_lt_u32 s%d, s%d" % (self.S_L, self.S_NL))
        e("s_cbranch_scc1 .Lloop_" + lbl)
        # ---- exit: the last layer's accumulator 2 and partial sums (not pipelined: nothing follows) ----
        e("s_nop 7")
        e("s_nop 1")
        self.lrelu_tile(self.X, self.ACC[2], self.Y)
        e("ds_write_b128 v%d, %s" % (A["LO_TILE"], self.vr(self.X)))
        e("ds_write_b128 v%d, %s" % (A["LO_PART"], self.vr(self.ACCR)))
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        self.stamp(12)  # (diagnostic: charged to the last step)
        # ---- Linear(N, 1): four partial fmaf chains (one per lane group q) over kt, r; fixed combine tree ((p0+p1)+(p2+p3)) + bl.
        # After the last swap the *_IN names are the buffers the last hidden layer wrote.
        self.reset_counters()
        PART, TMP = A["PART"], A["LO_TILE"]
        e("v_add_u32_e32 v%d, %d, v%d" % (A["LO_PART"], -(self.NT - 1) * 1024 & 0xffffffff, A["FW_IN"]))  # slot 0 of the buffer, this lane
        HB = "LO_PART"
        # register quads free here: the accumulators, the bias quads, both B buffers, HO, X, Y
        Q = [self.ACC[0], self.ACC[1], self.ACC[2], self.ACCR, T4[0], T4[1], T4[2], T4[3], self.B[0], self.B[1], self.HO, self.X, self.Y]
        NTF = self.NT - 1   # tiles 0 .. NT-2 are activations; tile NT-1 is folded from the partial sums
        # remainder tile first (its result waits in HO): h = lrelu((p0 + p1) + (p2 + p3)), weights wl[16 (NT-1) + 4 q + r] in TR
        pr = [Q[8], Q[9], Q[11], Q[12]]
        idp = [self.ds_read(pr[k], "PL_IN", k * 1024) for k in range(4)]
        idw12 = self.ds_read(Q[7], "WLA", (self.NT - 1) * 64)
        grpA = [(Q[0], Q[4]), (Q[1], Q[5]), (Q[2], Q[6])]
        grpB = [(Q[3], Q[8]), (Q[9], Q[11])]
        groups = []
        kt = 0
        while kt < NTF:
            g = grpA if len(groups) % 2 == 0 else grpB
            n = min(len(g), NTF - kt)
            groups.append([(kt + i, g[i][0], g[i][1]) for i in range(n)])
            kt += n
        ids = {}

        def ll_reads(gi):
            for (k, hreg, wreg) in groups[gi]:
                ids[(k, "h")] = self.ds_read(hreg, HB, k * 1024)
                ids[(k, "w")] = self.ds_read(wreg, "WLA", k * 64)
        ll_reads(0)
        self.wait_ds(idp[3])
        for r in range(4):
            e("v_add_f32_e32 v%d, v%d, v%d" % (pr[0] + r, pr[0] + r, pr[1] + r))
        for r in range(4):
            e("v_add_f32_e32 v%d, v%d, v%d" % (pr[2] + r, pr[2] + r, pr[3] + r))
        for r in range(4):
            e("v_add_f32_e32 v%d, v%d, v%d" % (pr[0] + r, pr[0] + r, pr[2] + r))
        self.lrelu_tile(self.HO, pr[0], pr[1])
        e("v_mov_b32_e32 v%d, 0" % PART)
        for gi, grp in enumerate(groups):
            if gi + 1 < len(groups):
                ll_reads(gi + 1)
            for (k, hreg, wreg) in grp:
                self.wait_ds(ids[(k, "w")])
                for r in range(4):
                    e("v_fmac_f32_e32 v%d, v%d, v%d" % (PART, wreg + r, hreg + r))
        self.wait_ds(idw12)
        for r in range(4):
            e("v_fmac_f32_e32 v%d, v%d, v%d" % (PART, Q[7] + r, self.HO + r))
        # pair = part + part[lane ^ 16]; out = (pair + pair[lane ^ 32]) + bl   (swap both copies, add: commutative, same bits)
        e("v_mov_b32_e32 v%d, v%d" % (TMP, PART))
        e("s_nop 1")
        e("v_permlane16_swap_b32_e32 v%d, v%d" % (TMP, PART))
        e("v_add_f32_e32 v%d, v%d, v%d" % (PART, TMP, PART))
        e("v_mov_b32_e32 v%d, v%d" % (TMP, PART))
        e("s_nop 1")
        e("v_permlane32_swap_b32_e32 v%d, v%d" % (TMP, PART))
        e("v_add_f32_e32 v%d, v%d, v%d" % (PART, TMP, PART))
        e("v_add_f32_e32 %%[out], %%[bl], v%d" % PART)
        # the next evaluation's layer 0 rewrites buffer 0: with an even number of hidden layers that is the buffer just read
        e("s_bitcmp1_b32 s%d, 0" % self.S_NL)
        e("s_cbranch_scc1 .Lodd_" + lbl)
        e("s_barrier")
        e(".Lodd_" + lbl + ":")
        return self.lines

    def bookkeeping(self):
        """end of a layer pass: where the leftover of this layer goes, and the in/out swap of the double buffers"""
        A, e = self.A, self.emit
        e("v_add_u32_e32 v%d, %d, v%d" % (A["LO_TILE"], 8192, A["HW_OUT"]))
        e("v_mov_b32_e32 v%d, v%d" % (A["LO_PART"], A["PW_OUT"]))
        e("v_mov_b32_e32 v%d, v%d" % (A["FOLDW"], A["FW_OUT"]))
        for a, b in (("HW_IN", "HW_OUT"), ("FW_IN", "FW_OUT"), ("PL_IN", "PL_OUT"), ("PW_IN", "PW_OUT")):
            e("v_swap_b32 v%d, v%d" % (A[a], A[b]))

    def refill_order(self):
        """fragment slots in the order one layer pass reloads them"""
        order = []
        for u in range(self.NT):
            ops = self.step_mfmas(u)
            last_use = {}
            for pos, (ai, e, r, w0) in enumerate(ops):
                last_use[e // 4 if e < 12 else 3] = pos
            # same traversal as layer(): by position, then insertion order of `at`
            bypos = {}
            for k, pos in last_use.items():
                bypos.setdefault(pos, []).append(k)
            for pos in sorted(bypos):
                for k in bypos[pos]:
                    order.append((u, k))
        return order

    def gen_init(self):
        self.lines = []
        e = self.emit
        e("s_mov_b32 s%d, %%[hid0]" % self.S_H0)
        if self.stamps:
            e("v_accvgpr_write_b32 a%d, 0" % self.ring_regs)
        for (u, k) in self.refill_order():   # what the last pass of an evaluation leaves in flight: the start of layer 0
            where, ut = self.refill_target(u, k)
            if where != "next":
                continue
            a0 = self.slot_base(ut, k)
            e("s_add_u32 s%d, s%d, %d" % (self.S_T, self.S_H0, self.frag_index(ut, k) * 1024))
            e("buffer_load_dwordx4 a[%d:%d], %%[voff], %%[rsrc], s%d offen" % (a0, a0 + 3, self.S_T))
        return self.lines


def cstring(lines):
    return "\n".join('  "%s\\n\\t"' % l for l in lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nt", type=int, default=13)
    ap.add_argument("--pd", type=int, default=7)
    ap.add_argument("--stamps", action="store_true")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    g = Gen(a.nt, a.pd, stamps=a.stamps)
    layers = list(g.gen_layers())
    init = list(Gen(a.nt, a.pd, stamps=a.stamps).gen_init())
    nmf = sum(1 for l in layers if l.startswith("v_mfma"))
    with open(a.out, "w") as f:
        f.write("// GENERATED by tools/gen_mlp_asm.py --nt %d -- do not edit.\n" % a.nt)
        f.write("// %d instructions, %d MFMAs in the layer body (both bodies of the last step counted).\n" % (len(layers), nmf))
        f.write("#define IONODE_MLPASM_VB_%d %d\n" % (a.nt, g.VB))
        f.write("#define IONODE_MLPASM_RING_REGS_%d %d\n" % (a.nt, g.ring_regs))
        f.write("#define IONODE_MLPASM_INIT_%d \\\n%s\n" % (a.nt, cstring(init).replace("\n", " \\\n")))
        f.write("#define IONODE_MLPASM_LAYERS_%d \\\n%s\n" % (a.nt, cstring(layers).replace("\n", " \\\n")))
        vclob = ", ".join('"v%d"' % r for r in range(g.VB, g.VB + g.n_vgpr))
        aclob = ", ".join('"a%d"' % r for r in range(g.ring_regs + (1 if a.stamps else 0)))
        sclob = ", ".join('"s%d"' % r for r in range(g.SB, g.SB + g.n_sgpr))
        f.write("#define IONODE_MLPASM_CLOBBER_V_%d %s\n" % (a.nt, vclob))
        f.write("#define IONODE_MLPASM_CLOBBER_A_%d %s\n" % (a.nt, aclob))
        f.write("#define IONODE_MLPASM_CLOBBER_S_%d %s\n" % (a.nt, sclob))


if __name__ == "__main__":
    main()
