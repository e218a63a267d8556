#!/usr/bin/env python3
"""Generator of the hand-scheduled MLP evaluation stream of the N = 200 tile (gfx950 inline asm).

    python3 tools/gen_mlp_asm.py --nt 13 --pd 7 --ns 1 --out neural-ode-ion-channels_amd/csrc/mlp_asm_nt13.inc
    python3 tools/gen_mlp_asm.py --nt 13 --pd 7 --ns 2 --out neural-ode-ion-channels_amd/csrc/mlp_asm_nt13x2.inc

What it emits (consumed by MlpTile in csrc/ionode_device.hpp): two C string macros (suffix <NT> or <NT>x<NS>),

    IONODE_MLPASM_INIT_*    prime the weight ring with hidden layer 0 (kernel start)
    IONODE_MLPASM_LAYERS_*  one whole evaluation net([x0, x1]) of the tile: Linear(2, N) + LeakyReLU on the VALU, for
                            l = 0 .. L-1 one pass of NT k-tile steps of v_mfma_f32_16x16x4_f32 with the layer boundary
                            software-pipelined, Linear(N, 1) on the VALU

plus the clobber lists.  NS = number of 16-trajectory COLUMN SETS per tile: with NS = 2 (32 trajectories per workgroup,
launches of at least two 16-tiles per compute unit) every weight fragment feeds two MFMAs -- half the fragment stream per
FLOP -- and wavefronts 0, 1 carry the Runge-Kutta state of column set 0, wavefronts 2, 3 that of set 1, so the scalar
integrator work is replicated twice instead of four times per trajectory.  The stage inputs are exchanged through LDS; the
result is returned for the wavefront's own set.

The arithmetic is the canonical accumulation order of ionode_device.hpp (MlpTile::eval): every accumulator's chain visits
the same (k-tile, k-step) sequence, so the bits do not change; what changes is the order in which INDEPENDENT
accumulators are interleaved, where the epilogue sits, and who allocates the registers:

  * fixed register assignment: weight ring in AGPRs (loaded by buffer_load straight into AGPRs, read by the MFMAs as
    srcA), accumulators / bias / B operands / temporaries in the top VGPRs; hipcc sees one opaque statement with clobbers and
    keeps its own values elsewhere -- no AGPR<->VGPR copies, no spills in the stream;
  * the LAST step of a layer finishes the wavefront's own tile (accumulator 0) and accumulator 1 first; the FIRST step of
    the next layer starts on them (B operand = own tile, from registers, C operand = bias prefetched two steps earlier)
    while accumulator 2 and the remainder tile's partial sums of the previous layer are post-processed; the workgroup
    barrier sits in the middle of that step and the first LDS-read B operand is issued right behind it;
  * measured (tools/ubench/mfma_valu.hip): VALU work does NOT hide under v_mfma_f32_16x16x4_f32 -- the f32 MFMA and the VALU
    share the SIMD's pipe, every VALU instruction costs its 4+ cycles and the first one behind an MFMA ~10 more -- so side
    work (LeakyReLU, fold, bookkeeping) is issued in few clusters and its instruction count is what matters;
  * s_waitcnt counts are derived by simulation of the in-order vmcnt / lgkmcnt queues (the emitted layer pass is a fixed
    point of the simulation).

Hazards (probed with hipcc on gfx950): MFMA result -> VALU / LDS-store read: 10 wait states (here: at least two younger
MFMAs issued in between = 64 cycles, or explicit s_nop); VALU write -> MFMA operand: 2 wait states; dependent MFMA on the
same accumulator: interlocked.
"""
import argparse

G = 4  # wavefronts per tile


class Gen:
    def __init__(self, nt, pd=7, ns=1, sb=84, stamps=False, short12=True):
        self.short12 = short12   # k-tile 12 in two MFMAs per accumulator (N <= 200: run-time switch %[sw])
        self.NT = nt
        self.NS = ns
        self.F = nt // G
        self.R = nt - G * self.F
        assert self.R == 1 and self.F == 3, "written for N = 200 (NT = 13): three full row tiles + one K-split remainder tile per wavefront"
        self.NOWN = (nt + G - 1) // G
        self.PD = pd            # ring depth in k-tile steps (slot of step u: u mod PD); NT <= 2 PD
        assert pd <= nt <= 2 * pd
        self.RD = self.NOWN if pd == nt else 2   # ring depth of the remainder fragments (owned steps 0, G, 2G, ...)
        assert self.NOWN % self.RD == 0
        self.ring_regs = 12 * pd + 4 * self.RD
        self.HSET, self.PSET = 16 * 1024, 4 * 1024          # LDS bytes between the column sets of an activation / partial-sum buffer
        # --- fixed VGPR map (top of the file) ---
        regs = []

        def alloc(n):
            regs.append(n)
            return sum(regs) - n
        self.ACC = [[alloc(4) for _ in range(4)] for _ in range(ns)]       # [set][0..2 full tiles, 3 remainder partial]
        self.T = [alloc(4) for _ in range(4)]                               # bias of the coming layer (C operands); fold temporaries
        self.B = [[alloc(4) for _ in range(ns)] for _ in range(2)]          # [buffer][set]
        self.HO = [alloc(4) for _ in range(ns)]                             # own tile of the coming layer (B operand of step 0)
        self.X, self.Y = alloc(4), alloc(4)
        names = ["HW_IN", "HW_OUT", "FW_IN", "FW_OUT", "PL_IN", "PL_OUT", "PW_IN", "PW_OUT",
                 "BIAS_A", "BIAS_R", "VOFF", "DUMMY", "W0A", "W0R", "WLA", "PART", "TMPA", "XCHW", "XCHR", "QOFF"]
        for cs in range(ns):
            names += ["DUP%d" % cs, "LO_TILE%d" % cs, "LO_PART%d" % cs, "FOLDW%d" % cs, "HWS12_%d" % cs]
        self.A = {n: alloc(1) for n in names}
        if sum(regs) % 2:
            alloc(1)                                      # (register pairs are 64-bit aligned: VB is a multiple of 4, so parity here is parity there)
        self.BX = [alloc(2) for _ in range(ns)]          # short form of k-tile 12: its two B operands per column set
        self.STAMPV = alloc(1)
        self.n_vgpr = (sum(regs) + 3) // 4 * 4
        self.VB = 256 - self.n_vgpr
        fix = lambda x: x + self.VB
        self.ACC = [[fix(a) for a in row] for row in self.ACC]
        self.T = [fix(a) for a in self.T]
        self.B = [[fix(a) for a in row] for row in self.B]
        self.HO = [fix(a) for a in self.HO]
        self.BX = [fix(a) for a in self.BX]
        self.X, self.Y, self.STAMPV = fix(self.X), fix(self.Y), fix(self.STAMPV)
        self.A = {k: fix(v) for k, v in self.A.items()}
        # --- fixed SGPR map ---
        self.SB = sb
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
        # round 5, the short form of k-tile 12 (see layer()): lanes of the groups q >= 2, the wavefront that owns the short form (or 99), a scratch
        self.S_QHI = (s(12), s(13))
        self.S_SW = s(14)
        self.S_X = s(15)
        self.n_sgpr = 16
        self.stamps = stamps   # diagnostic build: s_memtime deltas summed per position in the lanes of AGPR a[ring_regs]
        if stamps:
            self.S_NOW, self.S_LAST, self.S_DT = (s(16), s(17)), s(18), s(19)
            self.n_sgpr = 20
        self.lines = []
        self.ool = []
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

    def lrelu_tile(self, dst, src, tmp):
        """dst[r] = max(src[r], 0.01 * src[r]) (nn.LeakyReLU(0.01): fmaxf(x, x * 0.01f)) for the four registers of a tile:
        independent multiplies first, then the maxes"""
        for r in range(4):
            self.emit("v_mul_f32_e32 v%d, s%d, v%d" % (tmp + r, self.S_C01, src + r))
        for r in range(4):
            self.emit("v_max_f32_e32 v%d, v%d, v%d" % (dst + r, src + r, tmp + r))

    def fold_tile(self, dst, p, tmp):
        """dst = lrelu((p0 + p1) + (p2 + p3)), the remainder tile's fixed combine tree; p = four register quads (p[0] is overwritten)"""
        for r in range(4):
            self.emit("v_add_f32_e32 v%d, v%d, v%d" % (p[0] + r, p[0] + r, p[1] + r))
        for r in range(4):
            self.emit("v_add_f32_e32 v%d, v%d, v%d" % (p[2] + r, p[2] + r, p[3] + r))
        for r in range(4):
            self.emit("v_add_f32_e32 v%d, v%d, v%d" % (p[0] + r, p[0] + r, p[2] + r))
        self.lrelu_tile(dst, p[0], tmp)

    def stamp(self, idx):
        """diagnostic: add the cycles since the previous stamp to lane idx of a[ring_regs] (drains the LDS queue: lgkmcnt)"""
        if not self.stamps:
            return
        e, tmp, acc = self.emit, self.STAMPV, self.ring_regs
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
        """ordered MFMA list of step u: (column set, accumulator 0..3, ring element e, k-step r, wave-0-only).  An element e is
        used by NS consecutive MFMAs (one per column set)."""
        NT, NS = self.NT, self.NS
        own = (u % G == 0)
        w0only = own and not (u + G - 1 < NT)
        ops = []
        sets = range(NS)
        if u == 0 or u == NT - 1:
            # first the own tile and accumulator 1 (they end a layer early / start the next one from registers), then the rest
            for r in range(4):
                for i in (0, 1):
                    ops += [(cs, i, 3 * r + i, r, False) for cs in sets]
            for r in range(4):
                ops += [(cs, 2, 3 * r + 2, r, False) for cs in sets]
                if own:
                    ops += [(cs, 3, 12 + r, r, w0only) for cs in sets]
        else:
            for r in range(4):
                for i in range(3):
                    ops += [(cs, i, 3 * r + i, r, False) for cs in sets]
                if own:
                    ops += [(cs, 3, 12 + r, r, w0only) for cs in sets]
        return ops

    def layer(self):
        NT, NS, A = self.NT, self.NS, self.A
        X, Y = self.X, self.Y
        T4 = self.T
        half = 8 * NS   # MFMAs of the first half of steps 0 and NT-1 (accumulators 0 and 1 of every set)
        for u in range(NT):
            ops = self.step_mfmas(u)
            last_use = {}
            for pos, (cs, ai, e, r, w0) in enumerate(ops):
                last_use[e // 4 if e < 12 else 3] = pos
            side = {}  # position -> list of callables issued behind that MFMA

            def at(pos, fn):
                side.setdefault(pos, []).append(fn)

            # ---- LDS reads of the next step's B operands ----
            if 1 <= u and u + 1 < NT:
                for cs in range(NS):
                    self.ds_read(self.B[(u + 1) & 1][cs], "HW_IN", cs * self.HSET + (u + 1) * 1024, tag="B%d_%d" % (u + 1, cs))
            if self.short12 and u == NT - 5:
                # the two B operands of the short form of k-tile 12 (every wavefront meets that tile once, on one of the four steps that follow):
                # lane group q wants h[192 + 4 (q & 1) + (q >> 1)] and the value two further: the dwords at +0 and +8 of its own position in slot
                # 12 stepped back by (q >> 1) * 508 bytes (group q & 1's float4, component q >> 1).  The slot is final: the fold wrote it at step 1.
                for cs in range(NS):
                    self.emit("v_sub_u32_e32 v%d, v%d, v%d" % (A["HWS12_%d" % cs], A["FW_IN"], A["QOFF"]))
                    if cs:
                        self.emit("v_add_u32_e32 v%d, %d, v%d" % (A["HWS12_%d" % cs], cs * self.HSET, A["HWS12_%d" % cs]))
                    self.emit("ds_read2_b32 v[%d:%d], v%d offset1:2" % (self.BX[cs], self.BX[cs] + 1, A["HWS12_%d" % cs]))
                    self.ds_issue("BX%d" % cs)

            # ---- side work of this step ----
            if u == 0:
                # leftover of the previous layer: accumulator 2 -> activation tile, remainder partial sums -> Ps
                def leftover():
                    for cs in range(NS):
                        self.lrelu_tile(X, self.ACC[cs][2], Y)
                        self.ds_write("LO_TILE%d" % cs, X)
                        self.ds_write("LO_PART%d" % cs, self.ACC[cs][3])
                at(2, leftover)

                def barrier():
                    self.stamp(13)   # first half of step 0
                    self.wait_ds_all()
                    self.emit("s_barrier")
                    self.stamp(14)   # the barrier
                    for cs in range(NS):
                        self.ds_read(self.B[1][cs], "HW_IN", cs * self.HSET + 1024, tag="B1_%d" % cs)
                at(half - 1, barrier)
            if 1 <= u <= NS:
                # fold of the remainder tile of column set u-1: h = lrelu((p0 + p1) + (p2 + p3)) -> slot NT-1 of the input buffer
                cs = u - 1
                at(0, lambda cs=cs: [self.ds_read(T4[k], "PL_IN", cs * self.PSET + k * 1024, tag="P%d" % k) for k in range(4)])

                def fold(cs=cs):
                    self.wait_ds(self.tag_id["P3"])
                    self.fold_tile(X, T4, Y)
                    self.ds_write("FOLDW%d" % cs, X)
                at(5 * NS, fold)
            if u == NT - 3:
                # bias of the next layer -> T (C operands of its first MFMAs, shared by the column sets)
                at(0, lambda: [self.emit("v_add_u32_e32 v%d, %d, v%d" % (A["BIAS_A"], 16 * NT * 4, A["BIAS_A"])),
                               self.emit("v_add_u32_e32 v%d, %d, v%d" % (A["BIAS_R"], 16 * NT * 4, A["BIAS_R"])),
                               self.ds_read(T4[0], "BIAS_A", 0, tag="T0"), self.ds_read(T4[1], "BIAS_A", 256, tag="T1"),
                               self.ds_read(T4[2], "BIAS_A", 512, tag="T2"), self.ds_read(T4[3], "BIAS_R", 0, tag="T3")])
            if u == NT - 2:
                def mask_tr():
                    self.wait_ds(self.tag_id["T3"])
                    for r in range(4):  # partial sum 0 carries the bias, the others start from +0
                        self.emit("v_cndmask_b32_e64 v%d, 0, v%d, s[%d:%d]" % (T4[3] + r, T4[3] + r, self.S_W0[0], self.S_W0[1]))
                    # duplicate slot of the own tile (tiles 0..2 are stored twice: rotated reads at immediate offsets)
                    for cs in range(NS):
                        self.emit("v_add_u32_e32 v%d, %d, v%d" % (A["DUP%d" % cs], cs * self.HSET + NT * 1024, A["HW_OUT"]))
                        self.emit("v_cndmask_b32_e64 v%d, v%d, v%d, s[%d:%d]" % (A["DUP%d" % cs], A["DUMMY"], A["DUP%d" % cs], self.S_W3[0], self.S_W3[1]))
                at(3, mask_tr)
            if u == NT - 1:
                # own tile (accumulator 0) and accumulator 1 are complete after the first half
                def early():
                    for cs in range(NS):
                        self.lrelu_tile(self.HO[cs], self.ACC[cs][0], Y)
                        self.ds_write("HW_OUT", self.HO[cs], cs * self.HSET)
                        self.ds_write("DUP%d" % cs, self.HO[cs])
                        self.lrelu_tile(X, self.ACC[cs][1], Y)
                        self.ds_write("HW_OUT", X, cs * self.HSET + 4096)
                at(half + 2, early)

            # ---- refills: right behind the last MFMA that reads the fragment ----
            for k, pos in last_use.items():
                at(pos, (lambda k=k: self.refill(u, k)))

            # ---- emission ----
            if u >= 1:
                for cs in range(NS):
                    self.wait_ds(self.tag_id["B%d_%d" % (u, cs)])
            if self.short12 and NT - 4 <= u < NT - 1:
                # ---- the short form of k-tile 12 (round 5).  N = 200 pads its contraction index to 208: k-tile 12 holds k = 192 .. 199 and eight
                # padding columns, two of the four k of every MFMA.  The canonical order inside a k-tile is r-major, q-minor -- 192, 196, (200), (204),
                # 193, 197, ... -- so without the padding terms (exact no-ops) the chain is 192, 196, 193, 197 | 194, 198, 195, 199: TWO MFMAs whose
                # lane groups q = 0..3 supply those k (ionode_mlp_pack lays the tile's A fragments out that way; the B operands were read into BX
                # at step NT - 5).  The wavefront that meets k-tile 12 at this step (wave == NT - 1 - u) runs the MFMAs with r < 2 only: 6 of
                # its 169 MFMAs per layer less.  Two bodies issuing the same memory operations; the short one OUT OF LINE (a taken branch costs
                # ~40 cycles: the three wavefronts on the ordinary path fall through).
                lbl = "%=" + "_s%d" % u
                self.emit("s_cmp_eq_u32 s%d, %d" % (self.S_SW, NT - 1 - u))
                self.emit("s_cbranch_scc1 .Lshort_" + lbl)
                st = (self.ds_seq, self.ds_done, self.vm_seq, self.vm_done, dict(self.ring_id), dict(self.tag_id))
                self.emit_ops(u, ops, 0, side)
                self.emit(".Ljoin_" + lbl + ":")
                end0 = (self.ds_seq, self.ds_done, self.vm_seq, self.vm_done)
                main, self.lines = self.lines, []
                self.emit(".Lshort_" + lbl + ":")
                (self.ds_seq, self.ds_done, self.vm_seq, self.vm_done, self.ring_id, self.tag_id) = st
                self.short_b(u)
                self.emit_ops(u, [o for o in ops if o[3] < 2], 0, side, renumber=ops, bx=True)
                self.emit("s_branch .Ljoin_" + lbl)
                self.ool += self.lines
                self.lines = main
                assert (end0[0], end0[2]) == (self.ds_seq, self.vm_seq), "both bodies must issue the same memory operations"
                self.ds_done, self.vm_done = min(self.ds_done, end0[1]), min(self.vm_done, end0[3])
            elif u == NT - 1 and self.short12:
                # the last step: wave 0 owns the K-slice of the remainder tile AND meets k-tile 12 here: its bodies (the short form of everything,
                # or the full step when the short form is off) are out of line; the other wavefronts fall through the step without the
                # wave-0-only MFMAs
                lbl = "%="
                self.emit("s_cmp_eq_u64 s[%d:%d], 0" % self.S_W0)
                self.emit("s_cbranch_scc0 .Lw0_" + lbl)
                st = (self.ds_seq, self.ds_done, self.vm_seq, self.vm_done, dict(self.ring_id), dict(self.tag_id))
                self.emit_ops(u, [o for o in ops if not o[4]], 0, side, renumber=ops)
                self.emit(".Ljoin_" + lbl + ":")
                end0 = (self.ds_seq, self.ds_done, self.vm_seq, self.vm_done)
                main, self.lines = self.lines, []
                self.emit(".Lw0_" + lbl + ":")
                (self.ds_seq, self.ds_done, self.vm_seq, self.vm_done, self.ring_id, self.tag_id) = (st[0], st[1], st[2], st[3], dict(st[4]), dict(st[5]))
                self.emit("s_cmp_eq_u32 s%d, 0" % self.S_SW)
                self.emit("s_cbranch_scc0 .Lw0full_" + lbl)
                self.short_b(u)
                self.emit_ops(u, [o for o in ops if o[3] < 2], 0, side, renumber=ops, bx=True)
                self.emit("s_branch .Ljoin_" + lbl)
                end_s = (self.ds_seq, self.ds_done, self.vm_seq, self.vm_done)
                self.emit(".Lw0full_" + lbl + ":")
                (self.ds_seq, self.ds_done, self.vm_seq, self.vm_done, self.ring_id, self.tag_id) = st
                self.emit_ops(u, ops, 0, side)
                self.emit("s_branch .Ljoin_" + lbl)
                end_f = (self.ds_seq, self.ds_done, self.vm_seq, self.vm_done)
                self.ool += self.lines
                self.lines = main
                assert (end0[0], end0[2]) == (end_s[0], end_s[2]) == (end_f[0], end_f[2]), "all bodies must issue the same memory operations"
                self.ds_done, self.vm_done = min(end0[1], end_s[1], end_f[1]), min(end0[3], end_s[3], end_f[3])  # what all guarantee
            elif u == NT - 1:
                # the second half of the last step differs between wave 0 (K-slice owner) and the others: two bodies
                self.emit_ops(u, ops[:half], 0, side)
                tail = ops[half:]
                lbl = "%="
                self.emit("s_cmp_eq_u64 s[%d:%d], 0" % self.S_W0)
                self.emit("s_cbranch_scc1 .Lnw0_" + lbl)
                st = (self.ds_seq, self.ds_done, self.vm_seq, self.vm_done, dict(self.ring_id), dict(self.tag_id))
                self.emit_ops(u, tail, half, side)
                self.emit("s_branch .Ljoin_" + lbl)
                end0 = (self.ds_seq, self.ds_done, self.vm_seq, self.vm_done)
                self.emit(".Lnw0_" + lbl + ":")
                (self.ds_seq, self.ds_done, self.vm_seq, self.vm_done, self.ring_id, self.tag_id) = st
                self.emit_ops(u, [o for o in tail if not o[4]], half, side, renumber=tail)
                assert (end0[0], end0[2]) == (self.ds_seq, self.vm_seq), "both bodies must issue the same memory operations"
                self.ds_done, self.vm_done = min(self.ds_done, end0[1]), min(self.vm_done, end0[3])  # what both guarantee
                self.emit(".Ljoin_" + lbl + ":")
            else:
                self.emit_ops(u, ops, 0, side)
            self.stamp(u)

    def short_b(self, u):
        """short form of k-tile 12: its B operands were read into BX at step NT - 5 (LDS results: no wait states before an MFMA reads them)"""
        for cs in range(self.NS):
            self.wait_ds(self.tag_id["BX%d" % cs])

    def emit_ops(self, u, ops, pos0, side, renumber=None, bx=False):
        """MFMAs of a step from position pos0 on, each followed by its side work.  `renumber`: the full op list whose
        positions the side table refers to (body without the wave-0-only MFMAs: side work of a skipped position is issued
        behind the previous emitted MFMA)."""
        full = renumber if renumber is not None else ops
        for k, op in enumerate(full):
            pos = pos0 + k
            present = op in ops if renumber is not None else True
            if present:
                cs, ai, e, r, w0 = op
                key = (u, e // 4 if e < 12 else 3)
                self.wait_vm(self.ring_id[key])
                c = None
                if u == 0 and r == 0:
                    c = self.T[ai]
                    self.wait_ds(self.tag_id["T%d" % ai])
                bsrc = self.BX[cs] if bx else (self.HO[cs] if u == 0 else self.B[u & 1][cs])
                self.mfma(self.ACC[cs][ai], self.areg(u, e), bsrc + r, c)
            for fn in side.get(pos, []):
                fn()

    # ---------------- whole statements ----------------
    def gen_layers(self):
        A, NS, NT = self.A, self.NS, self.NT
        self.lines = []
        e = self.emit
        lbl = "%="
        T4 = self.T
        # ---- entry ----
        e("s_waitcnt lgkmcnt(0)")
        ins = ["HW_IN", "HW_OUT", "FW_IN", "FW_OUT", "PL_IN", "PL_OUT", "PW_IN", "PW_OUT", "BIAS_A", "BIAS_R", "VOFF", "DUMMY",
               "W0A", "W0R", "WLA"] + (["XCHW", "XCHR"] if NS > 1 else [])
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
        # short form of k-tile 12: %[sw] = the wavefront's index when the net's width allows it (N <= 200), else 99 (no step ever matches)
        e("s_mov_b32 s%d, %%[sw]" % self.S_SW)
        e("s_mov_b32 s%d, 0" % self.S_QHI[0])
        e("s_mov_b32 s%d, -1" % self.S_QHI[1])
        e("v_mov_b32_e32 v%d, 0" % A["QOFF"])
        e("v_mov_b32_e32 v%d, 508" % A["HWS12_0"])
        e("v_cndmask_b32_e64 v%d, v%d, v%d, s[%d:%d]" % (A["QOFF"], A["QOFF"], A["HWS12_0"], self.S_QHI[0], self.S_QHI[1]))
        self.reset_counters()
        # ---- the stage inputs of every column set.  NS == 1: the statement's own operands.  NS > 1: each wavefront holds the inputs of
        # ITS set (lane 16 q + j: trajectory j); they are exchanged through LDS [set][16] x {x0, x1} behind one barrier.
        xin = []
        if NS == 1:
            xin = [("%[x0]", "%[x1]")]
        else:
            e("v_mov_b32_e32 v%d, %%[x0]" % X_(self, 0))
            e("v_mov_b32_e32 v%d, %%[x1]" % X_(self, 1))
            e("ds_write_b64 v%d, v[%d:%d]" % (A["XCHW"], X_(self, 0), X_(self, 1)))
            self.ds_issue()
            self.wait_ds_all()
            e("s_barrier")
            assert NS == 2
            ids = []
            for cs in range(NS):
                e("ds_read_b64 v[%d:%d], v%d offset:%d" % (self.X + 2 * cs, self.X + 2 * cs + 1, A["XCHR"], 128 * cs))
                ids.append(self.ds_issue())
            xin = [("v%d" % (self.X + 2 * cs), "v%d" % (self.X + 2 * cs + 1)) for cs in range(NS)]
            self.xwait = ids[-1]
        for cs in range(NS):
            for n in ("LO_TILE", "LO_PART", "FOLDW"):
                e("v_mov_b32_e32 v%d, v%d" % (A["%s%d" % (n, cs)], A["DUMMY"]))
            e("v_add_u32_e32 v%d, %d, v%d" % (A["DUP%d" % cs], cs * self.HSET + NT * 1024, A["HW_IN"]))
            e("v_cndmask_b32_e64 v%d, v%d, v%d, s[%d:%d]" % (A["DUP%d" % cs], A["DUMMY"], A["DUP%d" % cs], self.S_W3[0], self.S_W3[1]))
        # ---- layer 0: Linear(2, N) + LeakyReLU on the VALU, h = lrelu(fmaf(w1, x1, fmaf(w0, x0, b))); rows {b, w0, w1, 0} in LDS.
        # Row tiles wave, wave + 4, wave + 8 (own tile -> HO and the B operand of step 0) and the remainder tile (every
        # wavefront writes it: identical bits), for every column set.  Two 16-register row buffers: the next tile's rows are in
        # flight meanwhile.  t -> Y, 0.01 t -> the rows' zero pad (so the rows survive for the next column set).
        rb1 = [q for row in self.B for q in row]
        if NS == 1:
            rb1 = rb1 + [self.X, self.Y]          # NS == 1: X is free (the inputs are operands); Y is the t temporary -> use ACC quads instead
            rb1 = [self.B[0][0], self.B[1][0], self.X, self.ACC[0][3]]
        RB = [T4, rb1[:4]]
        tiles = [("W0A", 0, None, 0), ("W0A", 1024, 0, 4096), ("W0A", 2048, 1, 8192), ("W0R", 0, 2, None)]
        rd = {}

        def l0_reads(ti):
            areg, off, _, _ = tiles[ti]
            rd[ti] = [self.ds_read(RB[ti & 1][r], areg, off + 16 * r) for r in range(4)]
        l0_reads(0)
        l0_reads(1)
        if NS > 1:
            self.wait_ds(self.xwait)
        for ti, (areg, off, acci, hoff) in enumerate(tiles):
            rows = RB[ti & 1]
            self.wait_ds(rd[ti][3])
            for cs in range(NS):
                x0, x1 = xin[cs]
                dst = self.HO[cs] if acci is None else self.ACC[cs][acci]
                for r in range(4):
                    e("v_fma_f32 v%d, v%d, %s, v%d" % (self.Y + r, rows[r] + 1, x0, rows[r]))
                for r in range(4):
                    e("v_fma_f32 v%d, v%d, %s, v%d" % (self.Y + r, rows[r] + 2, x1, self.Y + r))
                for r in range(4):
                    e("v_mul_f32_e32 v%d, s%d, v%d" % (rows[r] + 3, self.S_C01, self.Y + r))
                for r in range(4):
                    e("v_max_f32_e32 v%d, v%d, v%d" % (dst + r, self.Y + r, rows[r] + 3))
            if ti + 2 < len(tiles):
                l0_reads(ti + 2)
            for cs in range(NS):
                dst = self.HO[cs] if acci is None else self.ACC[cs][acci]
                if acci is None:
                    self.ds_write("HW_IN", dst, cs * self.HSET)
                    self.ds_write("DUP%d" % cs, dst)
                elif hoff is not None:
                    self.ds_write("HW_IN", dst, cs * self.HSET + hoff)
                else:
                    self.ds_write("FW_IN", dst, cs * self.HSET)
        # bias of hidden layer 0 -> C operands of its first MFMAs
        e("ds_read_b128 %s, v%d" % (self.vr(T4[0]), A["BIAS_A"]))
        e("ds_read_b128 %s, v%d offset:256" % (self.vr(T4[1]), A["BIAS_A"]))
        e("ds_read_b128 %s, v%d offset:512" % (self.vr(T4[2]), A["BIAS_A"]))
        e("ds_read_b128 %s, v%d" % (self.vr(T4[3]), A["BIAS_R"]))
        e("s_waitcnt lgkmcnt(0)")
        for r in range(4):
            e("v_cndmask_b32_e64 v%d, 0, v%d, s[%d:%d]" % (T4[3] + r, T4[3] + r, self.S_W0[0], self.S_W0[1]))
        e("s_mov_b32 s%d, s%d" % (self.S_LNEXT, self.S_H0))  # refill source of layer l: layer l + 1, or layer 0 after the last
        self.stamp(15)   # everything outside the layer loop
        e(".Lloop_" + lbl + ":")
        # at the loop top s_lnext is the offset of layer l; the refills of layer l stream layer (l + 1 < L ? l + 1 : 0)
        e("s_mov_b32 s%d, s%d" % (self.S_LCUR, self.S_LNEXT))
        e("s_add_u32 s%d, s%d, s%d" % (self.S_LNEXT, self.S_LNEXT, self.S_LB))
        e("s_add_u32 s%d, s%d, 1" % (self.S_T, self.S_L))
        e("s_cmp_lt_u32 s%d, s%d" % (self.S_T, self.S_NL))
        e("s_cselect_b32 s%d, s%d, s%d" % (self.S_LNEXT, self.S_LNEXT, self.S_H0))
        # ---- the layer body: simulate to the fixed point of the wait counts, emit the fixed point ----
        self.reset_counters()
        body = None
        for it in range(3):
            start = len(self.lines)
            self.ool = []
            if it == 0:
                for (u, k) in self.refill_order():   # ring as primed by init / left by the previous pass
                    where, ut = self.refill_target(u, k)
                    if where == "next":
                        self.vm_seq += 1
                        self.ring_id[(ut, k)] = self.vm_seq
                for k in range(4):
                    self.ds_issue("T%d" % k)
                self.ds_done = self.ds_seq
            self.layer()
            self.bookkeeping()
            text = self.lines[start:] + ["// out of line:"] + self.ool
            del self.lines[start:]
            if it >= 1:
                if body is not None:
                    assert body == text, "wait counts did not reach a fixed point"
                body = text
        k_ool = body.index("// out of line:")
        ool_text = body[k_ool + 1:]
        self.lines += body[:k_ool]
        # ---- loop control ----
        e("s_add_u32 s%d, s%d, 1" % (self.S_L, self.S_L))
        e("s_cmp_lt_u32 s%d, s%d" % (self.S_L, self.S_NL))
        e("s_cbranch_scc1 .Lloop_" + lbl)
        # ---- exit: the last layer's accumulator 2 and partial sums (not pipelined: nothing follows) ----
        e("s_nop 7")
        e("s_nop 1")
        for cs in range(NS):
            self.lrelu_tile(self.X, self.ACC[cs][2], self.Y)
            e("ds_write_b128 v%d, %s" % (A["LO_TILE%d" % cs], self.vr(self.X)))
            e("ds_write_b128 v%d, %s" % (A["LO_PART%d" % cs], self.vr(self.ACC[cs][3])))
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        self.stamp(12)  # (diagnostic: charged to the last step)
        # ---- Linear(N, 1) for the wavefront's OWN column set: four partial fmaf chains (one per lane group q) over kt, r; fixed
        # combine tree ((p0+p1)+(p2+p3)) + bl.  After the last swap the *_IN names are the buffers the last hidden layer wrote;
        # %[own_h] / %[own_p] are the own set's byte offsets inside an activation / partial-sum buffer (0 for NS == 1).
        self.reset_counters()
        PART, TMP, HBv, PLv = A["PART"], A["TMPA"], A["XCHW"], A["XCHR"]
        e("v_add_u32_e32 v%d, %d, v%d" % (HBv, -(NT - 1) * 1024 & 0xffffffff, A["FW_IN"]))  # slot 0 of the buffer, this lane
        e("v_mov_b32_e32 v%d, v%d" % (PLv, A["PL_IN"]))
        if NS > 1:
            e("v_add_u32_e32 v%d, %%[own_h], v%d" % (HBv, HBv))
            e("v_add_u32_e32 v%d, %%[own_p], v%d" % (PLv, PLv))
        # register quads free here: the accumulators, the bias quads, the B buffers, HO, X, Y
        Q = [a for row in self.ACC for a in row] + T4 + [q for row in self.B for q in row] + self.HO + [self.X, self.Y]
        if NS == 1:
            Q = self.ACC[0] + T4 + [self.B[0][0], self.B[1][0], self.HO[0], self.X, self.Y]
        else:
            Q = self.ACC[0] + T4 + [self.B[0][0], self.B[1][0], self.HO[0], self.X, self.Y]
        NTF = NT - 1   # tiles 0 .. NT-2 are activations; tile NT-1 is folded from the partial sums
        pr = [Q[8], Q[9], Q[11], Q[12]]
        idp = []
        for k in range(4):
            e("ds_read_b128 %s, v%d offset:%d" % (self.vr(pr[k]), PLv, k * 1024))
            idp.append(self.ds_issue())
        idw12 = self.ds_read(Q[7], "WLA", (NT - 1) * 64)
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
                e("ds_read_b128 %s, v%d offset:%d" % (self.vr(hreg), HBv, k * 1024))
                ids[(k, "h")] = self.ds_issue()
                ids[(k, "w")] = self.ds_read(wreg, "WLA", k * 64)
        ll_reads(0)
        self.wait_ds(idp[3])
        self.fold_tile(Q[10], pr, Q[12])   # -> Q[10] (HO quad); pr[3] = Q[12] doubles as the LeakyReLU temporary after the sums
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
            e("v_fmac_f32_e32 v%d, v%d, v%d" % (PART, Q[7] + r, Q[10] + r))
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
        if ool_text:
            # the out-of-line bodies of the layer pass (short form of k-tile 12, wave 0's last step): jumped over once per evaluation
            e("s_branch .Lend_" + lbl)
            self.lines += ool_text
            e(".Lend_" + lbl + ":")
        return self.lines

    def bookkeeping(self):
        """end of a layer pass: where the leftover of this layer goes, and the in/out swap of the double buffers"""
        A, e = self.A, self.emit
        for cs in range(self.NS):
            e("v_add_u32_e32 v%d, %d, v%d" % (A["LO_TILE%d" % cs], cs * self.HSET + 8192, A["HW_OUT"]))
            if cs == 0:
                e("v_mov_b32_e32 v%d, v%d" % (A["LO_PART%d" % cs], A["PW_OUT"]))
                e("v_mov_b32_e32 v%d, v%d" % (A["FOLDW%d" % cs], A["FW_OUT"]))
            else:
                e("v_add_u32_e32 v%d, %d, v%d" % (A["LO_PART%d" % cs], cs * self.PSET, A["PW_OUT"]))
                e("v_add_u32_e32 v%d, %d, v%d" % (A["FOLDW%d" % cs], cs * self.HSET, A["FW_OUT"]))
        for a, b in (("HW_IN", "HW_OUT"), ("FW_IN", "FW_OUT"), ("PL_IN", "PL_OUT"), ("PW_IN", "PW_OUT")):
            e("v_swap_b32 v%d, v%d" % (A[a], A[b]))

    def refill_order(self):
        """fragment slots in the order one layer pass reloads them"""
        order = []
        for u in range(self.NT):
            ops = self.step_mfmas(u)
            last_use = {}
            for pos, (cs, ai, e, r, w0) in enumerate(ops):
                last_use[e // 4 if e < 12 else 3] = pos
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


def X_(g, k):
    return g.Y + k   # scratch pair for the input exchange


def cstring(lines):
    return "\n".join('  "%s\\n\\t"' % l for l in lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nt", type=int, default=13)
    ap.add_argument("--pd", type=int, default=7)
    ap.add_argument("--ns", type=int, default=1, help="16-trajectory column sets per tile (1 or 2)")
    ap.add_argument("--stamps", action="store_true")
    ap.add_argument("--no-short12", action="store_true", help="A/B: without the two-MFMA form of the half-padded k-tile 12")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    sb = 80 if a.stamps else 84   # (the diagnostic build needs four more scalar registers)
    g = Gen(a.nt, a.pd, a.ns, sb=sb, stamps=a.stamps, short12=not a.no_short12)
    layers = list(g.gen_layers())
    init = list(Gen(a.nt, a.pd, a.ns, sb=sb, stamps=a.stamps, short12=not a.no_short12).gen_init())
    nmf = sum(1 for l in layers if l.startswith("v_mfma"))
    sfx = "%d" % a.nt + ("x%d" % a.ns if a.ns > 1 else "")
    with open(a.out, "w") as f:
        f.write("// GENERATED by tools/gen_mlp_asm.py --nt %d --pd %d --ns %d -- do not edit.\n" % (a.nt, a.pd, a.ns))
        f.write("// %d instructions, %d MFMAs in the layer body (both bodies of the last step counted); working set v[%d:255], ring a[0:%d].\n"
                % (len(layers), nmf, g.VB, g.ring_regs - 1))
        f.write("#define IONODE_MLPASM_VB_%s %d\n" % (sfx, g.VB))
        f.write("#define IONODE_MLPASM_RING_REGS_%s %d\n" % (sfx, g.ring_regs))
        f.write("#define IONODE_MLPASM_INIT_%s \\\n%s\n" % (sfx, cstring(init).replace("\n", " \\\n")))
        f.write("#define IONODE_MLPASM_LAYERS_%s \\\n%s\n" % (sfx, cstring(layers).replace("\n", " \\\n")))
        vclob = ", ".join('"v%d"' % r for r in range(g.VB, 256))
        aclob = ", ".join('"a%d"' % r for r in range(g.ring_regs + (1 if a.stamps else 0)))
        sclob = ", ".join('"s%d"' % r for r in range(g.SB, g.SB + g.n_sgpr))
        f.write("#define IONODE_MLPASM_CLOBBER_V_%s %s\n" % (sfx, vclob))
        f.write("#define IONODE_MLPASM_CLOBBER_A_%s %s\n" % (sfx, aclob))
        f.write("#define IONODE_MLPASM_CLOBBER_S_%s %s\n" % (sfx, sclob))


if __name__ == "__main__":
    main()
