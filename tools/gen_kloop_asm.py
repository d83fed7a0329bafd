"""Generates manipose_amd/csrc/kloop_asm.inc: the hand-scheduled 64-wide k-step of gemm_bf16_persist_kernel (gemm_bf16.hip) as ONE inline-asm
block per step - 64 v_mfma_f32_16x16x32_bf16 of a wave's 128 x 64 sub-tile, the 24 LDS fragment reads (ds_read_b128; the "T" operand of the
dgrad through ds_read_b64_tr_b16), and the step's share of the operand DMA (global_load_lds_dwordx4), in a written-out order:

  * fragments are requested TWO groups of 8 MFMAs ahead (hipcc's schedule: one group ahead, every wait an lgkmcnt(0)), into a ring of three A
    slots and two B sets held in FIXED registers v[200:255] (clobbers of the block: sub-registers of an asm operand cannot be named, and the
    transpose reads fill a fragment in two 64-bit halves); every wait is a counted lgkmcnt(N) that leaves the younger requests in flight;
  * the first fragments of the step are requested BEFORE the step's DMA instructions are issued (hipcc: after), and the DMA instructions are
    spread over the first four MFMA groups, one behind the third and one behind the sixth MFMA of a group (job b in groups 0-1, job a in 2-3);
  * no VALU / SALU instruction except the DMA's m0 set-up stands between the MFMAs.

    python tools/gen_kloop_asm.py            # rewrites the .inc
    python tools/gen_kloop_asm.py --check    # exit 1 if the committed .inc differs (CPU test)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "manipose_amd", "csrc", "kloop_asm.inc")

FA0, FB0 = 200, 224          # A ring: 3 slots x 2 fragments x 4 registers; B: 2 sets (k-step parity) x 4 fragments x 4 registers (set per variant: `base`)
NGROUP = 8                   # groups of 8 MFMAs: group g = (k-step g / 4, A fragments 2 (g % 4), 2 (g % 4) + 1) against the four B fragments


def vr(base, n=4):
    return f"v[{base}:{base + n - 1}]"


def fa(slot, f):             # A fragment f (0 / 1) of ring slot `slot`
    return FA0 + (slot * 2 + f) * 4


def fb(ks, j):
    return FB0 + (ks * 4 + j) * 4


def reads_a(g):
    """requests of the two A fragments of group g (k-step (g % 8) / 4, 16-row blocks 2 (g % 4), + 1) into ring slot g % 3"""
    ks, p = (g % NGROUP) // 4, g % 4
    return [f"ds_read_b128 {vr(fa(g % 3, f))}, %[aa{ks}] offset:{(2 * p + f) * 2048}" for f in range(2)], 2


def reads_b(kidx, trb, img="ba"):
    """requests of the four B fragments of the k-step with running index kidx (register set kidx % 2, address operands of image `img`)"""
    out = []
    ks = kidx % 2
    if trb == 0:
        for j in range(4):
            out.append(f"ds_read_b128 {vr(fb(ks, j))}, %[{img}{ks}] offset:{j * 2048}")
        return out, 4
    t = {"ba": "bt", "bb": "bu"}[img]
    for j in range(4):       # "T" image (256 output columns per reduction row, 512-byte rows): two transpose reads, 4 rows apart, per fragment
        for h in range(2):
            out.append(f"ds_read_b64_tr_b16 {vr(fb(ks, j) + 2 * h, 2)}, %[{t}{j}] offset:{ks * 16384 + h * 2048}")
    return out, 8


def dma(job, n, guarded=True):
    """DMA instruction n of job 'a' / 'b': 1 KiB = 64 lanes x 16 bytes from (SGPR base + per-lane offset) to the LDS address in m0"""
    L = f".Lkd{job}{n}_%="
    body = [f"s_add_u32 m0, %[lds{job}], {n * 1024}",
            "s_nop 0",
            f"global_load_lds_dwordx4 %[{job}o{n}], %[gb{job}]"]
    if not guarded:
        return body
    return [f"s_cmp_eq_u32 %[en{job}], 0", f"s_cbranch_scc1 {L}"] + body + [f"{L}:"]


def dma_job(job, mode="cond", pieces=4):
    """all instructions of a job: mode 'always' (unconditional), 'cond' (behind ONE test of its enable flag), 'none' (nothing)"""
    if mode == "none":
        return []
    out = []
    for n in range(pieces):
        out += dma(job, n, guarded=False)
    if mode == "always":
        return out
    L = f".Lkj{job}_%="
    return [f"s_cmp_eq_u32 %[en{job}], 0", f"s_cbranch_scc1 {L}"] + out + [f"{L}:"]


def bias_dma(mode="always"):
    if mode == "none":
        return []
    out = ["s_mov_b32 m0, %[ldsc]", "s_nop 0", "global_load_lds_dword %[co], %[gbc]"]
    if mode == "always":
        return out
    return ["s_cmp_eq_u32 %[enc], 0", "s_cbranch_scc1 .Lkdc_%="] + out + [".Lkdc_%=:"]


# Job configurations of a step: which DMA jobs it can carry (a / b = the four pieces of an A / B operand tile, c = the bias row); a job is
# 'always' there, 'none', or 'cond' = behind one test of a wave-uniform flag in a scalar register (%[ena] / %[enb] / %[enc]):
#   X0 / X1 / X2  the three steps of a k-tile of the split-precision loop: (A_lo, B_hi) requests A_hi; (A_hi, B_hi) requests B_lo and - unless
#                 this is the launch's last k-tile - the next A_lo; (A_hi, B_lo) requests the next B_hi likewise, and the bias behind the tile's last step
#   P             a step of the plain loop: both operands of the next k-tile unless it is the launch's last, the bias behind a tile's last step
# ONE asm block per step type: two blocks on the two sides of a branch make hipcc reconcile the 128 accumulator registers through scratch memory.
CONFIGS = {"X0": dict(a="always", b="none", c="none"), "X1": dict(a="cond", b="always", c="none"), "X2": dict(a="none", b="cond", c="cond"),
           "P": dict(a="cond", b="cond", c="cond"),
           # the same steps with the operand DMA issued by the YOUNGER wave of every SIMD only (waves 4-7: eight pieces per tile each, every job
           # behind a flag that is 0 in waves 0-3): the older wave wins the matrix pipe's arbitration, so DMA instructions in ITS head delay the
           # SIMD's first MFMA of the step, while the younger wave waits for the pipe anyway
           "Z0": dict(a="cond", b="none", c="none", pieces=8), "Z1": dict(a="cond", b="cond", c="none", pieces=8),
           "Z2": dict(a="none", b="cond", c="cond", pieces=8), "ZP": dict(a="cond", b="cond", c="cond", pieces=8)}
# schedule variants of the f16f8 steps (MP_KSTEP_VARIANT_F8): measured on the block of four GEMMs at B = 79, same box (profiles/r06_probes/f16f8_step_variants.log)
F8_VARIANTS = {0: dict(dma="head_before", ahead=1), 1: dict(dma="head_before", ahead=2), 2: dict(dma="head_after", ahead=2)}
# (emitted separately, "N" operands only) the f16f8 loop's steps
F8_CONFIGS = {"ZF": dict(a="cond", b="cond", c="none", pieces=8), "ZE": dict(a="cond", b="cond", c="cond", pieces=8)}

# Variants (MP_KSTEP_VARIANT selects one at compile time; the A/B of round 5 is in DESIGN.md section 5):
#   dma    'spread'       one DMA instruction behind the third and the sixth MFMA of groups 0-3 (job b in groups 0-1, job a in 2-3)
#          'head_before'  both jobs in front of the step's first fragment request (what hipcc makes of the HIP source)
#          'head_after'   both jobs behind the step's first fragment requests, in front of the first wait (their issue overlaps the LDS latency)
#   ahead  1 / 2          fragment requests one / two groups of 8 MFMAs ahead of their use
VARIANTS = {
    0: dict(dma="head_before", ahead=1),     # hipcc's own order, written out (measured: best for the plain loop)
    2: dict(dma="head_after", ahead=2),      # (measured: best for the split-precision loop; 1 = head_before / ahead 2 and 3 = spread / ahead 2 were slower)
}


def step(trb, cfg, dma="spread", ahead=2, base=200, pad=False, mfma="v_mfma_f32_16x16x32_bf16"):
    """the instruction list of one step; `pend` = request batches in flight, oldest first, as (group that needs them, count)"""
    global FA0, FB0
    FA0, FB0 = base, base + 24
    ins = []
    pend = []
    jm = CONFIGS.get(cfg) or F8_CONFIGS[cfg]

    def request(lines, n, needed_by):
        ins.extend(lines)
        pend.append((needed_by, n))

    def wait_for(g):          # everything group g needs has landed; younger batches stay in flight
        younger = sum(n for need, n in pend if need > g)
        ins.append(f"s_waitcnt lgkmcnt({younger})")
        if pad:
            ins.append("s_nop 0")
        pend[:] = [(need, n) for need, n in pend if need > g]

    images = jm.get("images", ("ba",))
    npc = jm.get("pieces", 4)
    ng = NGROUP * len(images)             # groups of 8 MFMAs in this block: 8 per (A image, B image) pair

    def img_of(g):
        return images[g // NGROUP]

    if dma == "head_before":
        ins.extend(dma_job("b", jm["b"], npc) + dma_job("a", jm["a"], npc))
    rb, nb = reads_b(0, trb, img_of(0))
    request(rb, nb, 0)
    r, n = reads_a(0)
    request(r, n, 0)
    if ahead >= 2:
        r, n = reads_a(1)
        request(r, n, 1)
    if dma == "head_after":
        ins.extend(dma_job("b", jm["b"], npc) + dma_job("a", jm["a"], npc))
    wait_for(0)
    for g in range(ng):
        kidx, p = g // 4, g % 4           # running k-step index (two per image pair), A fragment pair inside it
        slot = g % 3
        mf = []
        for f in range(2):
            for j in range(4):
                c = f"%[c{2 * p + f}{j}]"
                mf.append(f"{mfma} {c}, {vr(fb(kidx % 2, j))}, {vr(fa(slot, f))}, {c}")
        ins.append(mf[0])
        # requests issued behind the first MFMA of the group: the fragments of group g + ahead (B of the next k-step with its first A)
        t = g + ahead
        if t < ng:
            if t % 4 == 0:
                rb, nb = reads_b(t // 4, trb, img_of(t))
                request(rb, nb, t)
            r, n = reads_a(t)
            request(r, n, t)
        ins.extend(mf[1:3])
        if dma == "spread" and g < 4 and jm["ba"[g // 2]] != "none":
            ins.extend(globals()["dma"]("ba"[g // 2], 2 * (g % 2), guarded=jm["ba"[g // 2]] == "cond"))
        ins.extend(mf[3:6])
        if dma == "spread" and g < 4 and jm["ba"[g // 2]] != "none":
            ins.extend(globals()["dma"]("ba"[g // 2], 2 * (g % 2) + 1, guarded=jm["ba"[g // 2]] == "cond"))
        if g == 4:
            ins.extend(bias_dma(jm["c"]))
        ins.extend(mf[6:8])
        if g + 1 < ng:
            wait_for(g + 1)
    assert not pend, pend
    return ins


def step_f8(cfg, dma="head_after", ahead=2, base=200):
    """The correction step of the "f16f8" operand form (gemm_bf16.hip, SPLIT = 8) as one block: 32 v_mfma_scale_f32_16x16x128_f8f6f4 of a wave's
    128 x 64 sub-tile - one per 16 x 16 block, 128 reduction BYTES deep (the whole 64-index k-tile of the 2-byte correction planes), 8 passes each:
    the same 1024 matrix-pipe cycles as the 64 bf16 / fp16 MFMAs of the other steps.  A fragment = the lane's 16-byte chunks g and g + 4 of its
    row (two ds_read_b128 through the step's two A address operands: exactly the k-step 0 / 1 fragments of the 16-bit steps), 8 registers; ring of
    three A fragments (one group of four MFMAs each), the four B fragments stay for the step: 24 + 32 = the same 56 fixed registers.  The E8M0 block
    scales (2^-15 on the first source, 1 on the second; all lanes and blocks alike) come in two VGPR operands.  Requests: B0, A0 first - the first
    MFMA waits for those four reads only - then B1..B3 and A1; A(g + ahead) behind the first MFMA of group g; counted waits throughout."""
    ins, pend = [], []          # pend: request batches in flight, oldest first, as (tag, count)
    jm = F8_CONFIGS[cfg]
    npc = jm.get("pieces", 4)
    A0, B0 = base, base + 24

    def req(tag, lines):
        ins.extend(lines)
        pend.append((tag, len(lines)))

    def wait(tag):              # the batch `tag` (and everything older: LDS returns in order) has landed; younger batches stay in flight
        idx = [t for t, _ in pend].index(tag)
        ins.append(f"s_waitcnt lgkmcnt({sum(n for _, n in pend[idx + 1:])})")
        del pend[:idx + 1]

    def ra(i):
        r = A0 + (i % 3) * 8
        return [f"ds_read_b128 {vr(r)}, %[aa0] offset:{i * 2048}", f"ds_read_b128 {vr(r + 4)}, %[aa1] offset:{i * 2048}"]

    def rb(j):
        r = B0 + j * 8
        return [f"ds_read_b128 {vr(r)}, %[ba0] offset:{j * 2048}", f"ds_read_b128 {vr(r + 4)}, %[ba1] offset:{j * 2048}"]

    jobs = dma_job("b", jm["b"], npc) + dma_job("a", jm["a"], npc)
    if dma == "head_before":
        ins.extend(jobs)
    req("b0", rb(0)); req("a0", ra(0))
    for j in (1, 2, 3):
        req(f"b{j}", rb(j))
    if ahead >= 2:
        req("a1", ra(1))
    if dma != "head_before":
        ins.extend(jobs)
    live = {"b0", "a0", "b1", "b2", "b3"} | ({"a1"} if ahead >= 2 else set())
    for g in range(8):
        if f"a{g}" in live:
            wait(f"a{g}") if g else wait("a0")
            live.discard(f"a{g}")
        for j in range(4):
            if f"b{j}" in live and any(t == f"b{j}" for t, _ in pend):
                wait(f"b{j}")
            live.discard(f"b{j}")
            c = f"%[c{g}{j}]"
            ins.append(f"v_mfma_scale_f32_16x16x128_f8f6f4 {c}, {vr(B0 + j * 8, 8)}, {vr(A0 + (g % 3) * 8, 8)}, {c}, %[sca], %[scb] op_sel_hi:[0,0,0]")
            # (behind the group's SECOND MFMA: the ring slot's last reader - the previous group's last MFMA - is then two 8-pass MFMAs back)
            if j == 1 and g + ahead < 8:
                req(f"a{g + ahead}", ra(g + ahead)); live.add(f"a{g + ahead}")
            if j == 2 and g == 4:
                ins.extend(bias_dma(jm["c"]))
    assert not pend, pend
    return ins


def pipe_loop_x3(ahead=2):
    """The split-precision loop's k-tiles 0 .. nk-2 of a tile as ONE asm block with the loop inside (MP_KLOOP_PIPE): the fragment pipeline runs
    ACROSS the steps.  A step's barrier does not stand in front of it but two groups of 8 MFMAs before its end, in front of the first
    request that reads the NEXT step's buffers: behind `s_waitcnt vmcnt(V) lgkmcnt(0); s_barrier` every wave has finished reading this
    step's buffers (so the DMA into them may be issued at once) and the next step's tiles have landed for every wave; each wave still holds
    the fragments of two groups, whose 16 MFMAs (and the SIMD partner's) cover the latency of the next step's first requests - the ~300 idle
    cycles at the head of every step of the block-per-step form.  Buffers and DMA schedule per k-tile kt (A0 | B0 | A1 | B1):
      step 0 (A0 = A_lo, B0 = B_hi): behind its barrier A0 is free  -> A_lo[kt + 1];   needs A1 = A_hi[kt] landed: vmcnt(8) (B_lo[kt] may fly)
      step 1 (A1 = A_hi, B0):        behind its barrier B0 is free  -> B_hi[kt + 1];   needs B1 = B_lo[kt]: vmcnt(8) (A_lo[kt + 1] may fly)
      step 2 (A1, B1 = B_lo):        behind its barrier A1, B1 free -> A_hi[kt + 1], B_lo[kt + 1];   needs A0, B0 of kt + 1: vmcnt(0)
    Entry (behind the tile's first barrier): the head requests, then A_hi[0], B_lo[0].  DMA by waves 4-7 only (8 pieces of a tile each);
    source = a constant base per plane (+ 128 bytes: "the next k-tile") + the per-lane offsets, which advance by 128 bytes per k-tile.
    The tile's last k-tile runs in the block-per-step form (its jobs depend on the next tile).  "N" operand layout only (forward)."""
    global FA0, FB0
    FA0, FB0 = 200, 224
    OPB, STAGE = 32768, 65536
    ins, pend = [], []
    A = {0: ("a0k0", "a0k1"), 1: ("a1k0", "a1k1")}
    B = {0: ("b0k0", "b0k1"), 1: ("b1k0", "b1k1")}
    bufs = [(0, 0), (1, 0), (1, 1)]                  # (A buffer, B buffer) of steps 0 / 1 / 2

    def rd_a(G, abuf):
        g = G % 8
        return [f"ds_read_b128 {vr(fa(G % 3, f))}, %[{A[abuf][g // 4]}] offset:{(2 * (g % 4) + f) * 2048}" for f in range(2)], 2

    def rd_b(G, bbuf):
        ks = (G % 8) // 4
        return [f"ds_read_b128 {vr(fb(ks, j))}, %[{B[bbuf][ks]}] offset:{j * 2048}" for j in range(4)], 4

    def request(lines, n, needed_by):
        ins.extend(lines)
        pend.append((needed_by, n))

    def wait_for(G):
        if not any(need <= G for need, n in pend):
            return
        younger = sum(n for need, n in pend if need > G)
        ins.append(f"s_waitcnt lgkmcnt({younger})")
        pend[:] = [(need, n) for need, n in pend if need > G]

    def job(lds_const, base, offs, imm=0):
        out = []
        for n in range(8):
            out += [f"s_add_u32 m0, %[lds], {lds_const + n * 1024}", "s_nop 0",
                    f"global_load_lds_dwordx4 %[{offs}{n}], %[{base}]" + (f" offset:{imm}" if imm else "")]
        return out

    def young(tag, body):
        return [f"s_cmp_eq_u32 %[dmaw], 0", f"s_cbranch_scc1 .Lkp{tag}_%="] + body + [f".Lkp{tag}_%=:"]

    # entry: the head requests of step 0, then A_hi[0] -> A1 and B_lo[0] -> B1
    r, n = rd_b(0, 0); request(r, n, 0)
    r, n = rd_a(0, 0); request(r, n, 0)
    r, n = rd_a(1, 0); request(r, n, 1)
    ins.extend(young("e", job(STAGE, "pahi0", "ao") + job(STAGE + OPB, "pblo0", "bo")))      # (an instruction offset would also move the LDS address)
    ins.append(".Lkploop_%=:")
    head_pend = list(pend)
    VM = [8, 8, 0]
    for S in range(3):
        abuf, bbuf = bufs[S]
        nabuf, nbbuf = bufs[(S + 1) % 3]
        for g in range(8):
            G = 8 * S + g
            if g == 8 - ahead:
                ins.append(f"s_waitcnt vmcnt({VM[S]}) lgkmcnt(0)")
                ins.append("s_barrier")
                pend[:] = []
            else:
                wait_for(G)
            slot, ks, p = G % 3, g // 4, g % 4
            mf = []
            for f in range(2):
                for j in range(4):
                    c = f"%[c{2 * p + f}{j}]"
                    mf.append(f"v_mfma_f32_16x16x32_bf16 {c}, {vr(fb(ks, j))}, {vr(fa(slot, f))}, {c}")
            ins.append(mf[0])
            t = G + ahead
            nxt = t >= 8 * (S + 1)                   # a request into the next step's buffers (behind this step's barrier)
            # step 1 multiplies the B tile of step 0 again: its B fragments are still in the two register sets, no request
            if t % 4 == 0 and (t % 24) // 8 != 1:
                r, n = rd_b(t, nbbuf if nxt else bbuf); request(r, n, t)
            r, n = rd_a(t, nabuf if nxt else abuf); request(r, n, t)
            if g == 8 - ahead:
                if S == 0:
                    ins.extend(young("a", job(0, "palo1", "ao")))
                elif S == 1:
                    ins.extend(young("b", job(OPB, "pbhi1", "bo")))
                else:
                    adv = [f"v_add_u32 %[ao{n}], 128, %[ao{n}]" for n in range(8)] + [f"v_add_u32 %[bo{n}], 128, %[bo{n}]" for n in range(8)]
                    ins.extend(young("c", job(STAGE, "pahi1", "ao") + job(STAGE + OPB, "pblo1", "bo") + adv))
            ins.extend(mf[1:])
    # the requests in flight at the back edge are those in flight at the loop's head (same counts, group numbers modulo 24)
    assert [(need - 24, n) for need, n in pend] == head_pend, (pend, head_pend)
    ins += ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 .Lkploop_%=",
            "s_waitcnt lgkmcnt(0)"]                  # (the last iteration's requests for a step 0 that runs in the block-per-step form: let them land)
    return ins


HEADER = """// GENERATED by tools/gen_kloop_asm.py - do not edit (tests/test_host_cpu.py checks that it is up to date).
// One 64-wide k-step of a wave's 128 x 64 sub-tile as a single inline-asm block: see the generator's docstring for the schedule.
// Registers v[200:255] hold the fragment ring (clobbered); operands: the 32 accumulator tuples, the LDS fragment addresses, the DMA jobs.
"""


def emit():
    out = [HEADER]
    for v, opt in sorted(VARIANTS.items()):
        for cfg in CONFIGS:
            for trb in (0, 1):
                lines = step(trb, cfg, **opt)
                out.append(f"#define MP_KSTEP_ASM_{cfg}_TRB{trb}_V{v} \\")
                for i, l in enumerate(lines):
                    out.append(f'  "{l}\\n"' + (" \\" if i + 1 < len(lines) else ""))
                out.append("")
    # the two steps of a k-tile of the "f16f8" loop (SPLIT = 8; forward, "N" operands, DMA by waves 4-7): F = the fp16 planes (the plain step with the
    # fp16 instruction; it always requests the k-tile's correction tiles), E = the correction planes (requests the next k-tile's / tile's fp16 tiles)
    for v, opt in sorted(F8_VARIANTS.items()):
        lines = step(0, "ZF", mfma="v_mfma_f32_16x16x32_f16", **opt)
        out.append(f"#define MP_KSTEP_ASM_ZF_TRB0_V{v} \\")
        for i, l in enumerate(lines):
            out.append(f'  "{l}\\n"' + (" \\" if i + 1 < len(lines) else ""))
        out.append("")
        lines = step_f8("ZE", **opt)
        out.append(f"#define MP_KSTEP_ASM_ZE_TRB0_V{v} \\")
        for i, l in enumerate(lines):
            out.append(f'  "{l}\\n"' + (" \\" if i + 1 < len(lines) else ""))
        out.append("")
    lines = pipe_loop_x3()
    out.append("#define MP_KPIPE_X3_ASM \\")
    for i, l in enumerate(lines):
        out.append(f'  "{l}\\n"' + (" \\" if i + 1 < len(lines) else ""))
    out.append("")
    acc = ", ".join(f'[c{i}{j}] "+v"(acc[{i}][{j}])' for i in range(8) for j in range(4))
    out.append(f"#define MP_KSTEP_ACC_OPERANDS {acc}")
    for v, opt in sorted(VARIANTS.items()):
        b = opt.get("base", 200)
        clob = ", ".join(f'"v{r}"' for r in range(b, b + 56))
        out.append(f'#define MP_KSTEP_CLOBBERS_V{v} "memory", "scc", "m0", {clob}')      # m0: every block's DMA set-up writes it (a later compiler-made global_load_lds / readlane must re-initialise it)
    out.append("")
    return "\n".join(out)


if __name__ == "__main__":
    text = emit()
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == text else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT)
